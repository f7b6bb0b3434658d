#!/bin/bash
# concurrency profile of the captured two-stream step under rocprofv3 --kernel-trace
name=${1:-r06_ov}; shift
out=gpurun_out/$name; mkdir -p $out
export TMPDIR=/tmp
rocprofv3 --kernel-trace --output-format csv -d $out/tr -- python3 bench.py --steps 8 --warmup 4 --no-cpu-baseline --no-roofline --no-other "$@" > $out/bench.json 2> $out/bench.err || { tail -5 $out/bench.err; exit 1; }
python3 tools/trace_overlap.py $out/tr 5 > $out/overlap.txt
python3 tools/trace_by_kernel.py $out/tr 5 45 > $out/by_kernel.txt
rm -rf $out/tr
cat $out/overlap.txt
