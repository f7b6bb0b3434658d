#!/bin/bash
# kernel trace of the two-stream hipGraph step with the deferred weight-gradient phase: per-kernel totals + the tail of a step
name=${1:-r06_prof}; shift
out=gpurun_out/$name; mkdir -p $out
export TMPDIR=/tmp
rocprofv3 --kernel-trace --output-format csv -d $out/tr -- python3 bench.py --steps 8 --warmup 4 --no-cpu-baseline --no-roofline --no-other "$@" > $out/bench.json 2> $out/bench.err || { tail -5 $out/bench.err; exit 1; }
python3 tools/trace_by_kernel.py $out/tr 5 30 > $out/by_kernel.txt
python3 tools/trace_tail.py $out/tr 16 > $out/tail.txt
rm -rf $out/tr
cat $out/tail.txt; head -12 $out/by_kernel.txt
