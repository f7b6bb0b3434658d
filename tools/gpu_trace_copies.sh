#!/bin/bash
# kernel trace of a short extraction run; prints where the copy / fill kernels sit: tools/gpu_trace_copies.sh NAME
name=${1:-tcopies}; out=gpurun_out/$name; mkdir -p $out
export TMPDIR=/tmp
rocprofv3 --kernel-trace --output-format csv -d $out/tr -- python3 bench.py --mode infer --clips 20480 --no-cpu-baseline --no-roofline --no-live-profile > $out/bench.json 2> $out/bench.err || { tail -5 $out/bench.err; exit 1; }
python3 tools/trace_copies.py $out/tr | tee $out/copies.txt
rm -rf $out/tr
