#!/bin/bash
out=gpurun_out/r04_lazy2; mkdir -p $out
export TMPDIR=/tmp
for f in 0 1; do
  rocprofv3 --kernel-trace --output-format csv -d $out/trace_l$f -- python3 bench.py --steps 4 --warmup 2 --no-cpu-baseline --no-roofline --no-other --flag ops.LAZY_BN=$f > $out/two_l$f.json 2> $out/two_l$f.err
  python3 -c "import json;d=json.load(open('$out/two_l$f.json'));print('two-stream under rocprof LAZY=$f', d['ms_per_step'])"
  python3 tools/trace_by_kernel.py $out/trace_l$f 2 70 > $out/by_kernel_two_l$f.txt
  rm -rf $out/trace_l$f
done
