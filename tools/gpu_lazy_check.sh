#!/bin/bash
# lazy BatchNorm statistics: its tests, then a one-box A/B of the whole step (ops.LAZY_BN bits), per-kernel comparison under rocprofv3
out=gpurun_out/r04_lazy; mkdir -p $out
export TMPDIR=/tmp
timeout -k 10 600 python -m pytest tests/test_lazy_bn_gpu.py -x -q -s > $out/test_lazy.log 2>&1; rc=$?
tail -6 $out/test_lazy.log
[ $rc -ne 0 ] && exit $rc
for rep in 1 2; do
  for f in 0 1 2 3; do
    timeout -k 10 200 python3 bench.py --steps 30 --warmup 5 --no-cpu-baseline --no-roofline --no-other --flag ops.LAZY_BN=$f > $out/lazy$f.$rep.json 2> $out/lazy$f.$rep.err || { tail -5 $out/lazy$f.$rep.err; exit 1; }
    python3 -c "import json;d=json.load(open('$out/lazy$f.$rep.json'));print('LAZY_BN=$f', d['ms_per_step'], d['config'].get('final_loss'))"
  done
done
for f in 0 3; do
  rocprofv3 --kernel-trace --output-format csv -d $out/trace_l$f -- python3 bench.py --steps 4 --warmup 2 --no-cpu-baseline --no-roofline --no-other --flag ops.LAZY_BN=$f > $out/two_l$f.json 2> $out/two_l$f.err
  python3 tools/trace_by_kernel.py $out/trace_l$f 2 70 > $out/by_kernel_two_l$f.txt
  rm -rf $out/trace_l$f
done
