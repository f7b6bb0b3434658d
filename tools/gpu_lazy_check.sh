#!/bin/bash
# lazy BatchNorm statistics: its tests, then a one-box A/B of the whole step (ops.LAZY_BN = 1 / 0), then the rest of the GPU suite
out=gpurun_out/r04_lazy; mkdir -p $out
timeout -k 10 600 python -m pytest tests/test_lazy_bn_gpu.py -x -q -s > $out/test_lazy.log 2>&1; rc=$?
tail -15 $out/test_lazy.log
[ $rc -ne 0 ] && exit $rc
for rep in 1 2; do
  for f in 1 0; do
    timeout -k 10 200 python3 bench.py --steps 30 --warmup 5 --no-cpu-baseline --no-roofline --no-other --flag ops.LAZY_BN=$f > $out/lazy$f.$rep.json 2> $out/lazy$f.$rep.err || { tail -5 $out/lazy$f.$rep.err; exit 1; }
    python3 -c "import json;d=json.load(open('$out/lazy$f.$rep.json'));print('LAZY_BN=$f', d['ms_per_step'], d['config'].get('final_loss'))"
  done
done
timeout -k 10 1500 python -m pytest tests -m gpu -x -q > $out/full.log 2>&1; rc=$?
tail -8 $out/full.log
exit $rc
