#!/bin/bash
# one-box ceiling measurement: the training step with the BatchNorm finalize launches skipped (results wrong, timing valid)
out=gpurun_out/r04_diag_fin; mkdir -p $out
for f in 0 1 2 3 0 3; do
  python3 bench.py --steps 30 --warmup 5 --no-cpu-baseline --no-roofline --no-other --flag ops.DIAG_SKIP_FINALIZE=$f > $out/f$f.json 2> $out/f$f.err || true
  python3 -c "import json;d=json.load(open('$out/f$f.json'));print('skip=$f', d['ms_per_step'])"
done
