#!/bin/bash
# Round-3 GPU check: new kernels first (short, under their own timeout), then the timed-size parity tests, a one-box A/B of the
# fused BatchNorm-backward loads, the whole GPU suite and the default bench line. Everything lands in gpurun_out/<name>/.
name=${1:-check}; out=gpurun_out/$name; mkdir -p $out
set -o pipefail
timeout -k 10 300 python -m pytest tests/test_bf16_storage_gpu.py -k "operand_load or emits_bn" -x -q > $out/t_abn.log 2>&1 || { tail -30 $out/t_abn.log; exit 1; }
tail -2 $out/t_abn.log
timeout -k 10 600 python -m pytest tests/test_b256_gpu.py -x -q -s > $out/t_b256.log 2>&1; echo "b256 rc=$?"; tail -5 $out/t_b256.log
cp gpurun_out/b256_measured.json $out/ 2>/dev/null
for rep in 1 2; do
  for cfg in "fused:" "unfused:--flag ops.FUSE_BN_BWD_APPLY=0"; do
    tag=${cfg%%:*}; flags=${cfg#*:}
    timeout -k 10 200 python bench.py --no-cpu-baseline --no-roofline --no-other --steps 40 $flags > $out/ab_$tag.$rep.json 2> $out/ab_$tag.$rep.err || { tail -5 $out/ab_$tag.$rep.err; exit 1; }
    python - $out/ab_$tag.$rep.json $tag <<'PY'
import json,sys
d=json.load(open(sys.argv[1])); print(f"{sys.argv[2]:10s} {d['ms_per_step']} ms/step  loss {d['config']['final_loss']}", flush=True)
PY
  done
done
