#!/bin/bash
# the C = 256 fused eval FFN: tests, stand-alone time per tuning variant, and the extraction bench with and without it (one box)
set -e
timeout -k 10 300 python -m pytest tests/test_bf16_storage_gpu.py -q -x -k "ffn" 2>&1 | tail -3
for k in 1 2 4 1; do timeout -k 10 120 python tools/ffn256_time.py --fused-only ffn256=$k; done
for k in 1 0 1 0; do
  timeout -k 10 200 python bench.py --mode infer --no-cpu-baseline --no-roofline --tune ffn256=$k > gpurun_out/inf_f$k.json
  python - <<PY
import json
r = json.loads(open("gpurun_out/inf_f$k.json").read().strip().splitlines()[-1]); print("ffn256=$k", r["value"], r["ms_per_step"])
PY
done
