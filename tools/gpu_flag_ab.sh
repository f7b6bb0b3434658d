#!/bin/bash
# extraction bench under alternative host flags / tuning keys, interleaved on one box; each argument is a bench.py option string:
#   tools/gpu_flag_ab.sh "" "--tune ws_gemm=0" "--tune ffn256=0"
set -e
for rep in 1 2; do
  for opt in "$@"; do
    timeout -k 10 200 python bench.py --mode infer --no-cpu-baseline --no-roofline $opt > gpurun_out/flag_ab.json
    python - <<PY
import json
r = json.loads(open("gpurun_out/flag_ab.json").read().strip().splitlines()[-1]); print("[$opt]", r["value"], r["ms_per_step"])
PY
  done
done
