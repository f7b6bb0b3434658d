#!/bin/bash
# training-step bench under alternative tuning keys, interleaved on one box (3 rounds); each argument is a bench.py option string:
#   tools/gpu_train_tune_ab.sh "" "--tune bn_fin_tiles=0" "--tune bn_fin_tiles=128"
set -e
for rep in 1 2 3; do
  for opt in "$@"; do
    timeout -k 10 200 python bench.py --no-cpu-baseline --no-roofline --no-live-profile $opt > gpurun_out/train_tune_ab.json
    python - <<PY
import json
r = json.loads(open("gpurun_out/train_tune_ab.json").read().strip().splitlines()[-1]); print("[$opt]", r["value"], r["ms_per_step"])
PY
  done
done
