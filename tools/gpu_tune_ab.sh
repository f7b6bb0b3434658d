#!/bin/bash
# extraction bench under alternative values of tuning keys, interleaved on one box:
#   tools/gpu_tune_ab.sh "tall_min=1000000000" "tall_min=2048" ...
# ("-" = defaults)
set -e
for rep in 1 2; do
  for kv in "$@"; do
    arg=""; [ "$kv" != "-" ] && for one in $kv; do arg="$arg --tune $one"; done
    timeout -k 10 200 python bench.py --mode infer --no-cpu-baseline --no-roofline $arg > gpurun_out/tune_ab.json
    python - <<PY
import json
r = json.loads(open("gpurun_out/tune_ab.json").read().strip().splitlines()[-1]); print("$kv", r["value"], r["ms_per_step"])
PY
  done
done
