#!/bin/bash
# the captured training step under HIP runtime environment settings (one box, interleaved): tools/gpu_env_scan.sh "VAR=VAL" ...
out=gpurun_out/env_scan; mkdir -p $out
for rep in 1 2; do
  for kv in "" "$@"; do
    tag=${kv:-default}
    env $kv timeout -k 10 200 python bench.py --no-cpu-baseline --no-roofline --no-other --steps 40 > $out/r.json 2> $out/r.err || { echo "$tag FAILED: $(tail -1 $out/r.err)"; continue; }
    python - "$tag" <<'PY'
import json,sys
d=json.load(open("gpurun_out/env_scan/r.json")); print(f"{sys.argv[1]:48s} {d['ms_per_step']}", flush=True)
PY
  done
done
