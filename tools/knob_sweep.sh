#!/bin/bash
# One-box A/B of the whole training step over environment knobs of the kernel library: every line of the config list
# ("NAME=VALUE [NAME=VALUE ...]" or "base") is timed REPS times, interleaved, `bench.py --steps 40` each.
#   tools/knob_sweep.sh out_dir 2 "base" "NSID_W3_WGS=128" "NSID_MR_SPLIT=1"
out=gpurun_out/$1; reps=$2; shift 2
mkdir -p $out
for rep in $(seq 1 $reps); do
  i=0
  for cfg in "$@"; do
    i=$((i+1))
    envs=""; [ "$cfg" != base ] && envs="$cfg"
    env $envs python bench.py --no-cpu-baseline --no-roofline --steps 40 $BENCH_FLAGS > $out/c$i.$rep.json 2>/dev/null
    python - "$out/c$i.$rep.json" "$cfg" <<'PY'
import json,sys
d=json.load(open(sys.argv[1])); print(f"{sys.argv[2]:60s} {d['ms_per_step']}", flush=True)
PY
  done
done
