#!/bin/bash
# One-box A/B of the whole training step over tuning keys of the kernel library (include/nsid.h nsid_set_tuning): every config
# ("key=value [key=value ...]" or "base"; a leading "lib=<path>" selects an alternative build through NSID_LIB) is timed REPS
# times, interleaved, `bench.py --steps 40` each.
#   tools/knob_sweep.sh out_dir 2 "base" "w3_wgs=128" "mr_split=1" "lib=neuralsampleid_amd/libnsid_hip_alt.so"
out=gpurun_out/$1; reps=$2; shift 2
mkdir -p $out
for rep in $(seq 1 $reps); do
  i=0
  for cfg in "$@"; do
    i=$((i+1))
    tune=""; libenv=""
    if [ "$cfg" != base ]; then
      for kv in $cfg; do
        case $kv in lib=*) libenv="${kv#lib=}" ;; *) tune="$tune --tune $kv" ;; esac
      done
    fi
    NSID_LIB=$libenv python bench.py --no-cpu-baseline --no-roofline --no-other --steps 40 $tune $BENCH_FLAGS > $out/c$i.$rep.json 2>$out/c$i.$rep.err
    python - "$out/c$i.$rep.json" "$cfg" <<'PY'
import json,sys
d=json.load(open(sys.argv[1])); print(f"{sys.argv[2]:60s} {d['ms_per_step']}", flush=True)
PY
  done
done
