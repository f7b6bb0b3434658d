#!/bin/bash
# one-box A/B over library builds AND flags: tools/gpu_ab_lib.sh NAME REPS "label|libsuffix|bench flags" ...
name=$1; reps=$2; shift 2
out=gpurun_out/$name; mkdir -p $out
for rep in $(seq 1 $reps); do
  for cfg in "$@"; do
    IFS='|' read -r tag lib flags <<< "$cfg"
    libenv=""; [ -n "$lib" ] && libenv="$PWD/neuralsampleid_amd/libnsid_hip_$lib.so"
    NSID_LIB=$libenv timeout -k 10 200 python bench.py --no-cpu-baseline --no-roofline --no-other --steps 40 $flags > $out/$tag.$rep.json 2> $out/$tag.$rep.err || { tail -5 $out/$tag.$rep.err; exit 1; }
    python - $out/$tag.$rep.json "$tag" <<'PY'
import json,sys
d=json.load(open(sys.argv[1])); print(f"{sys.argv[2]:28s} {d['ms_per_step']} ms/step", flush=True)
PY
  done
done
