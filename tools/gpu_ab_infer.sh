#!/bin/bash
# one-box A/B of forward-only extraction: tools/gpu_ab_infer.sh NAME REPS "label:bench flags" ...
name=$1; reps=$2; shift 2
out=gpurun_out/$name; mkdir -p $out
for rep in $(seq 1 $reps); do
  for cfg in "$@"; do
    tag=${cfg%%:*}; flags=${cfg#*:}
    timeout -k 10 200 python bench.py --mode infer --no-cpu-baseline --no-roofline $flags > $out/$tag.$rep.json 2> $out/$tag.$rep.err || { tail -5 $out/$tag.$rep.err; exit 1; }
    python - $out/$tag.$rep.json "$tag" <<'PY'
import json,sys
d=json.load(open(sys.argv[1])); print(f"{sys.argv[2]:28s} {d['value']:10.0f} clips/s  {d['ms_per_step']} ms per micro-batch", flush=True)
PY
  done
done
