#!/bin/bash
# one-box A/B of the extraction bench over tuning sets: tools/gpu_infer_tune_ab.sh NAME REPS "label:flags" ...
name=$1; reps=$2; shift 2
out=gpurun_out/$name; mkdir -p $out
for rep in $(seq 1 $reps); do
  for cfg in "$@"; do
    tag=${cfg%%:*}; flags=${cfg#*:}
    python bench.py --mode infer --clips 100000 --no-cpu-baseline --no-roofline $flags > $out/$tag.$rep.json 2> $out/$tag.$rep.err || { tail -5 $out/$tag.$rep.err; exit 1; }
    python - $out/$tag.$rep.json $tag <<'PY'
import json,sys
d=json.load(open(sys.argv[1])); print(f"{sys.argv[2]:12s} {d['value']} clips/s  {d['ms_per_step']} ms per micro-batch", flush=True)
PY
  done
done
