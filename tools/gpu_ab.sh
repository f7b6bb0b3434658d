#!/bin/bash
# one-box A/B of the whole training step: tools/gpu_ab.sh NAME REPS "label:bench flags" ...   (interleaved, --steps 40)
name=$1; reps=$2; shift 2
out=gpurun_out/$name; mkdir -p $out
for rep in $(seq 1 $reps); do
  for cfg in "$@"; do
    tag=${cfg%%:*}; flags=${cfg#*:}
    timeout -k 10 200 python bench.py --no-cpu-baseline --no-roofline --no-other --steps 40 $flags > $out/$tag.$rep.json 2> $out/$tag.$rep.err || { tail -5 $out/$tag.$rep.err; exit 1; }
    python - $out/$tag.$rep.json "$tag" <<'PY'
import json,sys
d=json.load(open(sys.argv[1])); print(f"{sys.argv[2]:28s} {d['ms_per_step']} ms/step  loss {d['config']['final_loss']}", flush=True)
PY
  done
done
