#!/bin/bash
# round 5: full GPU suite + extraction A/B (raw bf16 kNN on / off) + the default bench line
set -o pipefail
OUT=${1:-gpurun_out/r05b}
mkdir -p $OUT
timeout -k 10 900 python -m pytest tests -m gpu -q > $OUT/pytest_all.log 2>&1; echo "all pytest rc=$?"; tail -12 $OUT/pytest_all.log
for rep in 1 2; do
  for cfg in "raw1:--tune knn_raw16=1" "raw0:--tune knn_raw16=0"; do
    tag=${cfg%%:*}; flags=${cfg#*:}
    python bench.py --mode infer --clips 100000 --no-cpu-baseline --no-roofline $flags > $OUT/infer_$tag.$rep.json 2> $OUT/infer_$tag.$rep.err
    python - $OUT/infer_$tag.$rep.json $tag <<'PY'
import json,sys
d=json.load(open(sys.argv[1])); print(f"{sys.argv[2]:8s} {d['value']} clips/s  {d['ms_per_step']} ms per micro-batch", flush=True)
PY
  done
done
