#!/bin/bash
# step time against clips per launch: (batch, overlap) grid of bench.py; one JSON line each into gpurun_out/batch_scaling/
# (answers "would launching both views as ONE batch of 2B clips beat two parallel branches of B?":
#  compare  t(2B, one stream)/2  with  t(B, two streams))
out=gpurun_out/batch_scaling; mkdir -p $out
for b in 128 256 512 1024; do
  for ov in "" "--no-overlap"; do
    tag="b${b}${ov:+_single}"
    python bench.py --batch $b $ov --no-cpu-baseline --no-roofline --no-other --steps 20 --warmup 3 > $out/$tag.json 2>$out/$tag.err || exit 1
    python - "$out/$tag.json" "$tag" <<'PY'
import json,sys
d=json.load(open(sys.argv[1])); print(sys.argv[2], "ms/step", d["ms_per_step"], "clips/s", d["value"])
PY
  done
done
