#!/bin/bash
# forward-only extraction: kernel tests that cover the changed kernels, then one-box A/B of the 100 000-clip bench
out=gpurun_out/r04_infer; mkdir -p $out
export TMPDIR=/tmp
timeout -k 10 900 python -m pytest tests/test_ops_gpu.py tests/test_bf16_storage_gpu.py tests/test_e2e_gpu.py -x -q -k "knn or mrconv or eval or fingerprint or extract" > $out/tests.log 2>&1; rc=$?
tail -4 $out/tests.log
[ $rc -ne 0 ] && exit $rc
for rep in 1 2; do
  for cfg in "base:" "$@"; do
    tag=${cfg%%:*}; flags=${cfg#*:}
    timeout -k 10 300 python3 bench.py --mode infer --no-cpu-baseline --no-roofline $flags > $out/$tag.$rep.json 2> $out/$tag.$rep.err || { tail -5 $out/$tag.$rep.err; exit 1; }
    python3 -c "import json;d=json.load(open('$out/$tag.$rep.json'));print('$tag', d['value'], d['ms_per_step'])"
  done
done
rocprofv3 --kernel-trace --stats --output-format csv -d $out/stats -- python3 bench.py --mode infer --clips 20480 --infer-streams 1 --no-cpu-baseline --no-roofline > $out/prof.json 2> $out/prof.err
cp $(find $out/stats -name "*kernel_stats.csv" | head -1) $out/kernel_stats_infer.csv; rm -rf $out/stats
