#!/bin/bash
# Gate of the streaming weight gradient (tools/variants/wgrad_stream.hip: hook it into the library first, see its header): its parity tests, then cold-operand timings against the staged forms
# over the workgroup target. Run on the GPU box: tools/gpu_wgs_gate.sh [outdir]
set -o pipefail
OUT=${1:-gpurun_out/r05_wgs}
mkdir -p $OUT
timeout -k 10 600 python -m pytest tools/variants/wgrad_stream_test.py -x -q > $OUT/pytest.log 2>&1
rc=$?
echo "pytest rc=$rc" | tee -a $OUT/pytest.log
tail -5 $OUT/pytest.log
[ $rc -ne 0 ] && exit $rc
SHAPES="16384x256x1024x1,16384x1024x256x1,16384x256x512x1,16384x256x256x1,16384x128x128x4,8192x512x2048x1,8192x2048x512x1,8192x512x1024x1,8192x512x512x1,8192x256x256x4,32768x128x512x1,32768x512x128x1,32768x128x256x1,32768x128x128x1"
run() {   # name, tune flags...
  name=$1; shift
  timeout -k 10 300 python tools/gemm_bench.py --cold --only bwd_weight --reps 30 --shapes $SHAPES "$@" > $OUT/$name.txt 2>&1 || exit 1
}
run staged --tune wgrad_stream=0
for wg in 64 128 256; do
  run wgs${wg}_s3 --tune wgs_wgs=$wg --tune wgs_slots=3
done
run wgs128_s4 --tune wgs_wgs=128 --tune wgs_slots=4
run aff_staged --aff --tune wgrad_stream=0
run aff_wgs128 --aff --tune wgs_wgs=128
for f in staged wgs64_s3 wgs128_s3 wgs256_s3 wgs128_s4 aff_staged aff_wgs128; do echo "== $f"; grep "M=" $OUT/$f.txt; done
