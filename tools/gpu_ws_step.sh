#!/bin/bash
# round 5: ws tests + micro gate + one-box A/B of the whole step (ws_gemm 0 / 1 / 3) + the whole GPU suite
set -o pipefail
OUT=${1:-gpurun_out/r05_step}
mkdir -p $OUT
python -m pytest tests/test_wsgemm_gpu.py -x -q > $OUT/pytest_ws.log 2>&1; echo "ws pytest rc=$?"; tail -12 $OUT/pytest_ws.log
for ws in 3; do
python tools/gemm_bench.py --cold --only fwd,bwd_data --reps 40 --tune ws_gemm=$ws --shapes 65536x64x64x1,65536x32x32x4,65536x64x128x1,65536x256x64x1,65536x64x256x1,32768x128x128x1,32768x64x64x4,32768x128x256x1,16384x128x128x4 > $OUT/micro_ws$ws.txt 2>&1; grep "M=" $OUT/micro_ws$ws.txt
done
bash tools/gpu_ab.sh $(basename $OUT)/ab 3 "base:--tune ws_gemm=0" "ws3:--tune ws_gemm=3" "ws7:--tune ws_gemm=7" 2>&1 | tee $OUT/ab.txt
timeout -k 10 900 python -m pytest tests -m gpu -q > $OUT/pytest_all.log 2>&1; echo "all pytest rc=$?"; tail -25 $OUT/pytest_all.log
