#!/bin/bash
# Gate of round 5 (VERDICT r4 task 1): the weight-stationary streaming GEMMs (csrc/wsgemm.hip) against the tile kernels they replace,
# cold operands (tools/gemm_bench.py --cold), forward with statistics. Run on the GPU box: tools/gpu_ws_gate.sh [outdir]
set -o pipefail
OUT=${1:-gpurun_out/r05_gate}
mkdir -p $OUT
python -m pytest tests/test_wsgemm_gpu.py -x -q > $OUT/pytest.log 2>&1
echo "pytest rc=$?" | tee -a $OUT/pytest.log
tail -5 $OUT/pytest.log
SHAPES="65536x64x64x1,65536x32x32x4,65536x256x64x1,32768x128x128x1,32768x64x64x4,16384x128x128x4"
AFFSHAPES="65536x64x128x1,65536x64x256x1,32768x128x256x1"
for rep in 1 2; do
  for ws in 0 1; do
    python tools/gemm_bench.py --cold --only fwd --reps 40 --shapes $SHAPES --tune ws_gemm=$ws > $OUT/fwd_ws${ws}_r${rep}.txt 2>&1
    python tools/gemm_bench.py --cold --only fwd --reps 40 --shapes $AFFSHAPES --aff --tune ws_gemm=$ws >> $OUT/fwd_ws${ws}_r${rep}.txt 2>&1
  done
done
for f in $OUT/fwd_ws*.txt; do echo "== $f"; cat $f; done
