mkdir -p gpurun_out/r02g
for cfg in "1,0" "2,2,1000000,0" "2,2,1000000,1" "1,4,1000000,0"; do
  NSID_GEMM_DEEP=$cfg NSID_GEMM_DEEP_KINDS=1 python bench.py --mode infer --no-cpu-baseline --no-roofline 2>/dev/null | python -c "import sys,json; d=json.load(sys.stdin); print('$cfg', d['value'], d['ms_per_step'])" >> gpurun_out/r02g/infer_ab.txt
  NSID_TALL_MIN=100000000 NSID_GEMM_DEEP=$cfg NSID_GEMM_DEEP_KINDS=1 python bench.py --mode infer --no-cpu-baseline --no-roofline 2>/dev/null | python -c "import sys,json; d=json.load(sys.stdin); print('notall $cfg', d['value'], d['ms_per_step'])" >> gpurun_out/r02g/infer_ab.txt
done
cat gpurun_out/r02g/infer_ab.txt
