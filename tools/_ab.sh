mkdir -p gpurun_out/r02h
python -m pytest tests/test_timed_arithmetic_gpu.py -q -k "shadows or eval_after" 2>&1 | tail -2
for rep in 1 2; do
for cfg in "2,2,512,0:5" "1,0:0" "2,2,512,1:5"; do
  d=${cfg%%:*}; k=${cfg##*:}
  NSID_GEMM_DEEP=$d NSID_GEMM_DEEP_KINDS=$k python bench.py --no-cpu-baseline --no-roofline 2>&1 >/dev/null | grep timed | sed "s/^/deep=$d kinds=$k /" >> gpurun_out/r02h/ab.txt
done
done
cat gpurun_out/r02h/ab.txt
