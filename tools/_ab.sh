python -m pytest tests/test_ops_gpu.py -q -k "downsample" 2>&1 | tail -15
python -m pytest tests -m gpu -q 2>&1 | tail -3
for rep in 1 2; do
  python bench.py --no-cpu-baseline --no-roofline 2>&1 >/dev/null | grep timed
done
