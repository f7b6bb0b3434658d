for rep in 1 2; do
  for w8 in 1024 0 512 2048; do
    NSID_W8_MIN=$w8 python bench.py --no-cpu-baseline --no-roofline 2>&1 >/dev/null | grep timed | sed "s/^/w8=$w8 /"
  done
done
for w8 in 1024 0 512; do
  NSID_W8_MIN=$w8 python bench.py --mode infer --no-cpu-baseline --no-roofline 2>/dev/null | python -c "import sys,json; d=json.load(sys.stdin); print('infer w8=$w8', d['value'])"
  NSID_W8_MIN=$w8 python bench.py --deep --no-cpu-baseline --no-roofline 2>&1 >/dev/null | grep timed | sed "s/^/deep w8=$w8 /"
done
