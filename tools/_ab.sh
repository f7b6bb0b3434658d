for rep in 1 2; do
for e in "X=1" "NSID_W3_MIN_TILES=32" "NSID_W3_MIN_TILES=16" "NSID_WGRAD_RECT=0" "NSID_W3_MIN_TILES=100000"; do
  env $e python bench.py --no-cpu-baseline --no-roofline 2>&1 >/dev/null | grep timed | sed "s/^/$e /"
done; done
