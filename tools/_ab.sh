for sp in 1 2 4; do echo "== channel split $sp"; NSID_MR_SPLIT=$sp python tools/op_bench.py --ops mr 2>/dev/null; done
python -m pytest tests -m gpu -q -k "mr or block or graph_kernels" 2>&1 | tail -2
NSID_MR_SPLIT=4 python -m pytest tests -m gpu -q -k "mr or block or graph_kernels" 2>&1 | tail -1
for rep in 1 2; do for sp in 1 2 4; do NSID_MR_SPLIT=$sp python bench.py --no-cpu-baseline --no-roofline 2>&1 >/dev/null | grep timed | sed "s/^/split=$sp /"; done; done
