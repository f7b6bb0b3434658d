#!/usr/bin/env python3
"""Marginal cost of a kernel family INSIDE the two-stream hipGraph step: bench.py with the family's launches skipped.

    python tools/ablate.py nsid_bn_bwd_apply,nsid_bn_bwd_finalize -- --no-cpu-baseline --no-roofline

The listed C-ABI entry points return without launching (the step's arithmetic is then garbage; the NaN guard may skip the
optimiser update): only the step time is meaningful. This lives in tools/ on purpose — the product call path
(neuralsampleid_amd/_lib.py) has no such switch. The patch is applied to `_lib.call` BEFORE any module binds it."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    if "--" not in sys.argv:
        raise SystemExit(__doc__)
    cut = sys.argv.index("--")
    skip = frozenset(n for a in sys.argv[1:cut] for n in a.split(",") if n)
    from neuralsampleid_amd import _lib
    real = _lib.call

    def call(name, *args):
        if name in skip:
            return None
        return real(name, *args)

    unknown = [n for n in skip if not hasattr(_lib.lib, n)]
    if unknown:
        raise SystemExit(f"not exported by libnsid_hip.so: {unknown}")
    _lib.call = call
    for n in skip:                       # entries that ops.py binds directly (they return 1 = "not this form" to ask for a fallback)
        setattr(_lib.lib, n, lambda *a: 0)
    sys.argv = [os.path.join(ROOT, "bench.py")] + sys.argv[cut + 1:]
    import bench
    print(f"[ablate] skipping {sorted(skip)}", file=sys.stderr)
    bench.main()


if __name__ == "__main__":
    main()
