import numpy as np, sys
a, b = np.load(sys.argv[1]), np.load(sys.argv[2])
for k in a.files:
    d = (a[k] != b[k])
    print(k, a[k].shape, "mismatching entries", int(d.sum()), "rows", int(d.reshape(-1, d.shape[-1]).any(1).sum()))
    if d.any():
        rows = np.argwhere(d.reshape(-1, d.shape[-1]).any(1))[:3, 0]
        for r in rows:
            print("   row", r, a[k].reshape(-1, d.shape[-1])[r], b[k].reshape(-1, d.shape[-1])[r])
