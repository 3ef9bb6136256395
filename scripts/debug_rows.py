"""GPU helper: where an MFMA inverse variant goes wrong -- error per (group, row) against the strict kernel."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from fincflow_amd import _lib, ops
from oracle import oracle
dev = torch.device("cuda:0")
B, C, H, W, K = (int(a) for a in sys.argv[1:6]) if len(sys.argv) > 5 else (1, 96, 64, 64, 3)
G, orient = 4, 0xE4
ws = oracle.make_stored_weights(G, C // G, K, K, orient=orient, seed=1, std=0.05)
wc = ops.canonicalize(torch.from_numpy(ws).to(dev), G, orient)
z = torch.randn(B, C, H, W, device=dev)
print(_lib.inverse_variant(B, G, C // G, H, W, K, K))
a = ops.finc_inverse(z, wc, G, orient, algo="auto")
s = ops.finc_inverse(z, wc, G, orient, algo="strict")
d = (a - s).abs().reshape(B, G, C // G, H, W)
sc = float(s.abs().max())
for g in range(G):
    per_row = d[:, g].amax(dim=(0, 1, 3)) / sc
    print("group", g, " ".join(f"{v:.0e}" for v in per_row.tolist()))
    bad = (d[0, g].amax(dim=0) / sc > 1e-4).nonzero()
    if len(bad):
        print("   first bad (row, col):", bad[0].tolist(), " bad count", len(bad))
if len(sys.argv) > 6:
    g = int(sys.argv[6])
    m = (d[0, g].amax(dim=0) / sc > 1e-4)
    print("bad rows:", sorted(set(m.nonzero()[:, 0].tolist())))
    for r in sorted(set(m.nonzero()[:, 0].tolist()))[:6]:
        print(" row", r, "bad cols:", m[r].nonzero().flatten().tolist()[:40])
    ch = (d[0, g].amax(dim=(1, 2)) / sc)
    print("per channel:", " ".join(f"{v:.0e}" for v in ch.tolist()))
