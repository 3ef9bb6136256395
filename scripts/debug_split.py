"""GPU helper: which entries of the output does the inverse leave untouched (sentinel fill), and how wrong are the written ones."""
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
out = torch.full_like(z, 12345.0)
ops.finc_inverse(z, wc, G, orient, algo="auto", out=out)
s = ops.finc_inverse(z, wc, G, orient, algo="strict")
o = out.reshape(B, G, C // G, H, W)
sr = s.reshape(B, G, C // G, H, W)
for g in range(G):
    untouched = (o[0, g] == 12345.0)
    wrong = ((o[0, g] - sr[0, g]).abs() > 1e-3) & ~untouched
    print("group", g, "untouched", int(untouched.sum()), "wrong-but-written", int(wrong.sum()))
    if untouched.any():
        idx = untouched.nonzero()
        print("   untouched channels", sorted(set(idx[:, 0].tolist()))[:30], "rows", sorted(set(idx[:, 1].tolist()))[:20])
    if wrong.any():
        idx = wrong.nonzero()
        print("   wrong channels", sorted(set(idx[:, 0].tolist()))[:30], "rows", sorted(set(idx[:, 1].tolist()))[:20], "first", idx[0].tolist(),
              float(o[0, g][tuple(idx[0].tolist())]), float(sr[0, g][tuple(idx[0].tolist())]))
