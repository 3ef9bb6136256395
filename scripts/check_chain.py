"""GPU helper: the short-step kernel (finc_chain.hip) against the strict reference-order kernel on a list of shapes, every
orientation in one call (FastFlowUnit's four corners), with the time per launch; untouched / wrong entries are located."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from fincflow_amd import _lib, ops
from oracle import oracle
dev = torch.device("cuda:0")
G, orient = 4, 0xE4
shapes = [(64, 48, 32, 32, 3), (16, 48, 32, 32, 3), (4, 48, 64, 64, 3), (8, 16, 16, 16, 3), (8, 32, 8, 8, 3), (8, 48, 4, 4, 3),
          (8, 64, 32, 32, 3), (8, 64, 32, 32, 2), (8, 12, 16, 16, 3), (8, 24, 8, 8, 3), (3, 40, 20, 24, 3), (5, 48, 36, 28, 3),
          (2, 48, 48, 64, 3), (2, 32, 17, 20, 2), (2, 16, 5, 12, 3), (64, 48, 32, 32, 3)]
if len(sys.argv) > 5:
    shapes = [tuple(int(a) for a in sys.argv[1:6])]
bad = 0
for (B, C, H, W, K) in shapes:
    ws = oracle.make_stored_weights(G, C // G, K, K, orient=orient, seed=1, std=0.05)
    wc = ops.canonicalize(torch.from_numpy(ws).to(dev), G, orient)
    torch.manual_seed(0)
    z = torch.randn(B, C, H, W, device=dev)
    out = torch.full_like(z, 12345.0)
    ops.finc_inverse(z, wc, G, orient, algo="auto", out=out)
    s = ops.finc_inverse(z, wc, G, orient, algo="strict")
    torch.cuda.synchronize()
    v = _lib.inverse_variant(B, G, C // G, H, W, K, K)
    err = float((out - s).abs().max() / s.abs().max())
    for _ in range(10): ops.finc_inverse(z, wc, G, orient, algo="auto", out=out)
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(50): ops.finc_inverse(z, wc, G, orient, algo="auto", out=out)
    b.record(); torch.cuda.synchronize()
    us = a.elapsed_time(b) / 50 * 1e3
    print(f"B{B} C{C} {H}x{W} k{K}: err {err:.2e}  {us:7.1f} us  variant {v}", flush=True)
    if not err < 1e-5:
        bad += 1
        o = out.reshape(B, G, C // G, H, W); sr = s.reshape(B, G, C // G, H, W)
        for g in range(G):
            untouched = (o[0, g] == 12345.0)
            wrong = ((o[0, g] - sr[0, g]).abs() > 1e-4 * float(s.abs().max())) & ~untouched
            print("   group", g, "untouched", int(untouched.sum()), "wrong-but-written", int(wrong.sum()))
            if wrong.any():
                idx = wrong.nonzero()
                print("      wrong channels", sorted(set(idx[:, 0].tolist()))[:16], "rows", sorted(set(idx[:, 1].tolist()))[:20], "cols",
                      sorted(set(idx[:, 2].tolist()))[:20], "first", idx[0].tolist(), float(o[0, g][tuple(idx[0].tolist())]),
                      float(sr[0, g][tuple(idx[0].tolist())]))
print("hlp timeouts", _lib.hlp_timeouts(), "bad", bad)
sys.exit(1 if bad else 0)
