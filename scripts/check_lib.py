"""GPU helper: correctness of an experiment build (FINCFLOW_LIB=...; c3 kernels only): round trip, agreement with the strict
kernel, and every orientation, at shapes that take the variant under test.  Prints one line per shape; exit 1 on mismatch."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from fincflow_amd import FastFlowUnit, _lib, ops
dev = torch.device("cuda:0")
bad = 0
for (B, H, W) in ((256, 64, 64), (3, 64, 64), (129, 16, 32), (130, 40, 48), (2, 33, 16), (131, 20, 80)):
    torch.manual_seed(B + H)
    unit = FastFlowUnit(96, 96, 3).to(dev)
    x = torch.randn(B, 96, H, W, device=dev)
    with torch.no_grad():
        z, _ = unit(x)
        xr = unit.reverse(z)
        wc = unit._cache.w_canon
        xs = ops.finc_inverse(z, wc, algo="strict") if B * H * W <= 131 * 20 * 80 else None
    e1 = float((xr - x).abs().max() / x.abs().max())
    e2 = float((xr - xs).abs().max() / xs.abs().max()) if xs is not None else -1.0
    v = _lib.inverse_variant(B, 4, 24, H, W, 3, 3)
    ok = e1 <= 1e-5 and e2 <= 1e-5 and bool(torch.isfinite(xr).all())
    bad += not ok
    print(f"B{B} {H}x{W}: variant {v['nw']}w sec={v['sec']} round-trip {e1:.2e} vs strict {e2:.2e} {'ok' if ok else 'MISMATCH'}", flush=True)
sys.exit(1 if bad else 0)
