"""GPU helper: where does the helper-wave kernel differ from the strict kernel?"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from fincflow_amd import FastFlowUnit, _lib, ops
dev = torch.device("cuda:0")
torch.manual_seed(0)
B, H, W = int(os.environ.get("DBG_B", "132")), int(os.environ.get("DBG_H", "32")), int(os.environ.get("DBG_W", "32"))
unit = FastFlowUnit(96, 96, 3).to(dev)
x = torch.randn(B, 96, H, W, device=dev)
import time
with torch.no_grad():
    z, _ = unit(x)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    xr = unit.reverse(z)
    torch.cuda.synchronize(); print("first call %.1f ms" % ((time.perf_counter() - t0) * 1e3), flush=True)
    t0 = time.perf_counter()
    for _ in range(5): xr = unit.reverse(z)
    torch.cuda.synchronize(); print("steady %.3f ms per call" % ((time.perf_counter() - t0) * 1e3 / 5), flush=True)
print(_lib.inverse_variant(B, 4, 24, H, W, 3, 3))
bad = ~torch.isfinite(xr) | ((xr - x).abs() > 1e-3)
print("bad elements", int(bad.sum()), "of", bad.numel())
if bad.any():
    idx = bad.nonzero()
    print("first bad (b,c,h,w):", idx[:5].tolist())
    print("bad per image (first 8):", bad.flatten(1).sum(1)[:8].tolist())
    print("bad per group of image 0:", [int(bad[0, g*24:(g+1)*24].sum()) for g in range(4)])
    b0 = bad[0, :24]
    print("bad per row (group 0):", b0.sum((0, 2)).tolist())
    print("bad per col (group 0):", b0.sum((0, 1)).tolist())
    print("bad per channel (group 0):", b0.sum((1, 2)).tolist())
    print("nan count", int(torch.isnan(xr).sum()))
