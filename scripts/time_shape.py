"""GPU helper: time the inverse / forward of one FastFlowUnit shape.  Usage: time_shape.py B C H W K [std]"""
import os, sys, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from fincflow_amd import FastFlowUnit
B, C, H, W, K = map(int, sys.argv[1:6])
std = float(sys.argv[6]) if len(sys.argv) > 6 else 0.05
dev = torch.device("cuda:0"); torch.manual_seed(0)
unit = FastFlowUnit(C, C, K).to(dev)
if std != 0.05:
    with torch.no_grad():
        for m in (unit.conv_tl, unit.conv_tr, unit.conv_bl, unit.conv_br):
            w = m.conv.weight; mask = m.mask.to(dev)
            w.mul_(1 - mask + mask * (std / 0.05))
x = torch.randn(B, C, H, W, device=dev)
with torch.no_grad():
    z, _ = unit(x); o = torch.empty_like(z)
    xr = unit.reverse(z)
    err = ((xr - x).abs().max() / x.abs().max()).item()
    res = []
    for fn in (lambda: unit._cache.inverse(z, unit._weights(), 4, 0xE4, out=o), lambda: unit._cache.forward(x, unit._weights(), 4, 0xE4, out=o)):
        t_end = time.perf_counter() + 0.3
        while time.perf_counter() < t_end:
            for _ in range(10): fn()
            torch.cuda.synchronize()
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        for _ in range(100): fn()
        b.record(); torch.cuda.synchronize()
        res.append(a.elapsed_time(b) / 100 * 1e3)
print("B%d C%d %dx%d k%d: inverse %.1f us  forward %.1f us  round-trip rel err %.2e" % (B, C, H, W, K, res[0], res[1], err))
