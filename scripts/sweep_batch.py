"""GPU helper: inverse/forward launch time vs batch size (one FastFlowUnit), to see occupancy steps."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from fincflow_amd import FastFlowUnit
dev = torch.device("cuda:0")
C, H, W, K = (int(a) for a in sys.argv[1:5]) if len(sys.argv) > 4 else (96, 64, 64, 3)
unit = FastFlowUnit(C, C, K).to(dev)
for B in (16, 32, 64, 96, 128, 160, 192, 224, 256, 320, 384, 512):
    x = torch.randn(B, C, H, W, device=dev)
    with torch.no_grad():
        z, _ = unit(x)
        out = torch.empty_like(z)
        def run_inv(): unit._cache.inverse(z, unit._weights(), 4, 0xE4, out=out)
        def run_fwd(): unit(x)
        res = []
        for fn in (run_inv, run_fwd):
            for _ in range(3): fn()
            torch.cuda.synchronize()
            a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            a.record()
            for _ in range(10): fn()
            b.record(); torch.cuda.synchronize()
            res.append(a.elapsed_time(b) / 10)
    print(f"B={B:4d} WGs={4*B:5d} inv {res[0]*1e3:8.1f} us  fwd {res[1]*1e3:8.1f} us   inv img/s {B/res[0]*1e3:10.0f}", flush=True)
