"""GPU helper: time inverse+forward for B in (16, 256) with the library selected by FINCFLOW_LIB (no checks)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from fincflow_amd import FastFlowUnit
dev = torch.device("cuda:0")
C, H, W, K = 96, 64, 64, 3
unit = FastFlowUnit(C, C, K).to(dev)
out = []
for B in (16, 256):
    x = torch.randn(B, C, H, W, device=dev)
    with torch.no_grad():
        z, _ = unit(x)
        o = torch.empty_like(z)
        for fn in (lambda: unit._cache.inverse(z, unit._weights(), 4, 0xE4, out=o), lambda: unit(x)):
            for _ in range(3): fn()
            torch.cuda.synchronize()
            a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            a.record()
            for _ in range(10): fn()
            b.record(); torch.cuda.synchronize()
            out.append(a.elapsed_time(b) / 10 * 1e3)
print(os.environ.get("FINCFLOW_LIB", "default").split("/")[-1],
      "B16 inv %.0f fwd %.0f | B256 inv %.0f fwd %.0f us" % tuple(out), flush=True)
