"""GPU helper: time the role-split inverse on band-split shapes (env FINC_SPLIT_BANDS / FINC_BSP_MODE select the form)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from fincflow_amd import FastFlowUnit, _lib
dev = torch.device("cuda:0")
out = []
for (B, C, H, W) in ((32, 96, 64, 64), (4, 96, 64, 64), (32, 96, 32, 64), (32, 96, 48, 64), (32, 96, 128, 64), (16, 48, 64, 64)):
    torch.manual_seed(0)
    unit = FastFlowUnit(C, C, 3).to(dev)
    x = torch.randn(B, C, H, W, device=dev)
    with torch.no_grad():
        z, _ = unit(x)
        o = torch.empty_like(z)
        fn = lambda: unit._cache.inverse(z, unit._weights(), 4, 0xE4, out=o)
        t_end = time.perf_counter() + 0.2
        while time.perf_counter() < t_end:
            for _ in range(10): fn()
            torch.cuda.synchronize()
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        for _ in range(100): fn()
        b.record(); torch.cuda.synchronize()
        err = float((o - x).abs().max() / x.abs().max())
    v = _lib.inverse_variant(B, 4, C // 4, H, W, 3, 3)
    out.append(f"{H}x{W} B{B} C{C}: {a.elapsed_time(b) * 10:.1f} us (wg/problem {v['workgroups'] // (B * 4)}, err {err:.1e})")
print(os.environ.get("FINC_SPLIT_BANDS", "-"), os.environ.get("FINC_BSP_MODE", "-"), "|", " | ".join(out), "| timeouts", _lib.hlp_timeouts(), flush=True)
