"""GPU helper: one line per (Cq, K) of the instantiation tables -- forward and inverse time of a FastFlowUnit at a batch that
fills the chip (B*4 = 1024 problems) on 64x64 maps, as a fraction of the fp32 MFMA peak.  Looks for performance cliffs
(a spilling or under-occupied instantiation shows up as an outlier of its neighbours)."""
import os, sys, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from fincflow_amd import FastFlowUnit, _lib
dev = torch.device("cuda:0"); torch.manual_seed(0)
PEAK = 157.3e12
seen = set()
rows = []
for r in _lib.inverse_table():
    key = (r["cqp"], r["kh"], r["kw"])
    if key in seen or r["kh"] != r["kw"]:
        continue
    seen.add(key)
    Cq, K = r["cqp"], r["kh"]
    B, H, W = 256, 64, 64
    if Cq * K * K > 24 * 9 * 2:          # big banks: fewer images, same order of work
        B = 64
    C = 4 * Cq
    unit = FastFlowUnit(C, C, K).to(dev)
    std = (0.05 if K < 5 else 0.02) * min(1.0, (24.0 / Cq) ** 0.5)
    with torch.no_grad():
        for m in (unit.conv_tl, unit.conv_tr, unit.conv_bl, unit.conv_br):
            mask = torch.as_tensor(m.mask).to(dev)
            m.conv.weight.mul_(1 - mask + mask * (std / 0.05))
        x = torch.randn(B, C, H, W, device=dev)
        z, _ = unit(x); o = torch.empty_like(z)
        xr = unit.reverse(z)
        err = ((xr - x).abs().max() / x.abs().max()).item()
        res = []
        for fn in (lambda: unit._cache.inverse(z, unit._weights(), 4, 0xE4, out=o), lambda: unit._cache.forward(x, unit._weights(), 4, 0xE4, out=o)):
            t_end = time.perf_counter() + 0.25
            while time.perf_counter() < t_end:
                for _ in range(10): fn()
                torch.cuda.synchronize()
            torch.cuda.synchronize()
            a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            a.record()
            for _ in range(30): fn()
            b.record(); torch.cuda.synchronize()
            res.append(a.elapsed_time(b) / 30 * 1e-3)
    flops = 2.0 * B * C * H * W * K * K * Cq
    v = _lib.inverse_variant(B, 4, Cq, H, W, K, K)
    # (the forward's figure is DIRECT-EQUIVALENT flops over time: the Winograd forms execute 1/2 .. 2/3 of those multiplies, so it can
    # exceed 100 % -- it is a speed relative to the direct sum's roof, not a utilisation)
    print(f"Cq={Cq:3d} K={K} B={B:3d}: inverse {res[0]*1e6:8.1f} us = {flops/res[0]/PEAK:5.1%} of fp32 peak (nw {v['nw']} form {v['sec']}) | "
          f"forward {res[1]*1e6:8.1f} us = {flops/res[1]/PEAK:5.1%} direct-equivalent | round trip {err:.1e}", flush=True)
    del unit, x, z, o, xr
    torch.cuda.empty_cache()
