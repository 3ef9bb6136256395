"""GPU helper: the band split of the role-split inverse (finc_split.hip, BSP) against the strict kernel and the chained form:
shapes with 2..5 bands, partial last bands, every flip (the four groups of a FastFlowUnit), repeated launches (the epoch of
the progress words advances on the device), a captured launch replayed, and the timing of both forms.
    python scripts/check_bands.py            (FINC_SPLIT_BANDS=0 in a second process gives the chained timings)"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from fincflow_amd import FastFlowUnit, _lib, ops
dev = torch.device("cuda:0")
bad = 0
for (B, C, H, W) in ((32, 96, 64, 64), (4, 96, 64, 64), (16, 48, 32, 32), (3, 96, 48, 64), (2, 96, 33, 64), (8, 48, 17, 72), (2, 96, 33, 32),
                     (5, 16, 80, 32), (1, 128, 40, 72), (7, 96, 31, 64)):
    torch.manual_seed(B + H)
    unit = FastFlowUnit(C, C, 3).to(dev)
    x = torch.randn(B, C, H, W, device=dev)
    v = _lib.inverse_variant(B, 4, C // 4, H, W, 3, 3)
    with torch.no_grad():
        z, _ = unit(x)
        xr = unit.reverse(z)
        xs = ops.finc_inverse(z, unit._cache.w_canon, algo="strict")
        again = [unit.reverse(z) for _ in range(5)]
    torch.cuda.synchronize()
    e1 = float((xr - x).abs().max() / x.abs().max())
    e2 = float((xr - xs).abs().max() / xs.abs().max())
    same = all(torch.equal(a, xr) for a in again)
    ok = e1 <= 1e-5 and e2 <= 1e-5 and same and bool(torch.isfinite(xr).all())
    bad += not ok
    nwg = v["workgroups"] // (B * 4) if v else 0
    print(f"B{B} C{C} {H}x{W}: form {v['sec'] if v else None} workgroups/problem {nwg} round-trip {e1:.2e} vs strict {e2:.2e} "
          f"repeat-identical {same} {'ok' if ok else 'MISMATCH'}", flush=True)
# a captured launch, replayed (the epoch lives on the device)
unit = FastFlowUnit(96, 96, 3).to(dev)
x = torch.randn(32, 96, 64, 64, device=dev)
with torch.no_grad():
    z, _ = unit(x)
    ref = unit.reverse(z)
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        out = unit.reverse(z)
    for _ in range(4):
        out.zero_()
        g.replay()
        torch.cuda.synchronize()
        if not torch.equal(out, ref):
            bad += 1
            print("graph replay MISMATCH")
print("graph replays ok" if not bad else "FAILED", "| timeouts", _lib.hlp_timeouts(), "| fault", _lib.fault_pending())
for (B, C, H, W) in ((32, 96, 64, 64), (16, 96, 64, 64), (4, 96, 64, 64), (16, 48, 32, 32), (32, 96, 48, 64), (32, 96, 128, 64)):
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
    v = _lib.inverse_variant(B, 4, C // 4, H, W, 3, 3)
    print(f"time B{B} C{C} {H}x{W}: {a.elapsed_time(b) * 10:.1f} us  workgroups/problem {v['workgroups'] // (B * 4)}", flush=True)
sys.exit(1 if bad else 0)
