"""GPU helper: random problem sets that take the band split of the role-split inverse (finc_split.hip, BSP) against the strict
kernel, then a soak of back-to-back launches (alternating shapes, so that slots and launch numbers of the progress words turn
over), all waits counted.      python scripts/stress_bands.py [cases] [soak launches]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from fincflow_amd import FastFlowUnit, _lib, ops
dev = torch.device("cuda:0")
cases = int(sys.argv[1]) if len(sys.argv) > 1 else 120
soak = int(sys.argv[2]) if len(sys.argv) > 2 else 4000
rng = np.random.default_rng(11)
bad = taken = 0
worst = 0.0
for case in range(cases):
    K = int(rng.choice([3, 3, 3, 2]))
    Cq = int(rng.choice([3, 4, 8, 12, 16, 22, 24, 28, 32]))
    C = 4 * Cq
    B = int(rng.integers(1, 33))
    H = int(rng.integers(17, 140))
    W = int(rng.choice([64, 64, 68, 72]))
    v = _lib.inverse_variant(B, 4, Cq, H, W, K, K)
    if v is None or v["sec"] != 4 or v["workgroups"] != ((H + 15) // 16) * B * 4:
        continue
    taken += 1
    torch.manual_seed(case)
    unit = FastFlowUnit(C, C, K)
    with torch.no_grad():
        for m in (unit.conv_tl, unit.conv_tr, unit.conv_bl, unit.conv_br):
            m.conv.weight.mul_(1 - (1 - min(1.0, (24.0 / Cq) ** 0.5) * (0.7 if H > 64 else 1.0)) * m.get_mask())
    unit = unit.to(dev)
    x = torch.randn(B, C, H, W, device=dev)
    with torch.no_grad():
        z, _ = unit(x)
        got = unit.reverse(z)
        strict = ops.finc_inverse(z, unit._cache.w_canon, algo="strict")
    torch.cuda.synchronize()
    if _lib.fault_pending():
        bad += 1
        print(f"TIMEOUT case {case}: B{B} C{C} (Cq {Cq}) {H}x{W} k{K}: a progress wait gave up", flush=True)
        _lib.clear_fault()
        continue
    e = float((got - strict).abs().max() / strict.abs().max())
    worst = max(worst, e)
    if not (e <= 1e-5) or not bool(torch.isfinite(got).all()):
        bad += 1
        print(f"MISMATCH case {case}: B{B} C{C} {H}x{W} k{K}: {e:.2e}", flush=True)
print(f"{taken} band-split cases of {cases}, worst vs strict {worst:.2e}, mismatches {bad}, timeouts {_lib.hlp_timeouts()}", flush=True)
units = []
for (B, C, H, W) in ((32, 96, 64, 64), (5, 48, 100, 64), (16, 96, 33, 72)):
    u = FastFlowUnit(C, C, 3).to(dev)
    xx = torch.randn(B, C, H, W, device=dev)
    with torch.no_grad():
        zz, _ = u(xx)
        ref = u.reverse(zz).clone()
    units.append((u, zz, ref))
t0 = time.perf_counter()
with torch.no_grad():
    for i in range(soak):
        u, zz, ref = units[i % 3]
        out = u.reverse(zz)
        if i % 500 == 499:
            torch.cuda.synchronize()
            if not torch.equal(out, ref):
                bad += 1
                print("soak MISMATCH at", i, flush=True)
torch.cuda.synchronize()
print(f"soak {soak} launches in {time.perf_counter() - t0:.1f} s, timeouts {_lib.hlp_timeouts()}, fault {_lib.fault_pending()}", flush=True)
sys.exit(1 if bad or _lib.hlp_timeouts() else 0)
