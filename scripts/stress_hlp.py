"""GPU helper: many launches of the inverse at several shapes; prints the slowest launch per shape and checks every result
bit for bit against the first one of its shape (and the round trip)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from fincflow_amd import FastFlowUnit, _lib
dev = torch.device("cuda:0"); torch.manual_seed(0)
n = int(sys.argv[1]) if len(sys.argv) > 1 else 500
shapes = [(256, 96, 64, 64), (129, 96, 16, 32), (130, 96, 40, 48), (132, 96, 32, 32), (64, 96, 128, 16), (257, 48, 32, 32), (130, 96, 20, 80), (512, 64, 16, 16)]
tot_bad = 0
for (B, C, H, W) in shapes:
    unit = FastFlowUnit(C, C, 3).to(dev)
    x = torch.randn(B, C, H, W, device=dev)
    with torch.no_grad():
        z, _ = unit(x)
        ref = unit.reverse(z).clone()
        rt = float((ref - x).abs().max() / x.abs().max())
        worst, bad, t0 = 0.0, 0, time.perf_counter()
        for i in range(n):
            a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            a.record(); out = unit.reverse(z); b.record(); torch.cuda.synchronize()
            worst = max(worst, a.elapsed_time(b))
            bad += 0 if torch.equal(out, ref) else 1
    v = _lib.inverse_variant(B, 4, C // 4, H, W, 3, 3)
    tot_bad += bad + (rt > 1e-5)
    print(f"B{B} C{C} {H}x{W} form {v['sec'] if v else None} nw {v['nw'] if v else None}: {n} launches {time.perf_counter()-t0:.1f} s, worst {worst:.3f} ms, "
          f"mismatches {bad}, round trip {rt:.1e}", flush=True)
sys.exit(1 if tot_bad else 0)
