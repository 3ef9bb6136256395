"""GPU helper: per-launch HIP-event times of the inverse over many launches -- mean, median, p99, max and the indices of the outliers
(a protocol that stalls now and then shows here, not in a mean).   python scripts/launch_outliers.py B C H W K [launches]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from fincflow_amd import FastFlowUnit, _lib
dev = torch.device("cuda:0")
B, C, H, W, K = (int(a) for a in sys.argv[1:6])
n = int(sys.argv[6]) if len(sys.argv) > 6 else 2000
torch.manual_seed(0)
unit = FastFlowUnit(C, C, K).to(dev)
x = torch.randn(B, C, H, W, device=dev)
with torch.no_grad():
    z, _ = unit(x)
    for _ in range(20): unit.reverse(z)
    torch.cuda.synchronize()
    evs = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(n)]
    for a, b in evs:
        a.record(); out = unit.reverse(z); b.record()
    torch.cuda.synchronize()
t = np.array([a.elapsed_time(b) for a, b in evs]) * 1e3
srt = np.sort(t)
print(f"C{C} {H}x{W} B={B}: {n} launches  mean {t.mean():.1f}  median {np.median(t):.1f}  p99 {srt[int(0.99 * n)]:.1f}  max {t.max():.1f} us; "
      f"launches over 3x the median: {[(int(i), round(float(t[i]), 1)) for i in np.nonzero(t > 3 * np.median(t))[0][:10]]}; "
      f"timeouts {_lib.hlp_timeouts()} fault {_lib.fault_pending()}; round trip {float((out - x).abs().max() / x.abs().max()):.1e}", flush=True)
