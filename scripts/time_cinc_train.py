"""GPU helper: forward / reverse / training step of CINCFlowUnit (one 3x3 conv over all channels) at C = 96 and C = 64."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from fincflow_amd import CINCFlowUnit, _lib
dev = torch.device("cuda:0")
def timeit(fn, n=10):
    t_end = time.perf_counter() + 0.3
    while time.perf_counter() < t_end:
        fn(); torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(n): fn()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / n
for (B, C, H, W) in ((256, 96, 64, 64), (256, 96, 32, 32), (256, 64, 64, 64), (64, 96, 60, 60)):
    torch.manual_seed(0)
    u = CINCFlowUnit(C, C, 3).to(dev)
    with torch.no_grad():
        u.conv_tl.conv.weight.mul_(1 - 0.5 * torch.as_tensor(u.conv_tl.mask).to(dev))
    x = torch.randn(B, C, H, W, device=dev)
    with torch.no_grad():
        z, _ = u(x)
        tf = timeit(lambda: u(x))
        tr = timeit(lambda: u.reverse(z))
        err = float((u.reverse(z) - x).abs().max() / x.abs().max())
    xg = x.clone().requires_grad_(True)
    gz = torch.randn_like(z)
    def train():
        xg.grad = None
        for p_ in u.parameters(): p_.grad = None
        zz, _ = u(xg); zz.backward(gz)
    tt = timeit(train, 5)
    fl = 2.0 * B * C * C * 9 * H * W
    v = _lib.backward_variant(B, 1, C, H, W, 3, 3)
    print(f"CINC C{C} {H}x{W} B{B}: forward {tf:.3f} ms ({fl/tf/1e9:.0f} TF), reverse {tr:.3f} ms ({fl/tr/1e9:.0f} TF), training step {tt:.3f} ms "
          f"({3*fl/tt/1e9:.0f} TF)  [{v}]  round trip {err:.1e}", flush=True)
