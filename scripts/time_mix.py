"""GPU helper: the mixing kernel (ops.finc_mix) against F.conv2d / torch.matmul at the bench shapes."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ.setdefault("MIOPEN_FIND_MODE", "2")
import torch
from fincflow_amd import ops
dev = torch.device("cuda:0")
def timeit(fn, n=50):
    t_end = time.perf_counter() + 0.3        # clocks ramp up over the first tenths of a second of load
    while time.perf_counter() < t_end:
        for _ in range(5): fn()
        torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(n): fn()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / n * 1e3
SHAPES = ((256, 96, 64, 64), (64, 48, 32, 32), (64, 192, 128, 128), (128, 12, 16, 16), (256, 96, 63, 63))
if len(sys.argv) > 1 and sys.argv[1] == "sweep":   # every channel count of the kernel's table at a chip-filling size
    SHAPES = tuple((max(8, 4096 // C) * 8, C, 64, 64) for C in (4, 8, 12, 16, 24, 32, 48, 64, 96, 128, 192))
for (B, C, H, W) in SHAPES:
    x = torch.randn(B, C, H, W, device=dev); M = torch.randn(C, C, device=dev) / C ** 0.5; b = torch.randn(C, device=dev)
    o = torch.empty_like(x)
    t = timeit(lambda: ops.finc_mix(x, M, b, out=o))
    ref = torch.nn.functional.conv2d(x, M.view(C, C, 1, 1), b)
    err = float((o - ref).abs().max() / ref.abs().max())
    tm = timeit(lambda: torch.matmul(M, x.view(B, C, H * W)), 20)
    tc = timeit(lambda: torch.nn.functional.conv2d(x, M.view(C, C, 1, 1), b), 20)
    gb = 8 * x.numel() / t / 1e3
    print(f"B{B} C{C} {H}x{W}: mix {t:.1f} us ({gb:.0f} GB/s, {2*x.numel()*C/t/1e6:.1f} TF) | torch.matmul {tm:.1f} | F.conv2d {tc:.1f} | rel err vs conv2d {err:.1e}", flush=True)
