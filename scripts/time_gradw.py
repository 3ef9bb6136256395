"""GPU helper: the grad-weight launch (kernel + reduce) alone, HIP events, over a few shapes.  FINCFLOW_LIB selects a variant
library (scripts/build_gradw_variant.sh); FINC_GRADW_NO_WINO=1 the staged kernel."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from fincflow_amd import _lib, ops
dev = torch.device("cuda:0")
shapes = ((256, 96, 64, 64, 3), (64, 96, 64, 64, 3), (256, 64, 64, 64, 3), (128, 128, 64, 64, 3), (256, 80, 32, 32, 3))
if len(sys.argv) > 1 and sys.argv[1] == "c3":
    shapes = shapes[:1]
elif len(sys.argv) > 1 and sys.argv[1] == "small":      # c2's bank
    shapes = ((64, 48, 32, 32, 3), (256, 48, 32, 32, 3), (64, 48, 64, 64, 3), (256, 48, 64, 64, 3))
elif len(sys.argv) > 1 and sys.argv[1] == "mid":        # the pair kernel's banks
    shapes = ((256, 64, 64, 64, 3), (256, 80, 64, 64, 3), (256, 96, 64, 64, 3), (128, 112, 64, 64, 3), (128, 128, 64, 64, 3))
elif len(sys.argv) > 1 and sys.argv[1] == "big":        # the banks on one tile pair per wave
    shapes = ((64, 192, 128, 128, 5), (128, 64, 64, 64, 5), (64, 128, 64, 64, 5), (64, 192, 64, 64, 3), (64, 256, 64, 64, 3), (32, 384, 64, 64, 3))
for (B, C, H, W, K) in shapes:
    torch.manual_seed(0)
    Cq = C // 4
    wc = torch.randn(4 * Cq, Cq, K, K, device=dev)
    x = torch.randn(B, C, H, W, device=dev)
    gz = torch.randn(B, C, H, W, device=dev)
    fn = lambda: ops.finc_backward(gz, x, wc, 4, 0xE4, need_gx=False, need_gw=True)
    for _ in range(10): fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(40): fn()
    b.record(); torch.cuda.synchronize()
    us = a.elapsed_time(b) / 40 * 1e3
    v = _lib.backward_variant(B, 4, Cq, H, W, K, K)["gradw"]
    fl = 2.0 * B * H * W * C * Cq * K * K
    print(f"C{C} {H}x{W} k{K} B={B:4d}: {us:7.1f} us  {v:9s} {fl / us * 1e-6:6.1f} direct-equivalent TFLOP/s", flush=True)
