"""GPU helper: the forward launch of a FastFlowUnit (packed bank cached) over a few shapes, HIP events; `set_forward_form(1)`
beside the library's choice.  usage: time_forward.py [wide]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from fincflow_amd import FastFlowUnit, _lib
dev = torch.device("cuda:0")
shapes = ((64, 128, 64, 64, 3), (64, 192, 64, 64, 3), (64, 256, 64, 64, 3), (32, 192, 128, 128, 3), (256, 112, 64, 64, 3), (16, 192, 64, 64, 3))
for (B, C, H, W, K) in shapes:
    torch.manual_seed(0)
    unit = FastFlowUnit(C, C, K).to(dev)
    x = torch.randn(B, C, H, W, device=dev)
    line = f"C{C} {H}x{W} k{K} B={B:4d}:"
    for form in (0, 1):
        _lib.set_forward_form(form)
        with torch.no_grad():
            for _ in range(5): unit(x)
            torch.cuda.synchronize()
            a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            a.record()
            for _ in range(30): unit(x)
            b.record(); torch.cuda.synchronize()
        us = a.elapsed_time(b) / 30 * 1e3
        v = _lib.backward_variant(B, 4, C // 4, H, W, K, K)["conv_form"]
        fl = 2.0 * B * H * W * C * (C // 4) * K * K
        line += f"  {v:10s} {us:8.1f} us ({fl / us * 1e-6:6.1f} direct-equivalent TFLOP/s)"
    _lib.set_forward_form(0)
    print(line, flush=True)
