"""GPU helper: forward time of one FastFlowUnit shape with the library selected by FINCFLOW_LIB (no checks): B C H W K"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from fincflow_amd import FastFlowUnit
dev = torch.device("cuda:0")
B, C, H, W, K = (int(a) for a in sys.argv[1:6])
unit = FastFlowUnit(C, C, K).to(dev)
x = torch.randn(B, C, H, W, device=dev)
with torch.no_grad():
    for _ in range(30): unit(x)
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(100): unit(x)
    b.record(); torch.cuda.synchronize()
print(os.environ.get("FINCFLOW_LIB", "default").split("/")[-1], f"B{B} C{C} {H}x{W}: forward {a.elapsed_time(b) / 100 * 1e3:.1f} us", flush=True)
