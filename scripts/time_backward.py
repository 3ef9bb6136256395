"""GPU helper: forward+backward timing of one FastFlowUnit (training step of the layer)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from fincflow_amd import FastFlowUnit
dev = torch.device("cuda:0")
B, C, H, W, K = (int(a) for a in sys.argv[1:6]) if len(sys.argv) > 5 else (256, 96, 64, 64, 3)
unit = FastFlowUnit(C, C, K).to(dev)
x = torch.randn(B, C, H, W, device=dev, requires_grad=True)
gz = torch.randn(B, C, H, W, device=dev)
def step():
    z, _ = unit(x)
    z.backward(gz)
for _ in range(2): step()
torch.cuda.synchronize()
a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
a.record()
for _ in range(5): step()
b.record(); torch.cuda.synchronize()
print("fwd+bwd %.2f ms per step (B=%d C=%d %dx%d k%d)" % (a.elapsed_time(b) / 5, B, C, H, W, K))
