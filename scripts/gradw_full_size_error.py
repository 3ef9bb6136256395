"""GPU helper: grad_w at a bench-size reduction (c3's bank, 64x64, B images) against fp64 CPU autograd: the Winograd kernel and
(FINC_GRADW_NO_WINO=1) the direct one.  usage: gradw_full_size_error.py [B]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, torch.nn.functional as F
from fincflow_amd import FastFlowUnit, _lib
B = int(sys.argv[1]) if len(sys.argv) > 1 else 64
dev = torch.device("cuda:0")
torch.manual_seed(0)
unit = FastFlowUnit(96, 96, 3).to(dev)
x = torch.randn(B, 96, 64, 64, device=dev, requires_grad=True)
z, _ = unit(x)
gz = torch.randn_like(z)
z.backward(gz)
worst = 0.0
for m, xc, gc in zip((unit.conv_tl, unit.conv_tr, unit.conv_bl, unit.conv_br), torch.chunk(x.detach().cpu().double(), 4, 1), torch.chunk(gz.cpu().double(), 4, 1)):
    w = m.conv.weight.detach().cpu().double().requires_grad_(True)
    F.conv2d(F.pad(xc, m.pad), w).backward(gc)
    expect = w.grad * m.mask.double()
    got = m.conv.weight.grad.cpu().double()
    worst = max(worst, float((got - expect).abs().max() / expect.abs().max()))
print(f"B={B} ({B * 4096} terms per entry): grad_w kernel {_lib.backward_variant(B, 4, 24, 64, 64, 3, 3)['gradw']}, worst max-normalised error {worst:.2e}")
