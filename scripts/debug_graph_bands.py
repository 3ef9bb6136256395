import os, sys
sys.path.insert(0, os.getcwd())
import torch
from fincflow_amd import FastFlowUnit, _lib
dev = torch.device("cuda:0")
B, C, H, W = 8, 96, 64, 64
torch.manual_seed(5)
unit = FastFlowUnit(C, C, 3).to(dev)
y = torch.randn(B, C, H, W, device=dev)
with torch.no_grad():
    ref = unit.reverse(y)
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        out = unit.reverse(y)
    for r in range(4):
        out.zero_()
        g.replay()
        torch.cuda.synchronize()
        d = (out != ref)
        rows = d.any(dim=3).any(dim=1).any(dim=0).nonzero().flatten().tolist()
        print("replay", r, "equal", bool(torch.equal(out, ref)), "bad rows", rows[:10], len(rows), "zero rows", (out == 0).all(dim=3).all(dim=1).all(dim=0).nonzero().flatten().tolist()[:10])
print("timeouts", _lib.hlp_timeouts(), "fault", _lib.fault_pending())
