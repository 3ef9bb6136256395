"""GPU helper: random shapes through FastFlowUnit's backward, grad_w and grad_x entry by entry against CPU fp64 autograd through
F.pad + F.conv2d (the reference's forward, layers/conv.py:102-107) times the gradient mask -- aimed at the Winograd grad-weight
kernels (3x3: 13..96 channels per group, 5x5: 13..48), with the direct kernels' shapes mixed in.  usage: fuzz_gradw.py [cases] [seed]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch, torch.nn.functional as F
from fincflow_amd import FastFlowUnit, _lib
n_cases = int(sys.argv[1]) if len(sys.argv) > 1 else 100
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 5)
dev = torch.device("cuda:0")
worst, bad = {}, 0
for case in range(n_cases):
    K = int(rng.choice([3, 3, 3, 5]))
    Cq = int(rng.choice([13, 16, 20, 22, 24, 28, 32, 40, 48, 64, 96] if K == 3 else [13, 16, 20, 32, 48]))
    if rng.random() < 0.15: Cq = int(rng.integers(1, 13))
    W = int(rng.choice([4, 8, 12, 16, 20, 24, 28, 32, 36, 40, 48, 52, 64, 68, 96]))
    if rng.random() < 0.1: W = int(rng.integers(3, 40))                   # widths the staged / Winograd forms do not take
    H = int(rng.integers(1, 20))
    B = int(rng.integers(1, 4))
    C = 4 * Cq
    torch.manual_seed(case)
    unit = FastFlowUnit(C, C, K).to(dev)
    x = torch.randn(B, C, H, W, device=dev, requires_grad=True)
    z, _ = unit(x)
    gz = torch.randn_like(z)
    z.backward(gz)
    xc = x.detach().cpu().double().requires_grad_(True)
    outs, ws = [], []
    for m, chunk in zip((unit.conv_tl, unit.conv_tr, unit.conv_bl, unit.conv_br), torch.chunk(xc, 4, 1)):
        w = m.conv.weight.detach().cpu().double().requires_grad_(True)
        ws.append(w)
        outs.append(F.conv2d(F.pad(chunk, m.pad), w))
    torch.cat(outs, 1).backward(gz.cpu().double())
    ex = float((x.grad.cpu().double() - xc.grad).abs().max() / xc.grad.abs().max())
    ew = 0.0
    for m, w in zip((unit.conv_tl, unit.conv_tr, unit.conv_bl, unit.conv_br), ws):
        expect = w.grad * m.mask.double()
        got = m.conv.weight.grad.cpu().double()
        ew = max(ew, float((got - expect).abs().max() / expect.abs().max()))
        assert torch.all(got[m.mask == 0] == 0)
    v = _lib.backward_variant(B, 4, Cq, H, W, K, K)
    ok = ew <= 2e-5 and ex <= 1e-5
    bad += not ok
    worst[v["gradw"]] = max(worst.get(v["gradw"], 0.0), ew)
    print(f"{case} B{B} Cq{Cq} {H}x{W} k{K}: gradw {v['gradw']:14s} err {ew:.1e}  grad_x {ex:.1e} {'ok' if ok else 'BAD'}", flush=True)
print("worst grad_w error per kernel:", {k: f"{e:.2e}" for k, e in worst.items()}, " failures:", bad)
sys.exit(1 if bad else 0)
