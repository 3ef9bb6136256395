import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__)))); sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import numpy as np, torch
from helpers import *
from test_glow_stack import build_ours
from oracle import oracle
from fincflow_amd import FastFlowUnit, ops
g = golden("stack_c4_small"); dev = torch.device("cuda:0")
layers = build_ours(); fill_stack_parameters(layers, ffu_weights=g); layers = [l.to(dev) for l in layers]
h = torch.from_numpy(g["x"]).to(dev)
with torch.no_grad():
    for idx, m in enumerate(layers):
        hin = h
        h, ld = m(h, None)
        if isinstance(m, FastFlowUnit):
            ws = torch.cat([w.detach() for w in m._weights()]).cpu().numpy()
            wc = oracle.canonicalize(ws, 4, 0xE4)
            ref = oracle.forward_f32(hin.cpu().numpy(), wc)
            e = rel_err(h.cpu().numpy(), ref)
            back = m.reverse(h)
            e2 = rel_err(back.cpu().numpy(), hin.cpu().numpy())
            a = ops.finc_forward(hin.contiguous(), torch.from_numpy(wc).to(dev), algo="strict")
            print(idx, "FFU", tuple(hin.shape), "contig", hin.is_contiguous(), "fwd err %.2e" % e, "rt err %.2e" % e2, "strict-vs-ref %.2e" % rel_err(a.cpu().numpy(), ref), "algo", ops._lib.lib().finc_forward_algo_for(hin.shape[1]//4, hin.shape[2], hin.shape[3], 3, 3))
        else:
            print(idx, type(m).__name__, tuple(h.shape), float(h.abs().max()))
print("fwd z err %.3e" % rel_err(h.cpu().numpy(), g["z"]))
with torch.no_grad():
    r = torch.from_numpy(g["z_in"]).to(dev)
    for idx, m in reversed(list(enumerate(layers))):
        rin = r
        r = m.reverse(r, None); r = r[0] if isinstance(r, tuple) else r
        print("rev", idx, type(m).__name__, "max %.3e" % float(r.abs().max()), "contig in", rin.is_contiguous(), rin.stride())
print("rev err %.3e" % rel_err(r.cpu().numpy(), g["x_rev"]))
