"""GPU helper for scripts/prof_stream.sh: a few launches of the streaming-bank inverse and forward at ONE shape (no reference-order
kernels beside them, so a profile holds these kernels only).  prof_stream.py B G Cq H W KH KW [reps]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from fincflow_amd import _lib, ops
from oracle import oracle

dev = torch.device("cuda:0")
B, G, Cq, H, W, KH, KW = (int(a) for a in sys.argv[1:8])
reps = int(sys.argv[8]) if len(sys.argv) > 8 else 5
orient = 0xE4 if G == 4 else 0
std = (0.05 if max(KH, KW) < 5 else 0.02) * min(1.0, (24.0 / Cq) ** 0.5)
ws = torch.from_numpy(oracle.make_stored_weights(G, Cq, KH, KW, orient=orient, seed=1, std=std)).to(dev)
per = ws.shape[0] // G
weights = [ws[i * per:(i + 1) * per].clone() for i in range(G)]
cache = ops.PackedWeights()
x = torch.randn(B, G * Cq, H, W, device=dev)
with torch.no_grad():
    z = cache.forward(x, weights, G, orient)
    o = torch.empty_like(z)
    for _ in range(reps):
        cache.inverse(z, weights, G, orient, out=o)
    torch.cuda.synchronize()
    err = float((o - x).abs().max() / x.abs().max())
    for _ in range(reps - 1):
        cache.forward(x, weights, G, orient, out=o)
    torch.cuda.synchronize()
print(f"stream B{B} G{G} Cq{Cq} {H}x{W} k{KH}x{KW}: form {_lib.inverse_variant(B, G, Cq, H, W, KH, KW)['sec']} round trip err {err:.1e}")
