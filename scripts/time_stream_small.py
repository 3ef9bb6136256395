"""GPU helper: the one-wave streaming-bank problems (Cq <= 48) at two problem counts, for A/B runs of ablation builds (FINCFLOW_LIB)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from fincflow_amd import _lib, ops
from oracle import oracle
dev = torch.device("cuda:0")
for (B, G, Cq, H, W, KH, KW) in [(64, 4, 12, 32, 32, 4, 4), (256, 4, 12, 32, 32, 4, 4), (64, 4, 12, 32, 32, 7, 7), (256, 4, 12, 32, 32, 7, 7),
                                 (64, 4, 40, 32, 32, 2, 2), (256, 4, 40, 32, 32, 2, 2)]:
    std = (0.05 if max(KH, KW) < 5 else 0.02) * min(1.0, (24.0 / Cq) ** 0.5)
    ws = torch.from_numpy(oracle.make_stored_weights(G, Cq, KH, KW, orient=0xE4, seed=1, std=std)).to(dev)
    per = ws.shape[0] // G
    weights = [ws[i * per:(i + 1) * per].clone() for i in range(G)]
    cache = ops.PackedWeights()
    x = torch.randn(B, G * Cq, H, W, device=dev)
    with torch.no_grad():
        z = cache.forward(x, weights, G, 0xE4)
        o = torch.empty_like(z)
        for _ in range(5): cache.inverse(z, weights, G, 0xE4, out=o)
        torch.cuda.synchronize()
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        for _ in range(20): cache.inverse(z, weights, G, 0xE4, out=o)
        b.record(); torch.cuda.synchronize()
    v = _lib.inverse_variant(B, G, Cq, H, W, KH, KW)
    steps = ((H + 15) // 16 - 1) * max(W + KW - 1, 20) + W + 15
    us = a.elapsed_time(b) / 20 * 1e3
    print(f"B{B} G{G} Cq{Cq} {H}x{W} k{KH}x{KW} ({B * G} problems, {v['nw']} wave, {steps} steps): {us:8.1f} us = {us / steps:6.3f} us per step", flush=True)
