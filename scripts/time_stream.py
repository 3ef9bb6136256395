"""GPU helper: the streaming-bank kernels (finc_stream.hip) beside the reference-order kernels they replace.
time_stream.py [quick]  ->  one line per shape: inverse / forward on form 7, strict inverse / forward, TFLOP/s of the executed MFMAs."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from fincflow_amd import _lib, ops
from oracle import oracle

dev = torch.device("cuda:0")
quick = len(sys.argv) > 1 and sys.argv[1] == "quick"
# (B, G, Cq, H, W, KH, KW, what)
SHAPES = [
    (256, 1, 192, 64, 64, 3, 3, "CINCFlowUnit C=192 (cinc_flow.py:9-30), 64x64, B=256"),
    (256, 4, 128, 32, 32, 3, 3, "FastFlowUnit C=512, 32x32, B=256"),
    (256, 4, 100, 32, 32, 3, 3, "FastFlowUnit C=400, 32x32, B=256"),
    (64, 4, 64, 32, 32, 5, 5, "FastFlowUnit 5x5 C=256, 32x32, B=64"),
    (64, 4, 12, 32, 32, 4, 4, "FastFlowUnit 4x4 C=48, 32x32, B=64"),
    (64, 4, 12, 32, 32, 7, 7, "FastFlowUnit 7x7 C=48, 32x32, B=64"),
    (64, 4, 40, 32, 32, 2, 2, "FastFlowUnit 2x2 C=160, 32x32, B=64"),
    (64, 4, 32, 32, 32, 3, 5, "FastFlowUnit 3x5 C=128, 32x32, B=64"),
]
if quick:
    SHAPES = SHAPES[:2]


def timed(fn, reps, warm):
    for _ in range(warm): fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(reps): fn()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / reps * 1e3


for (B, G, Cq, H, W, KH, KW, what) in SHAPES:
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
        v = _lib.inverse_variant(B, G, Cq, H, W, KH, KW)
        t_inv = timed(lambda: cache.inverse(z, weights, G, orient, out=o), 5, 2)
        err = float((o - x).abs().max() / x.abs().max())
        t_fwd = timed(lambda: cache.forward(x, weights, G, orient, out=o), 5, 2)
        gzr = torch.randn_like(x)
        t_bwd = timed(lambda: ops.finc_backward(gzr, x, cache.w_canon, G, orient), 3, 1)
        bv = _lib.backward_variant(B, G, Cq, H, W, KH, KW)
        nb = max(1, B // 16)                       # the reference-order kernels on a sixteenth of the batch (they scale with it)
        wc = cache.w_canon
        zs, xs = z[:nb].contiguous(), x[:nb].contiguous()
        t_sinv = timed(lambda: ops.finc_inverse(zs, wc, G, orient, algo="strict"), 1, 1) * (B / nb)
        t_sfwd = timed(lambda: ops.finc_forward(xs, wc, G, orient, algo="strict"), 1, 1) * (B / nb)
    cqp = v["cqp"] if v else 0
    mf = 2.0 * KH * KW * cqp * cqp * H * W * B * G           # executed multiply-adds x 2 (padded bank, image pixels only)
    af = 2.0 * KH * KW * Cq * Cq * H * W * B * G
    print(f"{what}: form {v['sec'] if v else None} inverse {t_inv:9.1f} us ({mf / t_inv * 1e-6:5.1f} TF executed, {af / t_inv * 1e-6:5.1f} algorithmic) "
          f"forward {t_fwd:9.1f} us ({mf / t_fwd * 1e-6:5.1f} TF) | strict inverse {t_sinv:11.1f} us (x{t_sinv / t_inv:6.1f}) strict forward {t_sfwd:10.1f} us "
          f"(x{t_sfwd / t_fwd:5.1f}) | backward (grad-input {bv['conv_form']} + grad-weight {bv['gradw']}) {t_bwd:9.1f} us | round trip err {err:.1e}", flush=True)
