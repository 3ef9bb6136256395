"""GPU helper: the fp64 inverse and forward -- matrix-core form (finc_f64.hip) against the reference-order kernels -- at the c2 and
c3 shapes; times by HIP events, fraction of the fp64 MFMA peak (78.6 TFLOP/s dense, MI355X_MICROARCH.md).   python scripts/time_f64.py"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from fincflow_amd import ops
from oracle import oracle
dev = torch.device("cuda:0")
FP64_PEAK = 78.6e12
for (B, C, H, W, K, strict_too) in ((64, 48, 32, 32, 3, True), (256, 96, 64, 64, 3, False), (32, 96, 64, 64, 3, True)):
    G, Cq = 4, C // 4
    ws = oracle.make_stored_weights(G, Cq, K, K, seed=1).astype(np.float64)
    wc = ops.canonicalize(torch.from_numpy(ws).to(dev), G, 0xE4)
    x = torch.randn(B, C, H, W, device=dev, dtype=torch.float64)
    z = ops.finc_forward(x, wc, G, 0xE4, algo="mfma")
    flops = 2.0 * B * C * H * W * K * K * Cq
    for name, fn in (("inverse", lambda a: ops.finc_inverse(z, wc, G, 0xE4, algo=a)), ("forward", lambda a: ops.finc_forward(x, wc, G, 0xE4, algo=a))):
        for algo in (("mfma", "strict") if strict_too else ("mfma",)):
            n = 20 if algo == "mfma" else 2
            for _ in range(2): out = fn(algo)
            torch.cuda.synchronize()
            a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            a.record()
            for _ in range(n): out = fn(algo)
            b.record(); torch.cuda.synchronize()
            ms = a.elapsed_time(b) / n
            err = float((out - x).abs().max() / x.abs().max()) if name == "inverse" else float((out - z).abs().max() / z.abs().max())
            print(f"C{C} {H}x{W} B={B} fp64 {name:8s} {algo:6s}: {ms:9.3f} ms  {flops / ms / 1e9:8.2f} TFLOP/s = {flops / ms / 1e-3 / FP64_PEAK:.3f} of the fp64 MFMA peak  "
                  f"{'round trip' if name == 'inverse' else 'vs mfma'} err {err:.1e}", flush=True)
