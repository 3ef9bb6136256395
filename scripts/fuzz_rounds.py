"""GPU helper: random problem sets aimed at the launch rules of round 5's third session, against the oracle.
  inverse: problem counts on either side of the round boundaries (257 .. 2,300 problems on tiny maps) -- the remainder launch, the 28-channel
           bank's borrowed two-wave kernel, the small-batch variants; auto against the oracle's fp64 path, strict bit-exact;
  forward: strip counts that are no power of two on maps of 8 .. 70 rows -- the row-chunk rule of every forward family (F(4,3), its M-split,
           F(2,3), F(2,5), strip kernel); forward against the oracle, and the inverse of the result back to x.
Usage: python scripts/fuzz_rounds.py [n_inverse] [n_forward] [seed].  Exits non-zero on the first mismatch."""
import os, sys
import numpy as np
import torch
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
from fincflow_amd import ops, _lib
from oracle import oracle

n_inv = int(sys.argv[1]) if len(sys.argv) > 1 else 60
n_fwd = int(sys.argv[2]) if len(sys.argv) > 2 else 60
rng = np.random.default_rng(int(sys.argv[3]) if len(sys.argv) > 3 else 0)
dev = torch.device("cuda:0")
nthr = min(oracle.max_threads(), 16)
worst, bad = 0.0, 0


def orient_of(G):
    return 0xE4 if G == 4 else int(rng.integers(0, 4)) if G == 1 else (0x1B & ((1 << (2 * G)) - 1))


for case in range(n_inv):
    K = int(rng.choice([3, 3, 3, 2]))
    Cq = int(rng.choice([4, 8, 12, 16, 20, 24, 25, 27, 28, 30, 32] if K == 3 else [8, 12, 16, 24, 32]))
    G = int(rng.choice([1, 2, 4, 4, 4]))
    edge = int(rng.choice([256, 512, 768, 1024, 1024, 1536, 2048]))
    p = max(G, edge + int(rng.integers(-40, 300)))
    B = max(1, p // G)
    H = int(rng.integers(1, 7))
    W = int(rng.choice([4, 8, 12, 16, 16, 32, 64]))
    if W == 64: H = min(H, 3)
    orient = orient_of(G)
    std = 0.05 * min(1.0, (24.0 / Cq) ** 0.5)
    ws = oracle.make_stored_weights(G, Cq, K, K, orient=orient, seed=case, std=std)
    wco = oracle.canonicalize(ws, G, orient)
    z = rng.standard_normal((B, G * Cq, H, W)).astype(np.float32)
    ref = oracle.inverse_via_f64(z, wco, G, orient, nthreads=nthr)
    ref32 = oracle.inverse_f32(z, wco, G, orient, nthreads=nthr)
    wc = ops.canonicalize(torch.from_numpy(ws).to(dev), G, orient)
    zt = torch.from_numpy(z).to(dev)
    auto = ops.finc_inverse(zt, wc, G, orient, algo="auto").cpu().numpy()
    strict = ops.finc_inverse(zt, wc, G, orient, algo="strict").cpu().numpy()
    scale = max(np.abs(ref).max(), 1e-30)
    e = np.abs(auto - ref).max() / scale
    tol = max(1e-5, 2.0 * np.abs(ref32 - ref).max() / scale)
    ok = e <= tol and np.array_equal(strict, ref32)
    v = _lib.inverse_variant(B, G, Cq, H, W, K, K)
    worst = max(worst, e); bad += 0 if ok else 1
    print("inverse %3d  B=%4d G=%d Cq=%2d %dx%-2d k%d  problems=%4d  main kernel cqp=%d nw=%d npw=%d form=%d  err %.2e  %s"
          % (case, B, G, Cq, H, W, K, B * G, v["cqp"], v["nw"], v["npw"], v["sec"], e, "ok" if ok else "MISMATCH"), flush=True)
    if not ok: sys.exit(1)

for case in range(n_fwd):
    K = int(rng.choice([3, 3, 3, 2, 5]))
    Cq = int(rng.choice([4, 8, 12, 16, 24, 28, 40, 64] if K == 3 else [8, 16, 24] if K == 2 else [8, 16, 48]))
    G = int(rng.choice([1, 4, 4]))
    H = int(rng.integers(8, 71))
    W = int(rng.choice([16, 32, 48, 64, 20, 128]))
    cap = (1 << 22) // (G * Cq * H * W) * 6 + 1               # keep the oracle's work per case around a second
    B = int(rng.integers(1, max(2, min(cap, 120))))
    orient = orient_of(G)
    std = (0.05 if K < 5 else 0.02) * min(1.0, (24.0 / Cq) ** 0.5)
    ws = oracle.make_stored_weights(G, Cq, K, K, orient=orient, seed=1000 + case, std=std)
    wco = oracle.canonicalize(ws, G, orient)
    x = rng.standard_normal((B, G * Cq, H, W)).astype(np.float32)
    zr = oracle.forward_f32(x, wco, G, orient, nthreads=nthr)
    wc = ops.canonicalize(torch.from_numpy(ws).to(dev), G, orient)
    xt = torch.from_numpy(x).to(dev)
    zt = ops.finc_forward(xt, wc, G, orient)
    e = np.abs(zt.cpu().numpy() - zr).max() / max(np.abs(zr).max(), 1e-30)
    form = _lib.backward_variant(B, G, Cq, H, W, K, K)["conv_form"]
    ok = e <= 1e-5
    worst = max(worst, e); bad += 0 if ok else 1
    print("forward %3d  B=%3d G=%d Cq=%2d %2dx%-3d k%d  %-10s err %.2e  %s" % (case, B, G, Cq, H, W, K, form, e, "ok" if ok else "MISMATCH"), flush=True)
    if not ok: sys.exit(1)
print("fuzz_rounds: %d inverse + %d forward cases, worst %.2e, %d mismatches" % (n_inv, n_fwd, worst, bad))
