"""GPU helper: random-shape parity sweep of the streaming-bank kernels (finc_stream.hip) against the oracle: inverse (auto against the
fp64 path, strict bit-exact with the fp32 order), forward, and -- every case -- both with a folded per-channel affine map (a non-zero
accumulator start: the one thing that makes a wrong-zero operand near a row's first columns visible, profiles/r05/stream/ablations.txt
item 6).  Shapes are drawn until the library says form 7 (a bank outside every table).
Usage: python scripts/fuzz_stream.py [n_cases] [seed].  Exits non-zero on the first mismatch."""
import os, sys
import numpy as np
import torch
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
from fincflow_amd import _lib, ops
from oracle import oracle

n = int(sys.argv[1]) if len(sys.argv) > 1 else 100
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 0)
dev = torch.device("cuda:0")
worst, done, forms = 0.0, 0, {}
while done < n:
    KH, KW = int(rng.integers(1, 8)), int(rng.integers(1, 8))
    Cq = int(rng.choice([rng.integers(1, 49), rng.integers(49, 130), rng.integers(130, 257)]))
    G = int(rng.choice([1, 1, 2, 3, 4, 4, 5, 8]))
    B = int(rng.integers(1, 4))
    H, W = int(rng.integers(1, 41)), int(rng.integers(1, 41))
    if rng.random() < 0.4:
        W = max(4, W // 4 * 4)                     # the 16-byte paths
    if B * G * Cq * H * W > 6_000_000:
        continue
    v = _lib.inverse_variant(B, G, Cq, H, W, KH, KW)
    if v is None or v["sec"] != 7 or _lib.lib().finc_forward_algo_for(Cq, H, W, KH, KW) != _lib.ALGO["mfma"]:
        continue
    orient = int(rng.integers(0, 1 << (2 * G)))
    std = (0.05 if max(KH, KW) < 5 else 0.02) * min(1.0, (24.0 / Cq) ** 0.5)
    ws = oracle.make_stored_weights(G, Cq, KH, KW, orient=orient, seed=done, std=std)
    wco = oracle.canonicalize(ws, G, orient)
    x = rng.standard_normal((B, G * Cq, H, W)).astype(np.float32)
    nthr = min(oracle.max_threads(), 16)
    z = oracle.forward_f32(x, wco, G, orient, nthreads=nthr)
    ref = oracle.inverse_via_f64(z, wco, G, orient, nthreads=nthr)
    ref32 = oracle.inverse_f32(z, wco, G, orient, nthreads=nthr)
    wc = ops.canonicalize(torch.from_numpy(ws).to(dev), G, orient)
    zt, xt = torch.from_numpy(z).to(dev), torch.from_numpy(x).to(dev)
    auto = ops.finc_inverse(zt, wc, G, orient, algo="auto").cpu().numpy()
    fwd = ops.finc_forward(xt, wc, G, orient).cpu().numpy()
    strict = ops.finc_inverse(zt, wc, G, orient, algo="strict").cpu().numpy()
    scale = max(np.abs(ref).max(), 1e-30)
    e_inv = np.abs(auto - ref).max() / scale
    e_fwd = np.abs(fwd - z).max() / max(np.abs(z).max(), 1e-30)
    exact = np.array_equal(strict, ref32)
    tol = max(1e-5, 2.0 * np.abs(ref32 - ref).max() / scale)
    # the folds (layers/actnorm.py:39-52): inverse(exp(s) y + t) and (forward(y) - t) exp(-s), each one launch, against the two-step form
    per = ws.shape[0] // G
    wts = [torch.from_numpy(ws[i * per:(i + 1) * per]).to(dev) for i in range(G)]
    ls = torch.from_numpy((0.1 * rng.standard_normal((1, G * Cq, 1, 1))).astype(np.float32)).to(dev)
    tr = torch.from_numpy((0.3 * rng.standard_normal((1, G * Cq, 1, 1))).astype(np.float32)).to(dev)
    cache = ops.PackedWeights()
    ia = cache.inverse_affine(zt, wts, G, orient, ls, tr)
    fa = cache.forward_affine(xt, wts, G, orient, ls, tr)
    e_ia = e_fa = 0.0
    if ia is not None:           # (None: activations not 16-byte aligned -- never here -- or no fused form)
        want = cache.inverse(torch.exp(ls) * zt + tr, wts, G, orient)
        e_ia = float((ia - want).abs().max() / want.abs().max().clamp_min(1e-30))
    if fa is not None:
        want = (cache.forward(xt, wts, G, orient) - tr) * torch.exp(-ls)
        e_fa = float((fa - want).abs().max() / want.abs().max().clamp_min(1e-30))
    folds_ok = ia is not None and fa is not None and e_ia <= tol and e_fa <= 1e-5
    worst = max(worst, e_inv, e_fwd, e_ia, e_fa)
    key = (v["nw"], v["cqp"] // (16 * v["nw"]))
    forms[key] = forms.get(key, 0) + 1
    tag = "ok" if (e_inv <= tol and e_fwd <= 1e-5 and exact and folds_ok) else "MISMATCH"
    print(f"{done:3d} B{B} G{G} Cq{Cq} {H}x{W} k{KH}x{KW} orient {orient:#x} waves {v['nw']} tiles {key[1]}: inv {e_inv:.1e} fwd {e_fwd:.1e} "
          f"folded inv {e_ia:.1e} fwd {e_fa:.1e} strict-exact {exact} {tag}", flush=True)
    if tag != "ok":
        sys.exit(1)
    done += 1
print("all ok, worst rel err %.2e; (waves, tiles per wave) -> cases: %s; faults pending %d, timeouts %d"
      % (worst, sorted(forms.items()), _lib.fault_pending(), _lib.hlp_timeouts()))
