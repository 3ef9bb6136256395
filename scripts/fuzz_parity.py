"""GPU helper: random-shape parity sweep of the product path against the oracle (inverse auto/strict, forward, grad-input
through autograd).  Usage: python scripts/fuzz_parity.py [n_cases] [seed].  Exits non-zero on the first mismatch."""
import os, sys
import numpy as np
import torch
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
sys.path.insert(0, os.path.join(REPO, "tests"))
from helpers import fuzz_case          # the generator tests/test_gpu_variants.py::test_bounded_fuzz_sweep uses
from fincflow_amd import ops
from oracle import oracle

n = int(sys.argv[1]) if len(sys.argv) > 1 else 80
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 0)
dev = torch.device("cuda:0")
ORI = 0xE4
worst = 0.0
for case in range(n):
    c = fuzz_case(rng, case)
    B, G, Cq, H, W, K, orient, std = c["B"], c["G"], c["Cq"], c["H"], c["W"], c["K"], c["orient"], c["std"]
    ws = oracle.make_stored_weights(G, Cq, K, K, orient=orient, seed=case, std=std)
    wco = oracle.canonicalize(ws, G, orient)
    x = rng.standard_normal((B, G * Cq, H, W)).astype(np.float32)
    z = oracle.forward_f32(x, wco, G, orient)
    ref = oracle.inverse_via_f64(z, wco, G, orient)
    wc = ops.canonicalize(torch.from_numpy(ws).to(dev), G, orient)
    zt, xt = torch.from_numpy(z).to(dev), torch.from_numpy(x).to(dev)
    auto = ops.finc_inverse(zt, wc, G, orient, algo="auto").cpu().numpy()
    fwd = ops.finc_forward(xt, wc, G, orient).cpu().numpy()
    scale = max(np.abs(ref).max(), 1e-30)
    e_inv = np.abs(auto - ref).max() / scale
    e_fwd = np.abs(fwd - z).max() / max(np.abs(z).max(), 1e-30)
    strict = ops.finc_inverse(zt, wc, G, orient, algo="strict").cpu().numpy()
    ref32 = oracle.inverse_f32(z, wco, G, orient)
    exact = np.array_equal(strict, ref32)
    # 1e-5 where the problem is well conditioned; a random bank at Cq >= 32 on a large map is not (the reference's own
    # fp32 order then differs from its fp64 path by 1e-4 and more), so the yardstick there is that fp32-order error
    tol = max(1e-5, 2.0 * np.abs(ref32 - ref).max() / scale)
    worst = max(worst, e_inv, e_fwd)
    tag = "ok" if (e_inv <= tol and e_fwd <= 1e-5 and exact) else "MISMATCH"
    print(f"{case:3d} B{B} G{G} Cq{Cq} {H}x{W} k{K} orient {orient:#x}: inv {e_inv:.1e} fwd {e_fwd:.1e} strict-exact {exact} |x|max {np.abs(ref).max():.1e} {tag}", flush=True)
    if tag != "ok":
        sys.exit(1)
print("all ok, worst rel err %.2e" % worst)
