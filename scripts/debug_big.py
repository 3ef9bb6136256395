"""GPU helper: the big-bank inverse (finc_big.hip, 64 < Cq <= 96 at 3x3) against the oracle on small problems, and its time at
the CINC C = 96 shapes.  Usage: python scripts/debug_big.py [time]"""
import os, sys, time
import numpy as np
import torch
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
sys.path.insert(0, os.path.join(REPO, "tests"))
from fincflow_amd import ops, _lib
from oracle import oracle

dev = torch.device("cuda:0")
rng = np.random.default_rng(0)
# (B, G, Cq, H, W, orient)
CASES = [(2, 1, 96, 16, 16, 0), (1, 1, 96, 40, 32, 3), (2, 4, 72, 20, 16, None), (1, 1, 65, 33, 64, 1), (3, 1, 96, 7, 20, 2),
         (1, 1, 96, 40, 48, 0), (1, 1, 96, 40, 80, 3), (1, 4, 72, 20, 128, None), (1, 1, 65, 33, 96, 1), (2, 1, 96, 50, 68, 2),
         (1, 1, 50, 40, 256, 0), (1, 1, 64, 20, 160, 3), (1, 4, 48, 18, 256, None), (1, 1, 36, 35, 512, 1), (1, 1, 50, 256, 256, 2)]
for B, G, Cq, H, W, orient in CASES:
    ori = 0xE4 if orient is None else orient
    ws = oracle.make_stored_weights(G, Cq, 3, 3, orient=ori, seed=Cq + H, std=0.05 * (24.0 / Cq) ** 0.5)
    wco = oracle.canonicalize(ws, G, ori)
    x = rng.standard_normal((B, G * Cq, H, W)).astype(np.float32)
    z = oracle.forward_f32(x, wco, G, ori)
    ref = oracle.inverse_via_f64(z, wco, G, ori)
    wc = ops.canonicalize(torch.from_numpy(ws).to(dev), G, ori)
    v = _lib.inverse_variant(B, G, Cq, H, W, 3, 3)
    out = ops.finc_inverse(torch.from_numpy(z).to(dev), wc, G, ori, algo="auto").cpu().numpy()
    err = np.abs(out - ref).max() / np.abs(ref).max()
    bad = np.argwhere(np.abs(out - ref) > 1e-4 * np.abs(ref).max())
    print(f"B{B} G{G} Cq{Cq} {H}x{W} orient {orient}: form {v['sec'] if v else None} cqp {v['cqp'] if v else None}, rel err {err:.2e}, bad entries {len(bad)}"
          + (f", first {bad[0].tolist()} last {bad[-1].tolist()}" if len(bad) else ""), flush=True)

if len(sys.argv) > 1 and sys.argv[1] == "time":
    for B, G, Cq, H, W in ((256, 1, 96, 64, 64), (256, 1, 96, 32, 32), (64, 4, 96, 32, 32), (256, 1, 80, 64, 64), (256, 1, 96, 64, 128),
                           (100, 1, 50, 256, 256), (64, 4, 64, 64, 256)):
        ws = oracle.make_stored_weights(G, Cq, 3, 3, orient=0xE4 if G == 4 else 0, seed=1, std=0.025)
        wc = ops.canonicalize(torch.from_numpy(ws).to(dev), G, 0xE4 if G == 4 else 0)
        z = torch.randn(B, G * Cq, H, W, device=dev)
        cache = ops.PackedWeights()
        # time through the packed path (what a layer does): pack once, launch per step
        L = _lib.lib()
        packed = torch.empty(L.finc_workspace_bytes(G, Cq, 3, 3), dtype=torch.uint8, device=dev)
        _lib.check(L.finc_pack_inverse_weights_f32(wc.data_ptr(), packed.data_ptr(), G, Cq, 3, 3, None), "pack")
        out = torch.empty_like(z)
        ori = 0xE4 if G == 4 else 0
        run = lambda: _lib.check(L.finc_inverse_packed_f32(z.data_ptr(), packed.data_ptr(), out.data_ptr(), B, G, Cq, H, W, 3, 3, ori, None), "inv")
        t_end = time.perf_counter() + 0.3
        while time.perf_counter() < t_end:
            run(); torch.cuda.synchronize()
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        for _ in range(10): run()
        b.record(); torch.cuda.synchronize()
        ms = a.elapsed_time(b) / 10
        fl = 2.0 * B * G * Cq * Cq * 9 * H * W
        pf = torch.empty(L.finc_workspace_bytes(G, Cq, 3, 3), dtype=torch.uint8, device=dev)
        _lib.check(L.finc_pack_forward_weights_f32(wc.data_ptr(), pf.data_ptr(), G, Cq, 3, 3, None), "packf")
        runf = lambda: _lib.check(L.finc_forward_packed_f32(z.data_ptr(), pf.data_ptr(), out.data_ptr(), B, G, Cq, H, W, 3, 3, ori, None), "fwd")
        for _ in range(3): runf()
        torch.cuda.synchronize()
        a.record()
        for _ in range(10): runf()
        b.record(); torch.cuda.synchronize()
        msf = a.elapsed_time(b) / 10
        print(f"B{B} G{G} Cq{Cq} {H}x{W}: inverse {ms:.3f} ms = {fl / ms / 1e9:.1f} TFLOP/s | forward {msf:.3f} ms = {fl / msf / 1e9:.1f} TFLOP/s", flush=True)
