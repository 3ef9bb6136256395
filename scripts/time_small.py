"""GPU helper: the inverse on the under-filled chip -- c3 at B = 4..128 and c2 at B = 16..64, with and without the band
pipeline (FINC_NO_BND is read once per process: two child processes)."""
import os, subprocess, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
if len(sys.argv) > 1 and sys.argv[1] == "child":
    import torch
    from fincflow_amd import FastFlowUnit, _lib
    dev = torch.device("cuda:0")
    for (C, H, W, K, Bs) in ((96, 64, 64, 3, (4, 8, 16, 32, 64, 128)), (48, 32, 32, 3, (16, 32, 64)), (96, 32, 32, 3, (64,)), (96, 48, 64, 3, (32,))):
        torch.manual_seed(0)
        unit = FastFlowUnit(C, C, K).to(dev)
        for B in Bs:
            x = torch.randn(B, C, H, W, device=dev)
            with torch.no_grad():
                z, _ = unit(x)
                o = torch.empty_like(z)
                fn = lambda: unit._cache.inverse(z, unit._weights(), 4, 0xE4, out=o)
                for _ in range(20): fn()
                torch.cuda.synchronize()
                a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                a.record()
                for _ in range(50): fn()
                b.record(); torch.cuda.synchronize()
                us = a.elapsed_time(b) / 50 * 1e3
                err = float((o - x).abs().max() / x.abs().max())
            v = _lib.inverse_variant(B, 4, C // 4, H, W, K, K)
            print(f"C{C} {H}x{W} B={B:4d}: {us:7.1f} us  bands {v.get('bands', 0)} nw {v['nw']} npw {v['npw']} form {v['sec']} chain {v.get('chain', 0)} "
                  f"-> {us / max(v.get("chain", 0), 1):.3f} us/step  err {err:.1e}", flush=True)
else:
    for env in ({"FINC_NO_BND": "1"}, {}):
        print("==", env or "default", flush=True)
        subprocess.run([sys.executable, os.path.abspath(__file__), "child"], env=dict(os.environ, **env), check=True)
