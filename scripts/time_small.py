"""GPU helper: the inverse on the under-filled chip -- c3 at B = 4..128, c2 at B = 16..64, the c4 unit shapes -- with the
role-split kernel (default) and without it (FINC_SPLIT_MAX=0: the wavefront kernel's table); two child processes, because the
switch is read once per process."""
import os, subprocess, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
if len(sys.argv) > 1 and sys.argv[1] == "child":
    import torch
    from fincflow_amd import FastFlowUnit, _lib
    dev = torch.device("cuda:0")
    for (C, H, W, K, Bs) in ((96, 64, 64, 3, (4, 8, 16, 32, 64, 128)), (48, 32, 32, 3, (16, 32, 64)), (96, 32, 32, 3, (64,)),
                             (96, 48, 64, 3, (32,)), (12, 16, 16, 3, (128,)), (24, 8, 8, 3, (128,)), (48, 4, 4, 3, (128,)),
                             (64, 32, 32, 2, (32,))):
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
            P = min(16, W)
            chain = ((H + P - 1) // P) * W + P - 1
            print(f"C{C} {H}x{W} k{K} B={B:4d}: {us:7.1f} us  waves {v['nw']} npw {v['npw']} form {v['sec']}  chain {chain} "
                  f"-> {us / chain:.3f} us/step  err {err:.1e}", flush=True)
else:
    for env in ({"FINC_SPLIT_MAX": "0"}, {}):
        print("==", env or "default", flush=True)
        subprocess.run([sys.executable, os.path.abspath(__file__), "child"], env=dict(os.environ, **env), check=True)
