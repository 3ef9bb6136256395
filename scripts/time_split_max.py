"""GPU helper: the role-split inverse with TWO problems per compute unit (FINC_SPLIT_MAX=512) against today's choice for
256 < B*G <= 512 (the packed K-split rows of the wavefront kernel); child processes (the switch is read once per process)."""
import os, subprocess, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
if len(sys.argv) > 1 and sys.argv[1] == "child":
    import torch
    from fincflow_amd import FastFlowUnit, _lib
    dev = torch.device("cuda:0")
    for (C, H, W, K, Bs) in ((96, 64, 64, 3, (64, 72, 96, 128, 160, 192)), (48, 32, 32, 3, (64, 96, 128, 192)), (96, 32, 32, 3, (64, 128)),
                             (64, 64, 64, 3, (128,)), (64, 32, 32, 2, (128,))):
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
            print(f"C{C} {H}x{W} k{K} B={B:4d}: {us:7.1f} us  waves {v['nw']} npw {v['npw']} form {v['sec']} workgroups {v['workgroups']}  err {err:.1e}", flush=True)
    print("timeouts", _lib.hlp_timeouts())
else:
    for env in ({}, {"FINC_SPLIT_MAX": "512"}, {"FINC_SPLIT_MAX": "768"}):
        print("==", env or "default (role-split up to 256 problems)", flush=True)
        subprocess.run([sys.executable, os.path.abspath(__file__), "child"], env=dict(os.environ, **env), check=True)
