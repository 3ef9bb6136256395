"""GPU helper: forward (and training step) of one FastFlowUnit with and without the Winograd kernel (FINC_NO_WINO=1), child processes."""
import os, subprocess, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
if len(sys.argv) > 1 and sys.argv[1] == "child":
    import torch
    from fincflow_amd import FastFlowUnit, _lib
    dev = torch.device("cuda:0")
    for (B, C, H, W, K) in ((256, 96, 64, 64, 3), (64, 48, 32, 32, 3), (32, 96, 64, 64, 3), (128, 64, 32, 32, 3)):
        torch.manual_seed(0)
        unit = FastFlowUnit(C, C, K).to(dev)
        x = torch.randn(B, C, H, W, device=dev)
        with torch.no_grad():
            z, _ = unit(x)
            import torch.nn.functional as F
            ref = torch.cat([F.conv2d(F.pad(c.double(), m.pad), m.conv.weight.detach().double()) for m, c in
                             zip((unit.conv_tl, unit.conv_tr, unit.conv_bl, unit.conv_br), torch.chunk(x[:4], 4, 1))], 1)
            err = float((z[:4].double() - ref).abs().max() / ref.abs().max())
            fn = lambda: unit(x)
            for _ in range(20): fn()
            torch.cuda.synchronize()
            a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            a.record()
            for _ in range(50): fn()
            b.record(); torch.cuda.synchronize()
            us = a.elapsed_time(b) / 50 * 1e3
        xg = x.clone().requires_grad_(True)
        gz = torch.randn_like(z)
        def train():
            xg.grad = None
            for p_ in unit.parameters(): p_.grad = None
            zz, _ = unit(xg); zz.backward(gz)
        for _ in range(5): train()
        torch.cuda.synchronize()
        a.record()
        for _ in range(20): train()
        b.record(); torch.cuda.synchronize()
        tr = a.elapsed_time(b) / 20 * 1e3
        flops = 2 * B * C * H * W * K * K * (C // 4)
        print(f"B{B} C{C} {H}x{W}: forward {us:7.1f} us = {flops / us / 1e6 / 157.3:.3f} of fp32 peak (direct-equivalent flops), "
              f"{8 * B * C * H * W / us / 1e3 / 8000:.3f} of HBM peak, err vs fp64 conv2d {err:.1e}; training step {tr:7.1f} us", flush=True)
else:
    for env in ({"FINC_NO_WINO": "1"}, {}):
        print("==", env or "default (Winograd F(2,3) along W)", flush=True)
        subprocess.run([sys.executable, os.path.abspath(__file__), "child"], env=dict(os.environ, **env), check=True)
