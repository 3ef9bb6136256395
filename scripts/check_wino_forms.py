"""GPU helper: parity (vs fp64 conv2d, forward and grad-input) and time of the 3x3 forward on each form -- direct strip kernel
(FINC_NO_WINO=1), Winograd F(2,3) (FINC_WINO_FORM=2), F(4,3) (FINC_WINO_FORM=4), and the library's own choice -- child processes."""
import os, subprocess, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
SHAPES = ((256, 96, 64, 64), (64, 48, 32, 32), (32, 96, 64, 64), (128, 96, 64, 64), (16, 96, 128, 128), (8, 92, 20, 60), (3, 40, 7, 4),
          (2, 24, 9, 132), (5, 8, 16, 68), (64, 64, 32, 32), (256, 96, 32, 64), (256, 48, 64, 64))
if os.environ.get("WINO_SHAPES"):
    SHAPES = tuple(tuple(int(v) for v in t.split(",")) for t in os.environ["WINO_SHAPES"].split(";"))
if len(sys.argv) > 1 and sys.argv[1] == "child":
    import torch
    import torch.nn.functional as F
    from fincflow_amd import FastFlowUnit, _lib
    dev = torch.device("cuda:0")
    for (B, C, H, W) in SHAPES:
        torch.manual_seed(0)
        unit = FastFlowUnit(C, C, 3).to(dev)
        x = torch.randn(B, C, H, W, device=dev)
        nb = min(B, 4)
        xg = x[:nb].clone().requires_grad_(True)
        z, _ = unit(xg)
        gz = torch.randn_like(z)
        z.backward(gz)
        xd = x[:nb].double().requires_grad_(True)
        ref = torch.cat([F.conv2d(F.pad(c, m.pad), m.conv.weight.detach().double()) for m, c in
                         zip((unit.conv_tl, unit.conv_tr, unit.conv_bl, unit.conv_br), torch.chunk(xd, 4, 1))], 1)
        ref.backward(gz.double())
        err = float((z.detach().double() - ref.detach()).abs().max() / ref.detach().abs().max())
        gerr = float((xg.grad.double() - xd.grad).abs().max() / xd.grad.abs().max())
        with torch.no_grad():
            fn = lambda: unit(x)
            for _ in range(20): fn()
            torch.cuda.synchronize()
            a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            a.record()
            for _ in range(50): fn()
            b.record(); torch.cuda.synchronize()
            us = a.elapsed_time(b) / 50 * 1e3
        form = _lib.backward_variant(B, 4, C // 4, H, W, 3, 3)["conv_form"]
        print(f"B{B:4d} C{C:3d} {H:3d}x{W:3d} {form:10s}: forward {us:8.1f} us = {8 * B * C * H * W / us / 1e3 / 8000:.3f} of HBM peak; "
              f"err fwd {err:.1e} grad-input {gerr:.1e}", flush=True)
else:
    forms = os.environ.get("WINO_FORMS", "d,2,4,lib").split(",")
    for env in [e for k, e in (("d", {"FINC_NO_WINO": "1"}), ("2", {"FINC_WINO_FORM": "2"}), ("4", {"FINC_WINO_FORM": "4"}), ("lib", {})) if k in forms]:
        print("==", env or "library's choice", flush=True)
        subprocess.run([sys.executable, os.path.abspath(__file__), "child"], env=dict(os.environ, **env), check=True)
