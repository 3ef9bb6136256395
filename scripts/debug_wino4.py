"""GPU helper: where the F(4,3) forward differs from fp64 conv2d (per channel / per column quad)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ.setdefault("FINC_WINO_FORM", "4")
import torch
import torch.nn.functional as F
from fincflow_amd import FastFlowUnit
B, C, H, W = (int(v) for v in (sys.argv[1:5] if len(sys.argv) > 4 else (1, 48, 4, 64)))
dev = torch.device("cuda:0")
torch.manual_seed(0)
unit = FastFlowUnit(C, C, 3).to(dev)
x = torch.randn(B, C, H, W, device=dev)
with torch.no_grad():
    z, _ = unit(x)
    ref = torch.cat([F.conv2d(F.pad(c.double(), m.pad), m.conv.weight.detach().double()) for m, c in
                     zip((unit.conv_tl, unit.conv_tr, unit.conv_bl, unit.conv_br), torch.chunk(x, 4, 1))], 1)
e = (z.double() - ref).abs() / ref.abs().max()
print("per channel:", " ".join(f"{v:.0e}" for v in e.amax(dim=(0, 2, 3)).tolist()))
print("per column :", " ".join(f"{v:.0e}" for v in e.amax(dim=(0, 1, 2)).tolist()))
print("per row    :", " ".join(f"{v:.0e}" for v in e.amax(dim=(0, 1, 3)).tolist()))
g0 = slice(0, C // 4)
zz, rr = z[0, g0].double().cpu(), ref[0, g0].cpu()
bad = ((zz - rr).abs() > 1e-3 * rr.abs().max()).nonzero()
print("bad elements in group 0:", bad.shape[0], "first:", bad[:6].tolist())
for (c, h, w) in bad[:4].tolist():
    print(f" ch {c} row {h} col {w}: got {zz[c, h, w]:+.5f} want {rr[c, h, w]:+.5f}; want at col-1 {rr[c, h, w - 1]:+.5f} col+1 {rr[c, h, w + 1]:+.5f}; "
          f"same col ch-4 {rr[c - 4, h, w]:+.5f} ch+4 {rr[min(c + 4, C // 4 - 1), h, w]:+.5f}; col+48 {rr[c, h, (w + 48) % W]:+.5f} got-want {zz[c,h,w]-rr[c,h,w]:+.5f}")
