"""GPU debug helper: per-channel error of the MFMA kernels on an identity filter bank."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from fincflow_amd import ops
from oracle import oracle
dev = torch.device("cuda:0")
for Cq in (4, 8, 24):
    H = W = 16
    w = np.zeros((Cq, Cq, 3, 3), np.float32)
    for c in range(Cq):
        w[c, c, 2, 2] = 1.0
    wc = torch.from_numpy(w).to(dev)
    x = torch.arange(Cq, dtype=torch.float32, device=dev).view(1, Cq, 1, 1).expand(1, Cq, H, W).contiguous() * 100 \
        + torch.arange(H * W, dtype=torch.float32, device=dev).view(1, 1, H, W) * 0.01
    for name, fn in (("fwd", ops.finc_forward), ("inv", ops.finc_inverse)):
        y = fn(x, wc, G=1, orient=0, algo="mfma")
        torch.cuda.synchronize()
        d = (y - x).abs().amax(dim=(0, 2, 3)).cpu().numpy()
        print(Cq, name, "per-channel max err:", np.round(d, 2))
        bad = np.nonzero(d > 1e-3)[0]
        for c in bad[:6]:
            print("   ch", c, "got[0,0..3]", y[0, c, 0, :4].cpu().numpy(), "want", x[0, c, 0, :4].cpu().numpy())
