"""GPU helper: the inverse with a remainder launch, many times over -- c3's layer at B = 264 / 288 / 320 (the remainder's 8 / 32 / 64 images run the role-split
kernel, at 8 and 32 images with its bands dealt out to idle compute units, right behind a kernel that filled the chip).  Every launch must give the first
launch's result bit for bit; no protocol wait may give up.  Usage: soak_remainder.py [launches per batch size]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from fincflow_amd import FastFlowUnit, _lib
n = int(sys.argv[1]) if len(sys.argv) > 1 else 1000
dev = torch.device("cuda:0"); torch.manual_seed(0)
unit = FastFlowUnit(96, 96, 3).to(dev)
for B in (264, 288, 320):
    assert _lib.inverse_remainder_images(B, 4, 24, 64, 64, 3, 3) == B - 256
    x = torch.randn(B, 96, 64, 64, device=dev)
    with torch.no_grad():
        z, _ = unit(x)
        first = unit.reverse(z).clone()
        err = ((first - x).abs().max() / x.abs().max()).item()
        out = torch.empty_like(z); diff = 0
        for k in range(n):
            r = unit._cache.inverse(z, unit._weights(), 4, 0xE4, out=out)
            if k % 50 == 49:
                diff += int(not torch.equal(out, first))
        torch.cuda.synchronize()
        diff += int(not torch.equal(out, first))
    print("B=%d: %d launches, round-trip rel err %.2e, results differing from the first (checked every 50th and the last): %d, waits given up: %d, fault pending: %s"
          % (B, n, err, diff, _lib.hlp_timeouts(), _lib.fault_pending()), flush=True)
    assert diff == 0 and _lib.hlp_timeouts() == 0 and not _lib.fault_pending() and err <= 1e-5
print("soak ok")
