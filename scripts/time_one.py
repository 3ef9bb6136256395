"""GPU helper: time the inverse for one shape with the library selected by FINCFLOW_LIB: time_one.py B C H W K"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from fincflow_amd import FastFlowUnit, _lib
dev = torch.device("cuda:0")
B, C, H, W, K = (int(a) for a in sys.argv[1:6])
torch.manual_seed(0)
unit = FastFlowUnit(C, C, K).to(dev)
x = torch.randn(B, C, H, W, device=dev)
with torch.no_grad():
    z, _ = unit(x)
    o = torch.empty_like(z)
    fn = lambda: unit._cache.inverse(z, unit._weights(), 4, 0xE4, out=o)
    for _ in range(30): fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(100): fn()
    b.record(); torch.cuda.synchronize()
    us = a.elapsed_time(b) / 100 * 1e3
    err = float((o - x).abs().max() / x.abs().max())
v = _lib.inverse_variant(B, 4, C // 4, H, W, K, K)
print(f"{os.environ.get('FINCFLOW_LIB','default').split('/')[-1]} C{C} {H}x{W} B={B}: {us:7.1f} us bands {v.get('bands', 0)} nw {v['nw']} "
      f"chain {v.get('chain', 0)} err {err:.1e} timeouts {_lib.hlp_timeouts()}", flush=True)
