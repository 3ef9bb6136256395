"""GPU helper (diagnostic build -DFINC_HLP_COUNT): how often does the compute wave find its helper late?"""
import ctypes, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from fincflow_amd import FastFlowUnit, _lib
dev = torch.device("cuda:0"); torch.manual_seed(0)
unit = FastFlowUnit(96, 96, 3).to(dev); x = torch.randn(256, 96, 64, 64, device=dev)
n = 50
with torch.no_grad():
    z, _ = unit(x)
    for _ in range(n): unit.reverse(z)
    torch.cuda.synchronize()
L = _lib.lib(); buf = (ctypes.c_uint * 2)()
L.finc_debug_hlp_late.argtypes = [ctypes.c_void_p]
assert L.finc_debug_hlp_late(buf) == 0
waits = n * 1024 * 68
print(f"late landings {buf[0]} ({buf[0]/waits:.1%} of {waits} checks), late x-ring reads {buf[1]} ({buf[1]/waits:.1%}); timeouts {_lib.hlp_timeouts()}")
