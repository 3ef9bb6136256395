"""GPU helper (diagnostic build -DFINC_SPLIT_STAMP): busy cycles per step of each wave of the role-split inverse."""
import ctypes, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from fincflow_amd import FastFlowUnit, _lib
dev = torch.device("cuda:0")
B, C, H, W, K = (int(a) for a in sys.argv[1:6]) if len(sys.argv) > 5 else (8, 96, 64, 64, 3)
unit = FastFlowUnit(C, C, K).to(dev)
z = torch.randn(B, C, H, W, device=dev)
with torch.no_grad():
    for _ in range(5):
        unit.reverse(z)
torch.cuda.synchronize()
buf = (ctypes.c_ulonglong * 16)()
L = _lib.lib()
L.finc_debug_split_stamps.argtypes = [ctypes.c_void_p]
assert L.finc_debug_split_stamps(buf) == 0
steps = buf[9]
print(f"steps {steps}: s_memtime ticks per step busy: A {buf[0]/steps:.0f}  B0 {buf[1]/steps:.0f}  B1 {buf[2]/steps:.0f}  B2 {buf[3]/steps:.0f} | loop total (cycle counter) per step {buf[8]/steps:.0f}")
