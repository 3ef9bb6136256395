"""GPU helper (diagnostic build -DFINC_SPLIT_STAMP of finc_chain.hip): busy cycles per step of each wave of the short-step inverse."""
import ctypes, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from fincflow_amd import FastFlowUnit, _lib
dev = torch.device("cuda:0")
B, C, H, W, K = (int(a) for a in sys.argv[1:6]) if len(sys.argv) > 5 else (64, 48, 32, 32, 3)
unit = FastFlowUnit(C, C, K).to(dev)
z = torch.randn(B, C, H, W, device=dev)
with torch.no_grad():
    for _ in range(5):
        unit.reverse(z)
torch.cuda.synchronize()
buf = (ctypes.c_ulonglong * 32)()
L = _lib.lib()
L.finc_debug_chain_stamps.argtypes = [ctypes.c_void_p]
assert L.finc_debug_chain_stamps(buf) == 0
steps = buf[9]
print(f"steps {steps}: s_memtime ticks per step busy: A {buf[0]/steps:.0f}  " + "  ".join(f"B{i} {buf[1+i]/steps:.0f}" for i in range(5)),
      "| prologue, loop:", [int(b) for b in buf[10:12]], "| A segs/step", [round(b / steps) for b in buf[16:19]], "| B1 segs/step", [round(b / steps) for b in buf[20:23]])
