"""GPU helper: run the c3 inverse once with the stamped build (scripts/stamp.sh) and print where one wave's step goes.
Usage: FINCFLOW_LIB=ablate_build/libfinc_stamp.so python scripts/stamp_read.py"""
import ctypes, os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from fincflow_amd import FastFlowUnit, _lib
dev = torch.device("cuda:0"); torch.manual_seed(0)
unit = FastFlowUnit(96, 96, 3).to(dev); x = torch.randn(int(os.environ.get("AB_BATCH", "256")), 96, 64, 64, device=dev)
with torch.no_grad():
    z, _ = unit(x)
    for _ in range(30): unit.reverse(z)
    torch.cuda.synchronize()
L = _lib.lib()
buf = (ctypes.c_ulonglong * 40)()
L.finc_debug_stamps.argtypes = [ctypes.c_void_p, ctypes.c_int]
assert L.finc_debug_stamps(buf, 40) == 0
names = ["phaseA", "z-term", "c0+post1", "c1+io", "c2+post2", "c3", "c4", "c5", "advance", "latch"]
nwin = 32.0
tot = 0
for ph in range(4):
    row = [buf[ph * 10 + k] / nwin for k in range(10)]
    tot += sum(row)
    print("step%d: " % ph + "  ".join("%s %6.0f" % (n, v) for n, v in zip(names, row)) + "   | sum %6.0f" % sum(row))
print("cycles per window %.0f (s_memtime ticks; stamp overhead ~40 each included)" % tot)
