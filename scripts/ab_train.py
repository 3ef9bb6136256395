"""GPU helper: training step (forward + grad-input + grad-weight) of the c3 unit with several library builds, alternating
(child processes: one ctypes handle per process): python scripts/ab_train.py lib1.so lib2.so ..."""
import os, subprocess, sys, statistics
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CODE = r'''
import sys, os, time, torch
sys.path.insert(0, %r)
from fincflow_amd import FastFlowUnit
dev = torch.device("cuda:0"); torch.manual_seed(0)
B, C, H, W = (int(v) for v in os.environ.get("AB_SHAPE", "256,96,64,64").split(","))
unit = FastFlowUnit(C, C, 3).to(dev); x = torch.randn(B, C, H, W, device=dev)
xg = x.clone().requires_grad_(True); z, _ = unit(xg); gz = torch.randn_like(z)
def train():
    xg.grad = None
    for p_ in unit.parameters(): p_.grad = None
    zz, _ = unit(xg); zz.backward(gz)
t_end = time.perf_counter() + 0.5
while time.perf_counter() < t_end:
    for _ in range(5): train()
    torch.cuda.synchronize()
a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
a.record()
for _ in range(100): train()
b.record(); torch.cuda.synchronize()
print("%%.1f" %% (a.elapsed_time(b) / 100 * 1e3))
''' % REPO
libs = sys.argv[1:]
out = {l: [] for l in libs}
for r in range(int(os.environ.get("AB_ROUNDS", "3"))):
    for l in libs:
        p = subprocess.run([sys.executable, "-c", CODE], env=dict(os.environ, FINCFLOW_LIB=os.path.abspath(l)), capture_output=True, text=True)
        try: out[l].append(float(p.stdout.strip().split()[-1]))
        except Exception: print("FAILED", l, p.stderr[-300:])
for l in libs:
    if out[l]: print(os.path.basename(l), "training step median %.1f us (all %s)" % (statistics.median(out[l]), " ".join("%.0f" % v for v in out[l])))
