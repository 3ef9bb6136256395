"""GPU helper: interleaved A/B of several libfinc builds in ONE process is not possible (one ctypes handle per
process), so this runs each library in a subprocess, ROUNDS times alternating, on the same box, and prints the
median steady-state launch time of the c3 inverse / forward."""
import os, subprocess, sys, statistics
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CODE = r'''
import sys, os, time, torch
sys.path.insert(0, %r)
from fincflow_amd import FastFlowUnit
dev = torch.device("cuda:0"); torch.manual_seed(0)
B = int(os.environ.get("AB_BATCH", "256"))
unit = FastFlowUnit(96, 96, 3).to(dev); x = torch.randn(B, 96, 64, 64, device=dev)
with torch.no_grad():
    z, _ = unit(x); o = torch.empty_like(z)
    res = []
    for fn in (lambda: unit._cache.inverse(z, unit._weights(), 4, 0xE4, out=o), lambda: unit._cache.forward(x, unit._weights(), 4, 0xE4, out=o)):
        t_end = time.perf_counter() + 0.3
        while time.perf_counter() < t_end:
            for _ in range(10): fn()
            torch.cuda.synchronize()
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        for _ in range(200): fn()
        b.record(); torch.cuda.synchronize()
        res.append(a.elapsed_time(b) / 200 * 1e3)
print("%%.1f %%.1f" %% tuple(res))
''' % REPO
libs = sys.argv[1:]
rounds = int(os.environ.get("AB_ROUNDS", "3"))
out = {l: [] for l in libs}
for r in range(rounds):
    for l in libs:
        env = dict(os.environ, FINCFLOW_LIB=os.path.abspath(l))
        p = subprocess.run([sys.executable, "-c", CODE], env=env, capture_output=True, text=True)
        vals = p.stdout.strip().split()
        if len(vals) == 2:
            out[l].append((float(vals[0]), float(vals[1])))
        else:
            print("FAILED", l, p.stderr[-300:])
for l in libs:
    if out[l]:
        print(os.path.basename(l), "inv median %.1f us (all %s) | fwd median %.1f us" % (
            statistics.median(v[0] for v in out[l]), " ".join("%.0f" % v[0] for v in out[l]), statistics.median(v[1] for v in out[l])))
