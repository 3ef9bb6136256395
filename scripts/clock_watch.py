"""GPU helper: what shader clock and power does the card run at while the c3 forward / inverse loop?  Launches the loop
in this process and samples `rocm-smi` (a child process that never touches HIP) once a second beside it.
    python scripts/clock_watch.py [forward|inverse|idle] [seconds]"""
import os, subprocess, sys, threading, time
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
what = sys.argv[1] if len(sys.argv) > 1 else "forward"
secs = float(sys.argv[2]) if len(sys.argv) > 2 else 8.0
stop = False
samples = []


def sampler():
    while not stop:
        p = subprocess.run(["rocm-smi", "--showclocks", "--showpower", "--showtemp", "--csv"], capture_output=True, text=True)
        samples.append((time.perf_counter(), p.stdout.strip() or p.stderr.strip()[-300:]))
        time.sleep(1.0)


import torch
from fincflow_amd import FastFlowUnit
dev = torch.device("cuda:0"); torch.manual_seed(0)
unit = FastFlowUnit(96, 96, 3).to(dev); x = torch.randn(256, 96, 64, 64, device=dev)
with torch.no_grad():
    z, _ = unit(x); o = torch.empty_like(z)
    fn = {"inverse": lambda: unit._cache.inverse(z, unit._weights(), 4, 0xE4, out=o),
          "forward": lambda: unit._cache.forward(x, unit._weights(), 4, 0xE4, out=o), "idle": lambda: time.sleep(0.01)}[what]
    th = threading.Thread(target=sampler); th.start()
    t_end = time.perf_counter() + secs
    n = 0
    t0 = time.perf_counter()
    while time.perf_counter() < t_end:
        for _ in range(50): fn()
        torch.cuda.synchronize(); n += 50
    dt = time.perf_counter() - t0
    stop = True; th.join()
print(what, "launches", n, "mean us", dt / n * 1e6)
for t, s in samples:
    print("--", round(t - t0, 1)); print(s)
