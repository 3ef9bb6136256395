"""GPU helper: the one-wave streaming-bank inverse at ONE problem count, in the form FINC_STREAM_ONE_WAVE_VEC selects (0 dword, 1 per-lane
16-byte): time_stream_crossover.py B   (G = 4; three banks).  A/B runs only: the switch shows up in finc_runtime_switches()."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from fincflow_amd import ops
from oracle import oracle
dev = torch.device("cuda:0")
B = int(sys.argv[1])
out = []
for (Cq, KH, KW) in ((12, 4, 4), (12, 7, 7), (40, 2, 2)):
    std = (0.05 if max(KH, KW) < 5 else 0.02) * min(1.0, (24.0 / Cq) ** 0.5)
    ws = torch.from_numpy(oracle.make_stored_weights(4, Cq, KH, KW, orient=0xE4, seed=1, std=std)).to(dev)
    per = ws.shape[0] // 4
    weights = [ws[i * per:(i + 1) * per].clone() for i in range(4)]
    cache = ops.PackedWeights()
    x = torch.randn(B, 4 * Cq, 32, 32, device=dev)
    with torch.no_grad():
        z = cache.forward(x, weights, 4, 0xE4)
        o = torch.empty_like(z)
        for _ in range(5): cache.inverse(z, weights, 4, 0xE4, out=o)
        torch.cuda.synchronize()
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        for _ in range(30): cache.inverse(z, weights, 4, 0xE4, out=o)
        b.record(); torch.cuda.synchronize()
    out.append("k%dx%d Cq%d %7.1f us" % (KH, KW, Cq, a.elapsed_time(b) / 30 * 1e3))
print("form %s, %4d problems: %s" % (os.environ.get("FINC_STREAM_ONE_WAVE_VEC", "library"), 4 * B, "  ".join(out)), flush=True)
