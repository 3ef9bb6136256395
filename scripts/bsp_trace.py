"""GPU helper (diagnostic build -DFINC_BSP_TRACE, ablate_build/libfinc_trace.so): per-job records of band-split launches --
which workgroup solved which band of which problem, on which XCD, from when to when.   bsp_trace.py B C H W K [launches]"""
import ctypes, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ.setdefault("FINCFLOW_LIB", "ablate_build/libfinc_trace.so")
import numpy as np
import torch
from fincflow_amd import FastFlowUnit, _lib
dev = torch.device("cuda:0")
B, C, H, W, K = (int(a) for a in sys.argv[1:6])
launches = int(sys.argv[6]) if len(sys.argv) > 6 else 6
torch.manual_seed(0)
unit = FastFlowUnit(C, C, K).to(dev)
x = torch.randn(B, C, H, W, device=dev)
L = _lib.lib()
buf = (ctypes.c_ulonglong * (1 + 4 * 4096))()
nprob = B * 4
with torch.no_grad():
    z, _ = unit(x)
    o = torch.empty_like(z)
    fn = lambda: unit._cache.inverse(z, unit._weights(), 4, 0xE4, out=o)
    for _ in range(20): fn()
    torch.cuda.synchronize()
    for it in range(launches):
        L.finc_debug_bsp_trace(buf, 1)
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record(); fn(); b.record(); torch.cuda.synchronize()
        L.finc_debug_bsp_trace(buf, 0)
        n = int(buf[0])
        r = np.array(buf[1:1 + 4 * n], dtype=np.uint64).reshape(n, 4)
        job, meta, t0, t1 = r[:, 0].astype(int), r[:, 1].astype(int), r[:, 2].astype(np.int64), r[:, 3].astype(np.int64)
        band, prob = job // nprob, job % nprob
        blk, xcc, nb = meta & 0xFFFF, (meta >> 16) & 15, meta >> 24
        base = t0.min()
        s, e = (t0 - base) / 100.0, (t1 - base) / 100.0          # us (100 MHz)
        owner = {(int(band[i]), int(prob[i])): i for i in range(n)}
        for i in range(n):                                        # bands taken by extension belong to the same record
            if nb[i] == 2: owner[(int(band[i]) + 2, int(prob[i]))] = i
        same = sum(1 for (bd, pr), i in owner.items() if bd >= 1 and (bd - 1, pr) in owner and xcc[owner[(bd - 1, pr)]] == xcc[i])
        pairs = sum(1 for (bd, pr) in owner if bd >= 1 and (bd - 1, pr) in owner)
        print(f"launch {it}: {a.elapsed_time(b) * 1e3:7.1f} us  jobs {n} (extended {int((nb == 2).sum())})  last end {e.max():6.1f} us  "
              f"consumer on its producer's XCD {same}/{pairs}")
        flag = np.array([(int(band[i]) ^ 1, int(prob[i])) in owner and xcc[owner[(int(band[i]) ^ 1, int(prob[i]))]] == xcc[i] for i in range(n)])
        if flag.any() and (~flag).any():
            ln = e - s
            print(f"   jobs whose partner (the other bands of the problem) runs on the SAME XCD: {int(flag.sum())}, length {ln[flag].mean():6.1f} (max {ln[flag].max():6.1f}); "
                  f"on another XCD: {int((~flag).sum())}, length {ln[~flag].mean():6.1f} (max {ln[~flag].max():6.1f})")
        ln = e - s
        if ln.max() > np.median(ln) + 6 or os.environ.get("BSP_TRACE_XCC"):
            print("      per XCD (jobs, of them with the partner on the same XCD, mean / max length): " + "  ".join(
                f"{x}: {int((xcc == x).sum())} {int((flag & (xcc == x)).sum())} {ln[xcc == x].mean():.1f}/{ln[xcc == x].max():.1f}" for x in range(8) if (xcc == x).any()))
        if ln.max() > np.median(ln) + 6:
            for i in np.argsort(-ln)[:6]:
                j = owner.get((int(band[i]) ^ 1, int(prob[i])))
                print(f"      slow job: band {band[i]} problem {prob[i]:3d} block {blk[i]:3d} xcc {xcc[i]} start {s[i]:6.1f} end {e[i]:6.1f} length {ln[i]:6.1f}"
                      + (f" | partner band {band[j]} block {blk[j]:3d} xcc {xcc[j]} start {s[j]:6.1f} end {e[j]:6.1f} length {ln[j]:6.1f}" if j is not None else ""))
        for bd in sorted(set(band.tolist())):
            m = band == bd
            print(f"   first band {bd}: {int(m.sum()):4d} jobs  start {s[m].min():6.1f} .. {s[m].max():6.1f}  end {e[m].min():6.1f} .. {e[m].max():6.1f}  "
                  f"length {np.median(e[m] - s[m]):6.1f} (max {(e[m] - s[m]).max():6.1f})  bands/job {nb[m].mean():.2f}")
