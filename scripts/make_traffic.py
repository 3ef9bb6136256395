"""profiles/traffic_<workload>.json from a PMC summary (scripts/pmc_run.sh -> summary.json): the HBM bytes per launch of the
workload's inverse kernel, which bench.py puts into `roofline.traffic`.      python scripts/make_traffic.py <workload> <summary.json>

FETCH_SIZE / WRITE_SIZE are in KiB.  gfx950's FETCH_SIZE tallies a 128-byte line fill at 64 bytes (MI355X_MICROARCH.md, HBM
section: double it for wide coalesced streaming reads).  The inverse kernels do not make such requests: their loads are 16-, 32- or
64-byte pieces of one row of one channel, every TCC_EA0_RDREQ is a 64-byte request (RDREQ x 64 B = FETCH_SIZE; calibration
profiles/r01/v1b_*: 402.9 MB read for 402.7 MB loaded), so NO x2 is applied to the inverse -- stated per kernel in `rule`."""
import json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
workload, path = sys.argv[1], sys.argv[2]
from bench import WORKLOADS
B, C, H, W, K, std = WORKLOADS[workload]
s = json.load(open(path))
# the workload's inverse kernel: the wavefront kernel ("inverse"), else the role-split / short-step one
inv = s.get("inverse") or s.get("inverse_chain") or s.get("inverse_split") or s.get("stream_inverse")
kernel_class = "inverse" if "inverse" in s else "inverse_chain" if "inverse_chain" in s else "inverse_split" if "inverse_split" in s else "stream_inverse"
# (the streaming-bank kernel, finc_stream.hip: register loads of 16-byte pieces, one per line -- TCC_EA0_RDREQ x 64 B = FETCH_SIZE there
# too (8.12e7 x 64 B = 5.2 GB against FETCH_SIZE 5.0 GB at 192 channels, profiles/r05/stream): no x2)
fetch, write = inv["FETCH_SIZE"] * 1024.0, inv["WRITE_SIZE"] * 1024.0
# the short-step kernel (finc_chain.hip) brings z in by LDS-DMA, 16 bytes per lane: the guide's case of a 128-byte line fill tallied at
# 64 bytes (TCC_EA0_RDREQ x 128 B = the image, x 64 B = half of it) -- its FETCH_SIZE is doubled; every other inverse kernel: as read
x2 = kernel_class == "inverse_chain"
if x2:
    fetch *= 2.0
alg = 8 * B * C * H * W + 4 * C * (C // 4) * K * K
out = {
    "workload": workload,
    "kernel_class": kernel_class,
    "inverse_hbm_bytes_per_launch": int(round(fetch + write)),
    "inverse_fetch_bytes": int(round(fetch)),
    "inverse_write_bytes": int(round(write)),
    "algorithmic_bytes": alg,
    "ratio": (fetch + write) / alg,
    "rdreq": inv.get("TCC_EA0_RDREQ_sum"), "rdreq_32B": inv.get("TCC_EA0_RDREQ_32B_sum"),
    "wrreq": inv.get("TCC_EA0_WRREQ_sum"), "wrreq_64B": inv.get("TCC_EA0_WRREQ_64B_sum"),
    "lds_bank_conflict_cycles_inverse": inv.get("SQ_LDS_BANK_CONFLICT"), "lds_active_cycles_inverse": inv.get("SQ_LDS_IDX_ACTIVE"),
    "rule": ("FETCH_SIZE x 2: LDS-DMA loads of 16 bytes per lane are 128-byte line fills, which gfx950 tallies at 64 bytes (MI355X_MICROARCH.md, HBM section)"
             if x2 else "no x2 on FETCH_SIZE: the inverse loads 16/32/64-byte pieces, TCC_EA0_RDREQ x 64 B = FETCH_SIZE (see the file's header)"),
    "source": f"{path} (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE, one counter per pass: "
              + ("scripts/pmc_stream_bytes.sh" if kernel_class == "stream_inverse" else "scripts/pmc_run.sh via scripts/profile_round.sh") + "; KiB -> bytes)",
}
dst = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "profiles", f"traffic_{workload}.json")
if os.path.exists(dst):                        # (keep what else the file holds: the forward / grad-weight notes of c3)
    old = json.load(open(dst))
    for k, v in old.items():
        out.setdefault(k, v)
json.dump(out, open(dst, "w"), indent=1)
print(dst, "ratio %.3f" % out["ratio"])
