#!/usr/bin/env python3
"""Register / LDS / scratch metadata of the kernels in a built libfinc_hip.so (no GPU): `kernel_regs.py [regex] [lib]`."""
import os, re, shutil, subprocess, sys, tempfile
LLVM = "/opt/rocm/lib/llvm/bin"
pat = re.compile(sys.argv[1] if len(sys.argv) > 1 else ".")
lib = sys.argv[2] if len(sys.argv) > 2 else os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "fincflow_amd", "libfinc_hip.so")
with tempfile.TemporaryDirectory() as d:
    shutil.copy(lib, os.path.join(d, "l.so"))
    subprocess.run([f"{LLVM}/llvm-objdump", "--offloading", "l.so"], cwd=d, stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
    for f in sorted(os.listdir(d)):
        if "gfx950" not in f:
            continue
        notes = subprocess.run([f"{LLVM}/llvm-readelf", "--notes", f], cwd=d, capture_output=True, text=True).stdout
        for blk in notes.split("- .agpr_count:")[1:]:
            name = re.search(r"\.name:\s+(\S+)", blk).group(1)
            dem = subprocess.run(["c++filt", name], capture_output=True, text=True).stdout.strip()
            if not pat.search(dem):
                continue
            g = lambda k: int(re.search(r"\.%s:\s+(\d+)" % k, blk).group(1))
            print(f"{dem[:100]:100s} agpr {int(blk.split()[0]):3d} total {g('vgpr_count'):3d} spill {g('vgpr_spill_count')} "
                  f"scratch {g('private_segment_fixed_size')} maxflat {g('max_flat_workgroup_size')}")
