"""Where does a kernel's SGPR spill code (v_writelane / v_readlane) sit?  Builds finc_split.hip to assembly and reports, per loop
of a kernel, the barriers and spill operations inside it -- spill code in a loop with barriers that is not the outermost (per-band)
loop would be in the path of a step.      python scripts/spill_sites.py [kernel-name-regex]"""
import os, re, subprocess, sys, tempfile
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
pat = re.compile(sys.argv[1] if len(sys.argv) > 1 else r"finc_split_kernelILi24ELi3ELi3ELi3ELb1E")
out = os.path.join(tempfile.mkdtemp(), "split.s")
subprocess.check_call(["hipcc", "-O3", "-fPIC", "--offload-arch=gfx950", "-std=c++20", "-mllvm", "-amdgpu-mfma-vgpr-form", "--cuda-device-only",
                       "-S", os.path.join(root, "fincflow_amd", "csrc", "finc_split.hip"), "-o", out], stderr=subprocess.DEVNULL)
lines = open(out).read().split("\n")
start = next(i for i, l in enumerate(lines) if re.match(r"^_ZN.*:", l) and pat.search(l))
end = next(i for i in range(start, len(lines)) if "s_endpgm" in lines[i])
lines = lines[start:end + 1]
spill = [i for i, l in enumerate(lines) if "v_writelane" in l or "v_readlane" in l]
bar = [i for i, l in enumerate(lines) if "s_barrier" in l]
labels = {m.group(1): i for i, l in enumerate(lines) for m in [re.match(r"^(\.LBB\d+_\d+):", l)] if m}
loops = set()
for i, l in enumerate(lines):
    m = re.search(r"s_c?branch\w* (\.LBB\d+_\d+)", l)
    if m and m.group(1) in labels and labels[m.group(1)] < i:
        loops.add((labels[m.group(1)], i))
print(f"{len(lines)} lines, {len(spill)} spill operations, {len(bar)} barriers")
worst = 0
with_bar = [(a, b) for a, b in sorted(loops) if any(a <= s <= b for s in bar)]
# the per-band loop: the loops that share the first head and enclose the step loops (several back edges: one per role)
head = min(a for a, b in with_bar if any(a2 > a and b2 < b for a2, b2 in with_bar))
for a, b in with_bar:
    ns, nb = sum(a <= s <= b for s in spill), sum(a <= s <= b for s in bar)
    band_loop = a == head
    print(f"loop {a}..{b}: {nb} barriers, {ns} spill operations{'   <- the per-band loop' if band_loop else ''}")
    if not band_loop: worst = max(worst, ns)
print("spill operations inside a step loop:", worst)
sys.exit(1 if worst else 0)
