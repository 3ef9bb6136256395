#!/usr/bin/env python3
"""Static check of the built library (no GPU): no wide buffer store is followed, within two wait states, by an instruction that
overwrites the store's data registers.

Why: a buffer_store_dwordx3/x4 reads its data registers a few cycles after issue.  The compiler inserts the wait states (s_nop 1)
itself when the store's SCALAR offset is an immediate, but treats the form with an SGPR offset as hazard-free -- and on gfx950 it
is not: `buffer_store_dwordx4 v[112:115], v136, s[24:27], s34 offen` followed directly by `v_pk_add_f32 v[112:113], ...` stored
a stale second dword from the lanes 12..15 of every row (round 3, the F(4,3) forward at Cq = 12).  Inline-asm stores are the
author's business entirely.  This script disassembles every gfx950 kernel and reports each such pair; `tests/test_abi.py` runs
it on the in-tree library.  `check_store_hazard.py [lib]` prints the findings and exits 1 if there are any."""
import os, re, shutil, subprocess, sys, tempfile

LLVM = "/opt/rocm/lib/llvm/bin"
WIDE = re.compile(r"^\s*(buffer_store_dwordx[34]|global_store_dwordx[34]|flat_store_dwordx[34]|scratch_store_dwordx[34])\s+(.*)$")
REG = re.compile(r"\b([va])\[(\d+):(\d+)\]|\b([va])(\d+)\b")


def regs(tok):
    out = set()
    for m in REG.finditer(tok):
        if m.group(1):
            out |= {(m.group(1), i) for i in range(int(m.group(2)), int(m.group(3)) + 1)}
        else:
            out.add((m.group(4), int(m.group(5))))
    return out


def wait_states(ins):
    m = re.match(r"s_nop\s+(\d+)", ins)
    return int(m.group(1)) + 1 if m else 1


def written(ins):
    """registers an instruction writes: the first operand of VALU / MFMA / LDS-read / load instructions (both operands of swaps)"""
    op, _, rest = ins.partition(" ")
    if op.startswith(("s_", "buffer_store", "global_store", "flat_store", "ds_write", "scratch_store")) or not rest:
        return set()
    ops = [o.strip() for o in rest.split(",")]
    w = regs(ops[0])
    if "swap" in op and len(ops) > 1:
        w |= regs(ops[1])
    return w


def scan(text):
    findings, kernel, lines = [], "?", []
    for raw in text.splitlines():
        m = re.match(r"^[0-9a-f]+ <(\S+)>:", raw)
        if m:
            kernel = m.group(1)
            continue
        ins = raw.split("//")[0].strip()
        if ins:
            lines.append((kernel, ins))
    for i, (kernel, ins) in enumerate(lines):
        m = WIDE.match(ins)
        if not m:
            continue
        ops = [o.strip() for o in m.group(2).split(",")]
        data = regs(ops[1] if m.group(1).startswith(("global", "flat", "scratch")) else ops[0])
        ws, j = 0, i + 1
        while j < len(lines) and ws < 2 and lines[j][0] == kernel:
            nxt = lines[j][1]
            if nxt.startswith("s_nop"):
                ws += wait_states(nxt)
            else:
                if written(nxt) & data:
                    findings.append((kernel, ins, nxt, ws))
                    break
                if nxt.startswith(("s_cbranch", "s_branch", "s_endpgm")):
                    break                      # (a branch target is checked from its own predecessors' side only)
                ws += 1
            j += 1
    return findings


def main(lib=None):
    lib = lib or os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "fincflow_amd", "libfinc_hip.so")
    bad = []
    with tempfile.TemporaryDirectory() as d:
        shutil.copy(lib, os.path.join(d, "l.so"))
        subprocess.run([f"{LLVM}/llvm-objdump", "--offloading", "l.so"], cwd=d, stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
        objs = [f for f in sorted(os.listdir(d)) if "gfx950" in f]
        assert objs, "no gfx950 code object in " + lib
        n = 0
        for f in objs:
            text = subprocess.run([f"{LLVM}/llvm-objdump", "-d", "--no-show-raw-insn", f], cwd=d, capture_output=True, text=True).stdout
            n += sum(1 for l in text.splitlines() if WIDE.match(l.split("//")[0]))
            bad += scan(text)
    print(f"{n} wide stores checked, {len(bad)} followed too closely by a write of their data registers")
    for k, st, nx, ws in bad:
        dem = subprocess.run(["c++filt", k], capture_output=True, text=True).stdout.strip()[:90]
        print(f"  {dem}\n      {st}\n      {nx}        ({ws} wait state(s) between)")
    return 1 if bad else 0


if __name__ == "__main__":
    sys.exit(main(sys.argv[1] if len(sys.argv) > 1 else None))
