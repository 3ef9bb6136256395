"""The C-ABI library loads and exports every symbol include/finc.h declares (no GPU, no compute calls)."""
import ctypes
import os
import re

import pytest

from helpers import REPO
from fincflow_amd import _lib


def header_symbols():
    text = open(os.path.join(REPO, "include", "finc.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(finc_[a-z0-9_]+)\s*\(", text)))


def test_library_exists_and_loads():
    assert os.path.exists(_lib.LIB_PATH), "run __graft_entry__.build() first"
    assert _lib.lib().finc_version() >= 101
    # a product build carries none of the measurement knobs of csrc/finc_experiment.h (ablations, stamps, reduced tables):
    # one stray -D would still pass finc_version() and compute garbage
    assert _lib.build_flags() == 0


def test_every_declared_symbol_is_exported():
    syms = header_symbols()
    assert len(syms) >= 15
    raw = ctypes.CDLL(_lib.LIB_PATH)
    for s in syms:
        assert hasattr(raw, s), f"{s} declared in include/finc.h but not exported"
    assert sorted(_lib.SYMBOLS) == syms


def test_status_strings():
    L = _lib.lib()
    assert L.finc_status_string(0) == b"ok"
    seen = {L.finc_status_string(i) for i in range(8)}
    assert len(seen) == 8 and b"unknown status" not in seen
    assert L.finc_status_string(99) == b"unknown status"


def test_algo_selection_is_host_side_and_consistent():
    L = _lib.lib()
    MFMA, STRICT = 2, 1
    # the named configs: c2 (Cq=12, 32x32), c3 (Cq=24, 64x64) -> MFMA wavefront kernel
    assert L.finc_inverse_algo_for(12, 32, 32, 3, 3) == MFMA
    assert L.finc_inverse_algo_for(24, 64, 64, 3, 3) == MFMA
    assert L.finc_forward_algo_for(24, 64, 64, 3, 3) == MFMA
    # c5 (Cq=48, 5x5): 900 fragments do not fit one wave -> K-split over the 4 waves of a workgroup, still MFMA
    assert L.finc_inverse_algo_for(48, 128, 128, 5, 5) == MFMA
    assert L.finc_forward_algo_for(48, 128, 128, 5, 5) == MFMA
    # outside the register-resident tables (Cq=64, 7x7; CINCFlowUnit's 192 channels in one group) -> the streaming-bank kernel, still MFMA
    assert L.finc_inverse_algo_for(64, 32, 32, 7, 7) == MFMA and L.finc_forward_algo_for(64, 32, 32, 7, 7) == MFMA
    assert L.finc_inverse_algo_for(192, 64, 64, 3, 3) == MFMA and L.finc_forward_algo_for(192, 64, 64, 3, 3) == MFMA
    assert _lib.inverse_variant(256, 1, 192, 64, 64, 3, 3)["sec"] == 7 and _lib.inverse_variant(8, 4, 12, 32, 32, 4, 4)["nw"] == 1
    # one-wave streaming-bank problems: the dword form (row -3) while problems x padded channels <= 10,240 or the chip has at most one
    # problem per compute unit, the per-lane 16-byte form (row -4) beyond -- the crossovers of profiles/r05/stream/one_wave_crossover.txt;
    # a width that is no multiple of four and the four-wave problems never report it
    for (B, Cq, W, row) in ((160, 12, 32, -3), (161, 12, 32, -4), (64, 40, 32, -3), (65, 40, 32, -4), (80, 20, 32, -3), (81, 20, 32, -4),
                            (256, 12, 30, -3), (256, 100, 32, -3)):
        assert _lib.inverse_variant(B, 4, Cq, 32, W, 4 if Cq < 100 else 3, 4 if Cq < 100 else 3)["row"] == row, (B, Cq, W, row)
    # the 28-channel 3x3 bank (no two-wave form of its own) borrows the 32-channel bank's packed two-wave kernel at even problem counts up
    # to 512 (one round against one, 0.81 of the time: profiles/r05/notes/c28_borrowed_bank.txt) -- beyond that only on maps so wide that at
    # most two one-wave problems fit a compute unit; channel counts 25 .. 27 pad to the 28-channel bank and follow it
    own = _lib.inverse_variant(257, 1, 28, 64, 64, 3, 3)
    assert own["cqp"] == 28 and own["nw"] == 1 and own["npw"] == 1
    for (B, G, Cq, H, W, borrowed) in ((128, 4, 28, 64, 64, True), (129, 4, 28, 64, 64, False), (192, 4, 28, 64, 64, False), (193, 4, 28, 64, 64, False),
                                       (256, 4, 28, 64, 64, False), (512, 4, 28, 64, 64, False), (515, 1, 28, 64, 64, False), (128, 4, 25, 64, 64, True),
                                       (65, 4, 28, 32, 32, True), (129, 4, 28, 32, 32, False), (256, 4, 28, 32, 32, False), (128, 4, 32, 64, 64, False),
                                       (128, 4, 24, 64, 64, False), (256, 4, 28, 16, 128, True), (384, 4, 28, 16, 128, True), (256, 4, 28, 64, 80, False)):
        v = _lib.inverse_variant(B, G, Cq, H, W, 3, 3)
        assert (v["cqp"] == 32 and v["nw"] == 2 and v["npw"] == 2 and Cq < 29) == borrowed, (B, G, Cq, H, W, v)
        assert v["row"] != own["row"] if borrowed else True
    # beyond that kernel's limits (9x9 filter; a 7x7 bank of 96 channels whose step ring exceeds the LDS) -> reference-order kernel
    assert L.finc_inverse_algo_for(16, 32, 32, 9, 9) == STRICT and L.finc_forward_algo_for(16, 32, 32, 9, 9) == STRICT
    assert L.finc_inverse_algo_for(96, 32, 32, 7, 7) == STRICT
    # W not a multiple of 4 -> reference-order kernel
    assert L.finc_inverse_algo_for(24, 64, 63, 3, 3) == STRICT
    assert L.finc_workspace_bytes(4, 24, 3, 3) >= 4 * 108 * 64 * 4
    assert L.finc_workspace_bytes(4, 48, 5, 5) >= 256


def test_argument_validation_without_touching_the_gpu():
    """NULL pointers and bad dims are rejected before any HIP call."""
    L = _lib.lib()
    assert L.finc_inverse_f32(None, None, None, 1, 4, 1, 8, 8, 3, 3, 0xE4, 0, None, 0, None) == 1
    one = ctypes.c_void_p(16)
    assert L.finc_inverse_f32(one, one, ctypes.c_void_p(32), 0, 4, 1, 8, 8, 3, 3, 0xE4, 0, None, 0, None) == 2
    assert L.finc_inverse_f32(one, one, ctypes.c_void_p(32), 1, 17, 1, 8, 8, 3, 3, 0xE4, 0, None, 0, None) == 2
    assert L.finc_inverse_f32(ctypes.c_void_p(18), one, ctypes.c_void_p(32), 1, 4, 1, 8, 8, 3, 3, 0xE4, 0, None, 0, None) == 7
    assert L.finc_forward_f32(one, one, one, 1, 4, 1, 8, 8, 3, 3, 0xE4, 0, None, 0, None) == 2  # in == out
    # MFMA algo without a workspace
    assert L.finc_inverse_f32(one, one, ctypes.c_void_p(32), 1, 4, 24, 64, 64, 3, 3, 0xE4, 2, None, 0, None) == 4
    # MFMA algo on a shape it has no instantiation for
    assert L.finc_inverse_f32(one, one, ctypes.c_void_p(32), 1, 4, 16, 32, 32, 9, 9, 0xE4, 2, one, 1 << 30, None) == 3


def test_a_measurement_knob_without_the_experiment_gate_does_not_compile():
    """csrc/finc_experiment.h: -DFINC_ABLATE (or any other timing-only knob) without -DFINC_EXPERIMENT is a compile error, so a
    stray flag cannot produce a library that passes finc_version() and computes garbage; with the gate it compiles and
    reports itself through finc_build_flags()."""
    import shutil
    import subprocess
    if not shutil.which("hipcc"):
        pytest.skip("no hipcc")
    src = os.path.join(REPO, "fincflow_amd", "csrc", "finc_generic.hip")
    base = ["hipcc", "--offload-arch=gfx950", "-std=c++20", "--cuda-host-only", "-fsyntax-only", src]
    bad = subprocess.run(base + ["-DFINC_ABLATE=1"], capture_output=True, text=True)
    assert bad.returncode != 0 and "FINC_EXPERIMENT" in bad.stderr
    good = subprocess.run(base + ["-DFINC_ABLATE=1", "-DFINC_EXPERIMENT"], capture_output=True, text=True)
    assert good.returncode == 0, good.stderr[-500:]


def test_check_raises_python_exceptions():
    with pytest.raises(ValueError):
        _lib.check(2, "x")
    with pytest.raises(_lib.FincError):
        _lib.check(3, "x")


def test_variant_plan_covers_every_row_and_agrees_with_the_library():
    """Host side of tests/test_gpu_variants.py: for every row of the MFMA inverse's instantiation table the planned
    problem counts really select that row (the library's own answer, a host-only call), in the 32-byte-I/O and the
    16-byte form -- so the GPU test provably launches every compiled variant."""
    from helpers import problem_counts_for_row, split_problems
    rows = _lib.inverse_table()
    assert len(rows) >= 29
    hit = set()
    for r, i in enumerate(rows):
        counts = problem_counts_for_row(rows, r)
        assert counts, (r, i)
        for n in counts:
            B, G, _ = split_problems(n)
            one_wave = i["nw"] == 1 and i["npw"] == 1
            for sec, (H, W) in ((2 if one_wave else 1, (10, 32)), (1, (19, 24)), (0, (9, 20))):
                v = _lib.inverse_variant(B, G, i["cqp"], H, W, i["kh"], i["kw"])
                if sec == 2 and v is not None and v["sec"] == 3:      # sector pairing with helper waves: problems in fours, banks that fit
                    assert n % 4 == 0 and i["cqp"] <= 24
                    v = dict(v, sec=2)
                assert v is not None and v["row"] == r and v["sec"] == sec and v["nw"] == i["nw"] and v["npw"] == i["npw"], (r, n, v)
                per_wg = B * G // v["workgroups"]                      # helper-wave form: 4 problems per 8-wave workgroup
                assert per_wg in (v["npw"], 4) and v["workgroups"] * per_wg == B * G and v["lds_bytes"] <= 160 * 1024
                hit.add((r, sec))
    assert len(hit) == 2 * len(rows) + sum(1 for i in rows if i["nw"] == 1 and i["npw"] == 1)
    # the bench shapes: c3 at full batch is the one-wave kernel, at B <= 128 the packed 2-wave split; c5 is the 4-wave split
    assert _lib.inverse_variant(256, 4, 24, 64, 64, 3, 3)["nw"] == 1 and _lib.inverse_variant(256, 4, 24, 64, 64, 3, 3)["sec"] == 3
    assert (_lib.inverse_variant(128, 4, 24, 64, 64, 3, 3)["nw"], _lib.inverse_variant(128, 4, 24, 64, 64, 3, 3)["npw"]) == (2, 2)
    assert _lib.inverse_variant(64, 4, 48, 128, 128, 5, 5)["nw"] == 4
    assert _lib.inverse_variant(2, 4, 24, 16, 15, 3, 3) is None          # W % 4 != 0: strict kernel


def test_role_split_kernel_takes_the_under_filled_chip():
    """finc_split.hip: problem sets that do not outnumber the compute units (c2; c3 at the per-GPU batch of a 4- or 8-way
    split; the c4 units) run on the role-split kernel -- form 4, one workgroup of four waves per problem; the banks of up to 16
    channels on its short-step form (finc_chain.hip, form 6: the recurrence wave, one wave per tap with a + b == 2, the I/O
    wave); everything else stays with the wavefront kernel's table (host-only calls)."""
    from helpers import split_takes, chain_takes
    for (B, G, Cq, H, W, K), want in (((64, 4, 12, 32, 32, 3), True), ((32, 4, 24, 64, 64, 3), True), ((64, 4, 24, 64, 64, 3), True),
                                      ((65, 4, 24, 64, 64, 3), False), ((128, 4, 3, 16, 16, 3), True), ((128, 4, 6, 8, 8, 3), True), ((129, 4, 3, 16, 16, 3), False),
                                      ((64, 4, 3, 16, 16, 3), True),
                                      ((16, 4, 12, 4, 4, 3), True), ((1, 1, 23, 40, 36, 3), True), ((8, 4, 24, 8, 80, 3), False),
                                      ((8, 4, 16, 30, 44, 2), True), ((8, 4, 16, 32, 32, 5), False), ((8, 4, 40, 32, 32, 3), False),
                                      ((8, 4, 12, 8, 80, 3), True), ((2, 4, 16, 16, 256, 3), True), ((8, 4, 20, 8, 80, 3), False)):
        v = _lib.inverse_variant(B, G, Cq, H, W, K, K)
        assert v is not None, (B, G, Cq, H, W, K)
        assert (v["sec"] in (4, 6)) == want, (B, G, Cq, H, W, K, v)
        assert split_takes(v["cqp"], K, K, B * G, H, W) == want
        chain = chain_takes(v["cqp"], K, K, B * G, H, W)
        assert (v["sec"] == 6) == chain, (B, G, Cq, H, W, K, v)
        if chain:
            assert v["cqp"] <= 16 and v["nw"] == (5 if K == 3 else 3) and v["workgroups"] == B * G and v["row"] == -1, v
            assert v["lds_bytes"] <= 64 * 1024 + 2 * (W - min(16, W) + 2) * 128, v
        elif want:
            # (round 4: with compute units to spare -- 2 B G <= 256 -- on a map of >= 2 bands and >= 64 columns the bands of a
            # problem are dealt out to two workgroups)
            per = (H + 15) // 16 if (2 * B * G <= 256 and H > 16 and W >= 64 and K > 1) else 1   # (round 5: one workgroup per band)
            assert v["nw"] == 4 and v["workgroups"] == per * B * G and v["row"] == -1 and v["lds_bytes"] <= 64 * 1024, v


def test_wide_maps_take_the_packed_two_wave_form():
    """finc_mfma.hip find_inst: once four one-wave problems do not fit a CU's LDS (the band hand-over FIFO grows with W) the
    packed two-wave form is chosen at any problem count; at 64x64 and below nothing changes (host-side call, no GPU)."""
    from fincflow_amd import _lib
    v = _lib.inverse_variant(256, 4, 24, 64, 64, 3, 3)
    assert (v["nw"], v["npw"], v["sec"]) == (1, 1, 3)                       # helper waves: four problems per workgroup
    for H, W in ((64, 80), (64, 96), (128, 128), (64, 72)):
        v = _lib.inverse_variant(256, 4, 24, H, W, 3, 3)
        assert (v["nw"], v["npw"]) == (2, 2) and v["lds_bytes"] <= 160 * 1024, (H, W, v)
    assert _lib.inverse_variant(256, 4, 24, 64, 56, 3, 3)["nw"] == 1        # four problems still fit: one wave each
    assert _lib.inverse_variant(131, 1, 24, 8, 80, 3, 3)["nw"] == 1         # an odd count cannot be packed in pairs: one wave each
    assert _lib.inverse_variant(256, 4, 28, 64, 80, 3, 3)["nw"] == 1        # no two-wave row for this bank; three one-wave problems to a unit: its own kernel
    v = _lib.inverse_variant(256, 4, 28, 64, 128, 3, 3)                     # ... two to a unit: it borrows the 32-channel bank's packed two-wave kernel
    assert (v["nw"], v["npw"], v["cqp"]) == (2, 2, 32)


def test_kernel_attribute_table_is_keyed_by_device_and_kernel():
    """finc_mfma_launch sets the 160 KiB dynamic-LDS attribute once per (device, kernel) -- a per-thread or per-kernel-only
    cache would skip the second device of a process that drives two.  Key logic, host only."""
    L = _lib.lib()
    tok = 0x7E57_0000
    assert L.finc_debug_attr_table_insert(5, tok) == 1          # new pair
    assert L.finc_debug_attr_table_insert(5, tok) == 0          # seen
    assert L.finc_debug_attr_table_insert(6, tok) == 1          # same kernel, other device: must be set again
    assert L.finc_debug_attr_table_insert(5, tok + 8) == 1      # other kernel, same device
    import threading
    got = []
    th = [threading.Thread(target=lambda: got.append(L.finc_debug_attr_table_insert(7, tok))) for _ in range(8)]
    [x.start() for x in th]
    [x.join() for x in th]
    assert sorted(got) == [0] * 7 + [1]                         # exactly one thread sets it


def test_no_wide_store_is_overwritten_behind_its_back():
    """gfx950 hazard found in round 3: a buffer_store_dwordx3/x4 whose channel offset rides in an SGPR reads its data registers
    a few cycles after issue, and the compiler inserts no wait state for that form -- an instruction that overwrites those
    registers right behind the store makes it write stale lanes (the F(4,3) forward at Cq = 12 lost the second dword of lanes
    12..15).  scripts/check_store_hazard.py disassembles every kernel of the in-tree library and must find no such pair; it
    also checks its own detector on a synthetic listing."""
    import importlib.util
    import shutil
    spec = importlib.util.spec_from_file_location("check_store_hazard", os.path.join(REPO, "scripts", "check_store_hazard.py"))
    chk = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(chk)
    bad = chk.scan("0000 <k>:\n\tbuffer_store_dwordx4 v[112:115], v136, s[24:27], s34 offen // 0\n\tv_pk_add_f32 v[112:113], v[44:45], v[48:49]\n")
    assert len(bad) == 1 and bad[0][0] == "k"
    ok = chk.scan("0000 <k>:\n\tbuffer_store_dwordx4 v[112:115], v136, s[24:27], 0 offen\n\ts_nop 1\n\tv_pk_add_f32 v[112:113], v[44:45], v[48:49]\n")
    assert ok == []
    assert chk.scan("0000 <k>:\n\tbuffer_store_dwordx2 v[2:3], v1, s[0:3], s9 offen\n\tv_mov_b32_e32 v2, 0\n") == []   # (64-bit stores: no hazard)
    if not shutil.which(os.path.join(chk.LLVM, "llvm-objdump")):
        pytest.skip("no llvm-objdump in this image")
    assert chk.main() == 0


def test_runtime_switches_are_reported():
    """the environment's A/B switches and a pinned forward form change kernel choice and timing; the library
    lists the ones in effect so that a benchmark line can show it ran the library's own dispatch (bench.py refuses otherwise)."""
    import subprocess
    import sys
    code = ("import sys; sys.path.insert(0, %r)\n"
            "from fincflow_amd import _lib\n"
            "a = _lib.runtime_switches()\n"
            "_lib.inverse_variant(256, 4, 24, 64, 64, 3, 3)\n"          # (consults FINC_NO_S64 / FINC_NO_HLP, host only)
            "b = _lib.runtime_switches()\n"
            "_lib.set_forward_form(4)\n"
            "c = _lib.runtime_switches()\n"
            "print(a, '|', b, '|', c)\n") % REPO
    clean = {k: v for k, v in os.environ.items() if not k.startswith("FINC")}
    out = subprocess.run([sys.executable, "-c", code], env=clean, capture_output=True, text=True, check=True).stdout.strip()
    assert out == "[] | [] | ['forward_form_override']", out
    out = subprocess.run([sys.executable, "-c", code], env=dict(clean, FINC_NO_HLP="1"), capture_output=True, text=True, check=True).stdout.strip()
    assert out == "[] | ['FINC_NO_HLP'] | ['FINC_NO_HLP', 'forward_form_override']", out


def test_library_override_is_announced():
    import subprocess
    import sys
    code = "import sys; sys.path.insert(0, %r)\nfrom fincflow_amd import _lib\nprint(_lib.library_info())\n" % REPO
    clean = {k: v for k, v in os.environ.items() if not k.startswith("FINC")}
    r = subprocess.run([sys.executable, "-c", code], env=clean, capture_output=True, text=True, check=True)
    assert "'env_override': False" in r.stdout and "'build_flags': 0" in r.stdout and "FINCFLOW_LIB" not in r.stderr
    r = subprocess.run([sys.executable, "-c", code], env=dict(clean, FINCFLOW_LIB=_lib.LIB_PATH), capture_output=True, text=True, check=True)
    assert "'env_override': True" in r.stdout and "FINCFLOW_LIB override" in r.stderr


def test_grad_weight_form_is_host_side_and_follows_the_documented_rules():
    """finc_debug_backward_variant (no launch): which grad-weight kernel a shape gets -- DESIGN 3.10's rules, spelled out: the pair
    kernel for 3x3 banks of 13..32 channels (W % 4 == 0), the tile-pair kernel above that from 32 columns up and for 5x5 banks
    above 12 channels from 16 columns up, the direct kernels everywhere else."""
    from fincflow_amd import _lib
    want = {
        (256, 4, 24, 64, 64, 3, 3): "winograd",        # c3
        (64, 4, 12, 32, 32, 3, 3): "staged",           # c2: too few strips for the tile-pair form
        (256, 4, 12, 32, 32, 3, 3): "winograd_tiled", (256, 4, 8, 32, 32, 3, 3): "staged",
        (64, 4, 48, 128, 128, 5, 5): "winograd_tiled", # c5
        (8, 4, 13, 8, 4, 3, 3): "winograd", (8, 4, 32, 8, 16, 3, 3): "winograd", (8, 4, 24, 8, 18, 3, 3): "dword",
        (8, 4, 33, 8, 32, 3, 3): "winograd_tiled", (8, 4, 33, 8, 28, 3, 3): "tiled", (8, 1, 96, 8, 64, 3, 3): "winograd_tiled",
        (8, 4, 16, 8, 16, 5, 5): "winograd_tiled", (8, 4, 12, 8, 16, 5, 5): "staged", (8, 4, 48, 8, 12, 5, 5): "tiled",
        (8, 4, 24, 8, 32, 2, 2): "staged", (8, 4, 4, 8, 32, 3, 5): "staged",
        # banks beyond the tables (forward / grad-input on the streaming-bank kernel): the tile-pair kernels take any tile count
        (8, 4, 128, 8, 32, 3, 3): "winograd_tiled", (8, 1, 192, 8, 28, 3, 3): "tiled", (8, 4, 64, 8, 16, 5, 5): "winograd_tiled",
        (8, 4, 40, 8, 16, 2, 2): "tiled", (8, 4, 8, 8, 16, 4, 4): "tiled", (8, 4, 8, 8, 16, 7, 7): "direct", (8, 4, 8, 8, 15, 4, 4): "direct",
    }
    got = {k: _lib.backward_variant(*k)["gradw"] for k in want}
    assert got == want, {k: (got[k], want[k]) for k in want if got[k] != want[k]}


def test_row_chunks_minimise_rounds_times_rows():
    """finc_common.h finc_row_chunks (the launch rule of the one-wave-per-SIMD forward kernels): the measured cases of
    profiles/r05/notes/row_chunks.txt -- c3's F(4,3) forward at B = 96 / 160 / 384 takes 5 / 3 / 2 chunks (the old fill-the-chip rule: 3 / 2 / 1,
    up to 27 % slower) -- and the shapes every bench line launches keep theirs."""
    f = _lib.lib().finc_debug_row_chunks
    rc = lambda units, slots, H, min_rows, extra: f(units, slots, H, min_rows, extra, 16)
    assert rc(384, 1024, 64, 8, 2) == 5 and rc(640, 1024, 64, 8, 2) == 3 and rc(1536, 1024, 64, 8, 2) == 2
    assert rc(1024, 1024, 64, 8, 2) == 1 and rc(512, 1024, 64, 8, 2) == 2 and rc(256, 1024, 64, 8, 2) == 4 and rc(128, 1024, 64, 8, 2) == 8
    assert rc(2048, 1024, 64, 8, 2) == 1 and rc(4096, 1024, 64, 8, 2) == 1
    assert rc(1, 1024, 64, 8, 2) == 8 and rc(1, 1024, 7, 8, 2) == 1 and rc(1, 1024, 16, 8, 2) == 2      # never chunks below min_rows
    assert rc(192, 256, 128, 8, 4) == 4 and rc(48, 256, 128, 8, 4) == 5 and rc(1024, 256, 128, 8, 4) == 1  # F(2,5): c5 at B = 12 (192 strips: three rounds of 32+4 rows), B = 3, and its full batch
    for units in (1, 7, 100, 383, 1000, 1025, 5000):
        for H in (1, 8, 9, 33, 64, 100, 128):
            c = rc(units, 1024, H, 8, 2)
            assert 1 <= c <= max(1, H // 8) and -(-H // -(-H // c)) == c                                  # a count its own row split reproduces
    assert rc(0, 1024, 64, 8, 2) == 0 and rc(1, 0, 64, 8, 2) == 0 and f(1, 1024, 64, 8, 2, 0) == 0
    # the two-waves-per-SIMD kernels (strip, F(2,3)): rounds in SIMDs, a row at 14/16 once waves outnumber them -- F(2,3) at c3's shape,
    # B = 24 (192 strips of 32 columns): 5 chunks (34.5 us; 6, the old rule's count: 43.2); the 2x2 strip kernel at B = 24 / 40 / 96
    # (384 / 640 / 1,536 waves): 5 / 3 / 2 (23.1 / 37.0 / 93.7 us, the fastest or within 5 % of it); full chips keep the old rule's counts
    t2 = lambda units, H, extra: f(units, 1024, H, 4, extra, 14)
    assert t2(192, 64, 2) == 5 and t2(384, 64, 2) == 5 and t2(640, 64, 2) == 3 and t2(1536, 64, 2) == 2
    assert t2(1024, 64, 3) == 2 and t2(2048, 64, 3) == 1 and t2(8192, 64, 3) == 1 and t2(256, 64, 3) == 4
    assert t2(512, 64, 2) == 4                                       # (2x2 strip kernel at B = 32: 4 chunks 26.9 us, the old rule's 2 chunks 29.1)


def test_premultiplied_form_only_where_the_call_is_one_launch():
    """finc_mfma.hip remainder_images: a problem set of whole rounds plus a remainder of at most 512 problems is two launches, and the
    remainder's kernels (role-split, short-step, two-wave variants) have no premultiplied-input form -- the query says so, and a flow stack
    then keeps the plain chain (c3, B = 264: 522 us in two launches against 0.93 x 761 in one)."""
    q = _lib.lib().finc_inverse_premultiplied_supported
    rem = _lib.inverse_remainder_images
    # (the launch's own decision: images of the second launch)
    assert [rem(B, 4, 24, 64, 64, 3, 3) for B in (64, 160, 256, 260, 264, 320, 384, 400, 512, 576)] == [0, 0, 0, 4, 8, 64, 128, 0, 0, 64]
    assert rem(160, 4, 32, 64, 64, 3, 3) == 32 and rem(128, 4, 32, 64, 64, 3, 3) == 0     # the packed two-wave kernels: rounds of 512
    assert rem(256, 4, 28, 64, 64, 3, 3) == 64 and rem(192, 4, 28, 64, 64, 3, 3) == 0     # 28 channels at 64x64: rounds of 768 (three one-wave problems to a unit)
    assert rem(1100, 1, 24, 4, 16, 3, 3) == 76 and rem(341, 3, 24, 4, 16, 3, 3) == 0      # G = 1; a remainder (1,023 problems: none) / off an image boundary
    assert rem(342, 3, 24, 4, 16, 3, 3) == 0                                               # 1,026 problems, G = 3: 2 problems are no whole image
    assert rem(256, 4, 192, 64, 64, 3, 3) == 0 and rem(320, 4, 48, 64, 64, 5, 5) == 0     # streaming-bank / four-wave kernels: one launch
    assert q(256, 4, 24, 64, 64, 3, 3) == 1 and q(512, 4, 24, 64, 64, 3, 3) == 1          # whole rounds
    assert q(264, 4, 24, 64, 64, 3, 3) == 0 and q(320, 4, 24, 64, 64, 3, 3) == 0 and q(384, 4, 24, 64, 64, 3, 3) == 0
    assert q(400, 4, 24, 64, 64, 3, 3) == 1                                               # remainder 576 > 512: one launch of two rounds
    assert q(132, 4, 24, 64, 64, 3, 3) == 1 and q(64, 4, 24, 64, 64, 3, 3) == 0           # (a single partial round; the role-split kernel's 256 problems)
