"""Shared helpers for the parity tests (numpy only)."""
import glob
import os

import numpy as np

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
GOLDEN = os.path.join(REPO, "tests", "golden")

ORDER_BITS = {"TL": 0, "TR": 1, "BL": 2, "BR": 3}
ORIENT_FASTFLOW = 0 | (1 << 2) | (2 << 4) | (3 << 6)


def golden(name):
    return np.load(os.path.join(GOLDEN, name + ".npz"))


def golden_names(prefix):
    return sorted(os.path.basename(p)[:-4] for p in glob.glob(os.path.join(GOLDEN, prefix + "*.npz")))


def unit_stored_weights(g):
    """The four state-dict tensors conv_{tl,tr,bl,br}.conv.weight, concatenated on dim 0."""
    return np.concatenate([g["w_tl"], g["w_tr"], g["w_bl"], g["w_br"]], axis=0)


def rel_err(a, b):
    """max |a-b| / max |b|  -- the 'relative fp32 error' of BASELINE.json's north_star."""
    a = np.asarray(a, dtype=np.float64)
    b = np.asarray(b, dtype=np.float64)
    return float(np.max(np.abs(a - b)) / max(np.max(np.abs(b)), 1e-30))


# ---------------------------------------------------------------------------
# config-4 style stack: one spec, built from the reference's classes by make_golden_stack.py and
# from fincflow_amd.glow by the GPU test; big parameters are filled deterministically instead of stored.
# ---------------------------------------------------------------------------
STACK_SPEC = [
    ("squeeze",),
    ("ffu", 12, 3), ("actnorm", 12), ("conv1x1", 12), ("coupling", (12, 8, 8), 32),
    ("ffu", 12, 3), ("actnorm", 12), ("conv1x1", 12), ("coupling", (12, 8, 8), 32),
    ("squeeze",),
    ("ffu", 48, 3), ("actnorm", 48), ("conv1x1", 48), ("coupling", (48, 4, 4), 32),
]
STACK_INPUT = (2, 3, 16, 16)


def det_fill(name, shape, scale):
    import zlib
    rng = np.random.default_rng(zlib.crc32(name.encode()))
    return (rng.standard_normal(shape) * scale).astype(np.float32)


def fill_stack_parameters(layers, ffu_weights=None):
    """Deterministic parameters for every non-FInC layer of a stack built from STACK_SPEC (works on the reference's
    modules and on ours: same parameter names).  FastFlowUnit weights come from `ffu_weights` (fixture) if given."""
    import torch
    with torch.no_grad():
        for idx, (spec, m) in enumerate(zip(STACK_SPEC, layers)):
            kind = spec[0]
            if kind == "actnorm":
                m.translation.copy_(torch.from_numpy(det_fill(f"L{idx}.t", m.translation.shape, 0.1)))
                m.log_scale.copy_(torch.from_numpy(det_fill(f"L{idx}.s", m.log_scale.shape, 0.1)))
                m.initialized.fill_(1)
            elif kind == "conv1x1":
                q = np.linalg.qr(det_fill(f"L{idx}.W", tuple(m.W.shape), 1.0).astype(np.float64))[0]
                m.W.copy_(torch.from_numpy(q.astype(np.float32)))
            elif kind == "coupling":
                for pname, prm in m.net.named_parameters():
                    scale = 0.01 if (pname.endswith("bias") or pname.endswith("logs")) else 0.05
                    prm.copy_(torch.from_numpy(det_fill(f"L{idx}.{pname}", tuple(prm.shape), scale)))
                # In the reference Conv2dZero.bias and .logs are two Parameters over ONE tensor
                # (layers/coupling.py:33-39), so whatever is written last (logs) is the value of both.
                m.net[4].bias.copy_(m.net[4].logs)
            elif kind == "ffu" and ffu_weights is not None:
                for o in ("tl", "tr", "bl", "br"):
                    getattr(m, f"conv_{o}").conv.weight.copy_(torch.from_numpy(ffu_weights[f"L{idx}.{o}"]))


# ---------------------------------------------------------------------------
# error yardsticks and the parity report
# ---------------------------------------------------------------------------
def elem_rel_err(a, b, floor=1e-3):
    """Element-wise relative error max |a-b| / max(|b|, floor * max|b|): every element is judged against its own
    magnitude, down to `floor` of the largest one (below that an fp32 result has no relative meaning)."""
    a = np.asarray(a, dtype=np.float64)
    b = np.asarray(b, dtype=np.float64)
    scale = np.maximum(np.abs(b), floor * max(np.max(np.abs(b)), 1e-30))
    return float(np.max(np.abs(a - b) / scale))


def report(kind, **fields):
    """Append one line to gpurun_out/parity_report.jsonl (merged back from the GPU box): the achieved errors of every
    parity case, so loosened tolerances are on record.  Best effort -- never fails a test."""
    import json
    try:
        d = os.path.join(REPO, "gpurun_out")
        os.makedirs(d, exist_ok=True)
        with open(os.path.join(d, "parity_report.jsonl"), "a") as f:
            f.write(json.dumps(dict(kind=kind, **fields)) + "\n")
    except OSError:
        pass


# ---------------------------------------------------------------------------
# the MFMA inverse's instantiation table (finc_mfma.hip g_insts), walked by tests
# ---------------------------------------------------------------------------
SPLIT_MAX_PROBLEMS = 256      # finc_split.hip split_max_problems()
SPLIT_BANKS = {(3, 3): (4, 8, 12, 16, 20, 24, 28, 32), (2, 2): (4, 8, 12, 16, 24, 32)}


def chain_takes(cqp, kh, kw, problems, H, W):
    """Mirror of finc_chain.hip finc_chain_takes (inside finc_split_takes): the short-step form of the role-split kernel takes the
    2x2 / 3x3 banks of up to 16 channels on any map whose width is a multiple of 4 (its hand-over FIFO is 128 bytes per step)."""
    if problems > 2 * SPLIT_MAX_PROBLEMS or cqp > 16 or cqp not in SPLIT_BANKS.get((kh, kw), ()) or H < 1 or W < 4 or W % 4:
        return False
    P = min(16, W)
    nbw = sum(1 for a in range(kh) for b in range(kw) if a + b == 2)
    fixed = 8 * (64 + 8 + 1 + 8) * 16 + 2 * nbw * 1024 + (cqp // 4) * 8 * 1024
    if not (P >= kh - 1 and fixed + 2 * (W - P + 2) * 128 <= 160 * 1024):
        return False
    # (finc_split_uses_chain: a 16-channel problem whose bands the role-split kernel would deal out to two workgroups stays there)
    band_split = 2 * problems <= 256 and H > 16 and W >= 64 and kh > 1 and (W - P + kh + kw - 2) * 4 * (kh - 1) * 4 <= 2048 - 4
    return not (cqp == 16 and band_split)


def split_takes(cqp, kh, kw, problems, H, W):
    """Mirror of finc_split.hip finc_split_takes: the role-split kernel runs the problem sets that do not outnumber the
    compute units, for the 2x2 / 3x3 banks one wave holds, on maps whose hand-over FIFO fits its 2 KB per k-step (the banks of
    up to 16 channels, on the short-step form: any width)."""
    if chain_takes(cqp, kh, kw, problems, H, W):
        return True
    if problems > SPLIT_MAX_PROBLEMS or cqp not in SPLIT_BANKS.get((kh, kw), ()) or H < 1 or W < 4 or W % 4:
        return False
    P = min(16, W)
    return P >= kh - 1 and (W - P + kh + kw - 2) * 4 * (kh - 1) * 4 <= 2048 - 4


def pick_row(rows, cqp, kh, kw, problems, H=10, W=32):
    """Mirror of the library's selection rule: -1 when the role-split kernel takes the problem set, else find_inst's table
    walk (table order; max_problems; problems % npw) -- LDS fit not modelled, the test shapes are narrow.  The host test
    checks this mirror against the library's own answer."""
    if split_takes(cqp, kh, kw, problems, H, W):
        return -1
    for r, i in enumerate(rows):
        if (i["cqp"], i["kh"], i["kw"]) != (cqp, kh, kw):
            continue
        if i["max_problems"] > 0 and problems > i["max_problems"]:
            continue
        if problems % i["npw"] != 0:
            continue
        if (cqp, kh, kw) == (28, 3, 3) and problems % 2 == 0 and borrowed_form_wins(problems, W):
            return next(k for k, x in enumerate(rows) if (x["cqp"], x["kh"], x["kw"], x["nw"], x["npw"]) == (32, 3, 3, 2, 2))
        return r
    return None


def borrowed_form_wins(problems, W):
    """The 28-channel 3x3 bank on the 32-channel bank's packed two-wave kernel (finc_mfma.hip, borrowed_form_wins): the chip takes
    one-wave problems n1 to a compute unit (as many of their rings as fit 160 KB, at most four), two-wave problems two; a two-wave
    round takes 13/16 of a one-wave round; beyond 512 problems only where at most two one-wave problems fit a unit."""
    P = min(W, 16)
    lds = 4 * (7 * 12 * 64 + 7 * 8 * 64 + (W - P + 1) * 56 + 56 + 64)
    n1 = max(1, min(4, (160 * 1024 - 64) // lds))
    if n1 >= 3 and problems > 512:          # (the bank's own kernel has the better rate and hands a remainder to the role-split kernel)
        return False
    r1, r2 = -(-problems // (n1 * 256)), -(-problems // 512)
    return r2 * 13 < r1 * 16


def problem_counts_for_row(rows, r):
    """Problem counts (B*G) that select row r: the smallest, and the ones next to each max_problems edge of the shape (for
    the banks the role-split kernel serves: counts beyond its 256 problems)."""
    i = rows[r]
    shape = (i["cqp"], i["kh"], i["kw"])
    edges = sorted({x["max_problems"] for x in rows if (x["cqp"], x["kh"], x["kw"]) == shape and x["max_problems"] > 0})
    cands = [1, 2, 3, 4, 6, 8, 257, 258, 259, 260, 262, 264, 513, 514, 515, 516, 518, 520]   # (beyond the role-split kernel's 256 and its short-step form's 512)
    for e in edges:
        cands += [e - 2, e - 1, e, e + 1, e + 2, e + 4]
    hits = [n for n in sorted(set(cands)) if n > 0 and pick_row(rows, *shape, n) == r]
    if not hits:
        return []
    out = [hits[0]]
    quad = [n for n in hits if n % 4 == 0]          # a FastFlowUnit-grouped count (G = 4) whenever the row admits one
    if quad:
        out.append(quad[0])
    if edges:                                        # and the counts right at the far side of the edges
        out += [n for n in (max((h for h in hits if h % 2 == 1), default=None),
                            max((h for h in hits if h % 2 == 0), default=None)) if n]
    if len(set(out)) < 2 and len(hits) > 1:
        out.append(hits[1])
    return sorted(set(out))


def split_problems(n):
    """problems -> (B, G, orient): FastFlowUnit grouping when the count allows it, else a single-group layer."""
    if n % 4 == 0:
        return n // 4, 4, ORIENT_FASTFLOW
    return n, 1, n % 4


# ---------------------------------------------------------------------------
# random-shape parity sweep (scripts/fuzz_parity.py and tests/test_gpu_variants.py share the generator)
# ---------------------------------------------------------------------------
def fuzz_case(rng, case):
    K = int(rng.choice([2, 3, 3, 3, 5]))
    # (channel counts between the compiled banks -- 22, 36, 44, 50 at 3x3, 20 at 5x5 -- run on the next larger bank)
    # (72, 96: the big banks of finc_big.hip)
    cq_opts = [1, 2, 3, 4, 6, 8, 12, 16, 20, 22, 24, 28, 32, 36, 40, 44, 48, 50, 64, 72, 96] if K == 3 else \
        ([1, 3, 4, 8, 12, 13, 16, 24, 32] if K == 2 else [2, 4, 8, 12, 16, 20, 32, 48])
    Cq = int(rng.choice(cq_opts))
    G = int(rng.choice([1, 4, 4, 4]))
    H = int(rng.integers(1, 41))
    W = int(rng.choice([rng.integers(1, 41), 4 * rng.integers(1, 12), 8 * rng.integers(1, 9), 16 * rng.integers(1, 5)]))   # W % 16 == 0: the staged forward
    B = int(rng.integers(1, 4))
    if case % 5 == 4:          # every fifth case: more problems than compute units on a small map (the full-chip forms of the inverse)
        B = int(rng.integers(65, 90)) * (4 if G == 1 else 1)
        H = int(rng.integers(1, 20))
        W = int(rng.choice([8, 12, 16, 16, 32]))
    orient = ORIENT_FASTFLOW if G == 4 else int(rng.integers(0, 4))
    std = (0.05 if K < 5 else 0.02) * min(1.0, (24.0 / Cq) ** 0.5)   # keep the operator norm of the bank roughly constant
    return dict(case=case, B=B, G=G, Cq=Cq, H=H, W=W, K=K, orient=orient, std=std)
