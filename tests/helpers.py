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
