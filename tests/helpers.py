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
