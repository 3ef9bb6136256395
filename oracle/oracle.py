"""ctypes front end of the CPU oracle (oracle/finc_oracle.c).

TEST INFRASTRUCTURE ONLY: imported by tests/, __graft_entry__.smoke() and
bench.py's cpu_baseline leg -- never by fincflow_amd/.  numpy in, numpy out.
"""
import ctypes
import os
import subprocess

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(HERE, "libfinc_oracle.so")

# orient: two bits per group, bit0 = W-flipped storage (TR), bit1 = H-flipped (BL)
ORDER_BITS = {"TL": 0, "TR": 1, "BL": 2, "BR": 3}
ORIENT_FASTFLOW = 0 | (1 << 2) | (2 << 4) | (3 << 6)  # TL,TR,BL,BR = fastflow.py:24-27


def build(force=False):
    src = os.path.join(HERE, "finc_oracle.c")
    if force or not os.path.exists(LIB_PATH) or os.path.getmtime(LIB_PATH) < os.path.getmtime(src):
        subprocess.check_call(["make", "-C", HERE, "libfinc_oracle.so"], stdout=subprocess.DEVNULL)
    return LIB_PATH


_lib = None


def lib():
    global _lib
    if _lib is None:
        build()
        L = ctypes.CDLL(LIB_PATH)
        fp = ctypes.POINTER(ctypes.c_float)
        dp = ctypes.POINTER(ctypes.c_double)
        i, u = ctypes.c_int, ctypes.c_uint
        L.finc_oracle_canonicalize_f32.argtypes = [fp, fp, i, i, i, i, u]
        L.finc_oracle_check_invariant_f32.argtypes = [fp, i, i, i, i]
        L.finc_oracle_check_invariant_f32.restype = i
        L.finc_oracle_inverse_f32.argtypes = [fp, fp, fp, i, i, i, i, i, i, i, u, i]
        L.finc_oracle_inverse_f32_via_f64.argtypes = [fp, fp, fp, i, i, i, i, i, i, i, u, i]
        L.finc_oracle_inverse_f64_inplace.argtypes = [dp, dp, i, i, i, i, i, i, i]
        L.finc_oracle_forward_f32.argtypes = [fp, fp, fp, i, i, i, i, i, i, i, u, i, i]
        L.finc_oracle_max_threads.restype = i
        _lib = L
    return _lib


def _f(a):
    a = np.ascontiguousarray(a, dtype=np.float32)
    return a, a.ctypes.data_as(ctypes.POINTER(ctypes.c_float))


def _dims(x, wc, G):
    B, C, H, W = x.shape
    assert C % G == 0
    Cq = C // G
    assert wc.shape[0] == C and wc.shape[1] == Cq, (wc.shape, C, Cq)
    return B, Cq, H, W, wc.shape[2], wc.shape[3]


def canonicalize(w_stored, G, orient):
    """w_stored [G*Cq, Cq, KH, KW] (per-order flipped storage) -> TL-canonical."""
    ws, pws = _f(w_stored)
    wc = np.empty_like(ws)
    Cq = ws.shape[0] // G
    lib().finc_oracle_canonicalize_f32(pws, wc.ctypes.data_as(ctypes.POINTER(ctypes.c_float)), G, Cq,
                                       ws.shape[2], ws.shape[3], orient)
    return wc


def check_invariant(wc, G):
    wc, pw = _f(wc)
    return lib().finc_oracle_check_invariant_f32(pw, G, wc.shape[0] // G, wc.shape[2], wc.shape[3])


def inverse_f32(z, wc, G=4, orient=ORIENT_FASTFLOW, nthreads=1):
    """fp32, reference visitation + term order (cinc_cuda_kernel_level2.cu:59-72)."""
    z, pz = _f(z)
    wc, pw = _f(wc)
    B, Cq, H, W, KH, KW = _dims(z, wc, G)
    x = np.zeros_like(z)  # the reference's mandatory zeros_like (fastflow.py:91)
    lib().finc_oracle_inverse_f32(pz, pw, x.ctypes.data_as(ctypes.POINTER(ctypes.c_float)),
                                  B, G, Cq, H, W, KH, KW, orient, nthreads)
    return x


def inverse_via_f64(z, wc, G=4, orient=ORIENT_FASTFLOW, nthreads=1):
    """The Cython CPU path: fp32 -> fp64 solve -> fp32 (layers/conv.py:113-163)."""
    z, pz = _f(z)
    wc, pw = _f(wc)
    B, Cq, H, W, KH, KW = _dims(z, wc, G)
    x = np.empty_like(z)
    lib().finc_oracle_inverse_f32_via_f64(pz, pw, x.ctypes.data_as(ctypes.POINTER(ctypes.c_float)),
                                          B, G, Cq, H, W, KH, KW, orient, nthreads)
    return x


def inverse_f64(z, wc, G=1):
    """fp64 in/out twin of solve_parallel (solve_parallel_mc.pyx:77-126), canonical orientation."""
    y = np.array(z, dtype=np.float64, order="C", copy=True)
    wd = np.ascontiguousarray(wc, dtype=np.float64)
    B, C, H, W = y.shape
    Cq = C // G
    dp = ctypes.POINTER(ctypes.c_double)
    lib().finc_oracle_inverse_f64_inplace(y.ctypes.data_as(dp), wd.ctypes.data_as(dp), B, G, Cq, H, W,
                                          wd.shape[2], wd.shape[3])
    return y


def forward_f32(x, wc, G=4, orient=ORIENT_FASTFLOW, accumulate_f64=True, nthreads=1):
    x, px = _f(x)
    wc, pw = _f(wc)
    B, Cq, H, W, KH, KW = _dims(x, wc, G)
    z = np.empty_like(x)
    lib().finc_oracle_forward_f32(px, pw, z.ctypes.data_as(ctypes.POINTER(ctypes.c_float)),
                                  B, G, Cq, H, W, KH, KW, orient, int(accumulate_f64), nthreads)
    return z


def max_threads():
    return lib().finc_oracle_max_threads()


# ---------------------------------------------------------------------------
# Synthetic weights = PaddedConv2d.reset_parameters restated (layers/conv.py:63-79)
# ---------------------------------------------------------------------------
def make_stored_weights(G, Cq, KH, KW, orient=ORIENT_FASTFLOW, seed=1234, std=0.05):
    """Per group: N(0, std^2), then w[c,c,-1,-1]=1, w[c,c+1:,-1,-1]=0, then the
    per-order flip.  Returns the STORED form [G*Cq, Cq, KH, KW]."""
    out = np.empty((G * Cq, Cq, KH, KW), dtype=np.float32)
    for g in range(G):
        rng = np.random.default_rng(seed + g)
        w = (rng.standard_normal((Cq, Cq, KH, KW)) * std).astype(np.float32)
        for c in range(Cq):
            w[c, c, -1, -1] = 1.0
            w[c, c + 1:, -1, -1] = 0.0
        o = (orient >> (2 * g)) & 3
        if o & 1:
            w = w[:, :, :, ::-1]
        if o & 2:
            w = w[:, :, ::-1, :]
        out[g * Cq:(g + 1) * Cq] = w
    return out
