"""CPU oracle (oracle/finc_oracle.c) pinned against the reference's own outputs.

The fixtures in tests/golden were produced by tests/golden/make_golden.py, which
imports the reference's CPU path.  No GPU needed.
"""
import os

import numpy as np
import pytest

from oracle import oracle
from helpers import ORDER_BITS, ORIENT_FASTFLOW, golden, golden_names, rel_err, unit_stored_weights

UNIT_CASES = golden_names("unit_")
PADDED_CASES = golden_names("padded_")
LITERAL_CASES = golden_names("literal_")


def test_fixture_inventory():
    assert len(UNIT_CASES) == 11 and len(PADDED_CASES) == 6 and len(LITERAL_CASES) == 6 and len(golden_names("grad_")) == 8


@pytest.mark.parametrize("name", UNIT_CASES)
def test_unit_inverse_matches_reference_cython_path(name):
    """finc_oracle_inverse_f32_via_f64 == FastFlowUnit.reverse_level1 (fastflow.py:57-76), bit for bit."""
    g = golden(name)
    wc = oracle.canonicalize(unit_stored_weights(g), 4, ORIENT_FASTFLOW)
    assert oracle.check_invariant(wc, 4) == 0
    got = oracle.inverse_via_f64(g["z"], wc, 4, ORIENT_FASTFLOW)
    assert np.array_equal(got, g["x_rev_cython"])


@pytest.mark.parametrize("name", [n for n in UNIT_CASES if "x_rev_python_fp32" in golden(n)])
def test_unit_inverse_fp32_matches_reference_python_path(name):
    """finc_oracle_inverse_f32 == PaddedConv2d.reverse_python per group (layers/conv.py:165-189), bit for bit."""
    g = golden(name)
    wc = oracle.canonicalize(unit_stored_weights(g), 4, ORIENT_FASTFLOW)
    got = oracle.inverse_f32(g["z"], wc, 4, ORIENT_FASTFLOW)
    assert np.array_equal(got, g["x_rev_python_fp32"])


@pytest.mark.parametrize("name", UNIT_CASES)
def test_unit_forward_matches_reference(name):
    """finc_oracle_forward_f32 vs FastFlowUnit.forward (fastflow.py:31-50); ATen's summation order is
    not ours, so 1e-5 relative (the north_star tolerance); logdet is exactly 0."""
    g = golden(name)
    wc = oracle.canonicalize(unit_stored_weights(g), 4, ORIENT_FASTFLOW)
    for acc64 in (True, False):
        got = oracle.forward_f32(g["x"], wc, 4, ORIENT_FASTFLOW, accumulate_f64=acc64)
        assert rel_err(got, g["z"]) <= 1e-5
    assert float(g["logdet"]) == 0.0


@pytest.mark.parametrize("name", UNIT_CASES)
def test_unit_round_trip(name):
    g = golden(name)
    wc = oracle.canonicalize(unit_stored_weights(g), 4, ORIENT_FASTFLOW)
    xr = oracle.inverse_f32(g["z"], wc, 4, ORIENT_FASTFLOW)
    # The "heavy" fixture (free taps x2, Cq=16) is ill-conditioned on purpose: there the reference's OWN
    # fp32 (python) and fp64 (cython) paths differ by 1.7e-4, so 1e-5 is only meaningful at init-scale weights.
    tol = 1e-3 if "heavy" in name else 1e-5
    assert rel_err(xr, g["x"]) <= tol
    assert rel_err(xr, g["x_rev_cython"]) <= tol


@pytest.mark.parametrize("name", PADDED_CASES)
def test_padded_all_orders(name):
    g = golden(name)
    o = ORDER_BITS[str(g["order"])]
    wc = oracle.canonicalize(g["w"], 1, o)
    assert np.array_equal(oracle.inverse_via_f64(g["z"], wc, 1, o), g["x_rev_cython"])
    if "x_rev_python_fp32" in g:
        got = oracle.inverse_f32(g["z"], wc, 1, o)
        assert np.array_equal(got, g["x_rev_python_fp32"])
        if "x_rev_python_fp32_diag" in g:  # utils/solve_mc.py:8-50 == :88-114
            assert np.array_equal(got, g["x_rev_python_fp32_diag"])
    if "zdirect" not in name:
        assert rel_err(oracle.forward_f32(g["x"], wc, 1, o), g["z"]) <= 1e-5
    assert float(g["logdet"]) == 0.0 and float(g["logdet_rev"]) == 0.0


def test_mask_and_pad_contract():
    """layers/conv.py:41-55 (pad tuples) and :81-96 (mask) as recorded in the fixtures."""
    pads = {"TL": (2, 0, 2, 0), "TR": (0, 2, 2, 0), "BL": (2, 0, 0, 2), "BR": (0, 2, 0, 2)}
    for order, pad in pads.items():
        g = golden(f"padded_{order}_B2_C3_7x7_k3")
        assert tuple(g["pad"]) == pad
        m = oracle.canonicalize(g["mask"], 1, ORDER_BITS[order])
        for c in range(3):
            assert np.all(m[c, c:, -1, -1] == 0) and np.all(m[c, :c, -1, -1] == 1)
        assert m.sum() == m.size - 6


@pytest.mark.parametrize("name", LITERAL_CASES)
def test_literal_known_answers(name):
    """cuda/cinc_cuda/test_cuda_kernel.py:3-56 and fastflow/test_examples.py:6-25,54-73."""
    g = golden(name)
    o = ORDER_BITS[str(g["order"])]
    wc = oracle.canonicalize(g["w"], 1, o)
    if bool(g["reverse_first"]):
        out = oracle.inverse_f32(g["inp"], wc, 1, o)
        assert np.array_equal(out, g["out"])
        # util.py:36 -- re-convolving the inverse gives the input back (exact: small integers)
        assert np.array_equal(oracle.forward_f32(out, wc, 1, o), g["inp"])
    else:
        out = oracle.forward_f32(g["inp"], wc, 1, o)
        assert np.array_equal(out, g["out"])
        assert np.array_equal(oracle.inverse_f32(out, wc, 1, o), g["inp"])


def test_literal_closed_forms():
    """identity kernel => output == input; I2 => x[h,w] = z[h,w] - x[h-1,w-1]."""
    g = golden("literal_cinc_id2")
    assert np.array_equal(g["out"], g["inp"])
    g = golden("literal_cinc_eye2")
    z, x = g["inp"][0, 0], g["out"][0, 0]
    exp = z.copy()
    for h in range(1, 4):
        for w in range(1, 4):
            exp[h, w] = z[h, w] - exp[h - 1, w - 1]
    assert np.array_equal(x, exp)


def test_invariant_violation_detected():
    w = oracle.make_stored_weights(1, 4, 3, 3, orient=0)
    assert oracle.check_invariant(w, 1) == 0
    w[1, 1, -1, -1] = 0.5
    assert oracle.check_invariant(w, 1) != 0
    w = oracle.make_stored_weights(1, 4, 3, 3, orient=0)
    w[0, 2, -1, -1] = 0.1
    assert oracle.check_invariant(w, 1) != 0


def test_synthetic_weights_follow_reference_init():
    """oracle.make_stored_weights restates PaddedConv2d.reset_parameters (layers/conv.py:63-79)."""
    ws = oracle.make_stored_weights(4, 6, 3, 3)
    wc = oracle.canonicalize(ws, 4, ORIENT_FASTFLOW)
    assert oracle.check_invariant(wc, 4) == 0
    free = wc[:, :, :2, :]
    assert abs(float(free.std()) - 0.05) < 0.01


def test_against_live_reference_solver_if_built():
    """When oracle/_ref holds the reference's rebuilt Cython module, compare directly on fresh data."""
    from oracle import build_ref
    solve_parallel = build_ref.load()
    if solve_parallel is None:
        pytest.skip("oracle/_ref not built (reference tree absent)")
    rng = np.random.default_rng(5)
    wc = oracle.make_stored_weights(1, 7, 3, 3, orient=0, seed=77)
    z = rng.standard_normal((2, 7, 9, 13)).astype(np.float32)
    ref = solve_parallel(np.array(z, dtype=np.float64), np.array(wc, dtype=np.float64), (3, 3))
    assert np.array_equal(oracle.inverse_f64(z, wc, 1), ref)
    assert np.array_equal(oracle.inverse_via_f64(z, wc, 1, 0), ref.astype(np.float32))


def test_openmp_threads_do_not_change_results():
    rng = np.random.default_rng(3)
    ws = oracle.make_stored_weights(4, 5, 3, 3)
    wc = oracle.canonicalize(ws, 4, ORIENT_FASTFLOW)
    z = rng.standard_normal((3, 20, 10, 12)).astype(np.float32)
    a = oracle.inverse_f32(z, wc, nthreads=1)
    b = oracle.inverse_f32(z, wc, nthreads=4)
    assert np.array_equal(a, b)


def test_config5_init_std_is_unstable_in_the_reference_arithmetic():
    """configs[4] (5x5, Cq=48) at the reference's init std 0.05 (layers/conv.py:64): the inverse ITSELF diverges -- in the
    reference's own fp64 arithmetic |inverse(N(0,1))| grows by orders of magnitude across a 48x48 map -- so no fp32
    implementation (the reference's included) can hold 1e-5 there.  The c5 parity and bench runs therefore use std 0.02
    (same operator norm as 3x3 / Cq=24; precedent for scaling with k^2: cuda/cinc_cuda/test_large_cuda_kernel.py:12-13).
    This test pins the measurement behind that deviation."""
    rng = np.random.default_rng(0)
    z = rng.standard_normal((1, 48, 48, 48)).astype(np.float32)
    grow = {}
    for std in (0.05, 0.02):
        wc = oracle.make_stored_weights(1, 48, 5, 5, orient=0, seed=1234, std=std)
        x = oracle.inverse_f64(z, wc, 1)
        grow[std] = float(np.abs(x).max())
    assert grow[0.05] > 1e5, grow          # diverges (3e7 measured)
    assert grow[0.02] < 50, grow           # the std the c5 runs use stays O(|z|)
    # and a round trip through fp32 storage cannot survive the unstable case even with an fp64 solve: the 1e-7 rounding
    # of z = forward(x) is amplified like everything else
    x = rng.standard_normal((1, 48, 48, 48)).astype(np.float32)
    err = {}
    for std in (0.05, 0.02):
        wc = oracle.make_stored_weights(1, 48, 5, 5, orient=0, seed=1234, std=std)
        zz = oracle.forward_f32(x, wc, 1, 0)
        xr = oracle.inverse_via_f64(zz, wc, 1, 0)
        err[std] = float(np.abs(xr - x).max() / np.abs(x).max())
    assert err[0.05] > 1e-3 and err[0.02] < 1e-5, err


def test_trained_like_weights_lose_their_digits_at_64x64_in_the_reference_arithmetic():
    """Why tests/test_gpu_round4.py::test_f43_on_trained_like_weights... has no inverse half (the claim was
    on record, not pinned).  One group of the c3 bank -- 24 channels, 3x3, the reference's init (layers/conv.py:63-79) -- with its
    free taps x 1.5 (tests/golden/make_golden.py's `heavy` rule): the round trip x -> forward (fp32) -> the reference-order fp64
    solve loses its digits as the map grows -- 1e-5 is out of reach from 48x48 on (3.8e-5 measured), at 64x64 the error is 1.6e-3,
    a hundred and fifty times the bar -- while the init weights themselves round-trip to 6e-8 at 64x64.  The 1e-7 rounding of z is amplified by the solve
    itself (the fp32-order solve: no better), so no implementation can be held to 1e-5 there; the inverse meets trained-like
    weights in the `heavy` golden fixtures, at sizes the reference can still solve."""
    rng = np.random.default_rng(3)
    ws = oracle.make_stored_weights(1, 24, 3, 3, orient=0, seed=1234, std=0.05)
    mask = np.ones_like(ws)
    for c in range(24):
        mask[c, c:, 2, 2] = 0.0                     # the fixed corner entries (layers/conv.py:81-96) keep their values
    heavy = (ws * (1.0 + 0.5 * mask)).astype(np.float32)
    err = {}
    for name, w in (("init", ws), ("x1.5", heavy)):
        for n in (16, 32, 48, 64):
            x = rng.standard_normal((1, 24, n, n)).astype(np.float32)
            z = oracle.forward_f32(x, w, 1, 0)
            xr = oracle.inverse_via_f64(z, w, 1, 0)
            err[name, n] = float(np.abs(xr - x).max() / np.abs(x).max())
    assert err["init", 64] < 1e-5, err
    assert err["x1.5", 16] < 1e-5, err              # small maps are fine: the `heavy` fixtures live there
    assert err["x1.5", 48] > 1e-5 and err["x1.5", 64] > 1e-4, err
    assert err["x1.5", 16] < err["x1.5", 32] < err["x1.5", 48] < err["x1.5", 64], err


def test_the_compiled_reference_never_travels_to_the_gpu_box():
    """BASELINE.md 2 / SURVEY 8c: the reference runs only in the build container.  oracle/_ref/ (the reference's .pyx, cythonised
    and compiled by oracle/build_ref.py) must stay excluded from the gpurun snapshot AND from history, and nothing that runs on
    the GPU box may load it: bench.py's cpu_baseline is the pinned port (kind "port")."""
    import os
    import re
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    ignore = [l.strip() for l in open(os.path.join(root, ".gpurunignore")) if l.strip() and not l.startswith("#")]
    assert "oracle/_ref/" in ignore or "oracle/_ref" in ignore, ignore
    assert not any(l.startswith("oracle/_ref/") and l != "oracle/_ref/" for l in ignore), ignore   # (no narrower pattern instead)
    git_ignore = [l.strip() for l in open(os.path.join(root, ".gitignore"))]
    assert "oracle/_ref/" in git_ignore
    bench = open(os.path.join(root, "bench.py")).read()
    assert "build_ref" not in bench and not re.search(r"kind\W+reference\W+sample", bench)
    for name in ("__init__.py", "_lib.py", "ops.py", "layers.py", "glow.py", "dist.py"):
        src = open(os.path.join(root, "fincflow_amd", name)).read()
        assert "oracle" not in src, name
