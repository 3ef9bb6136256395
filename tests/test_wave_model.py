"""The lane-level model of the MFMA wavefront kernel (tests/wave_model.py) reproduces the oracle.
This pins the kernel's schedule / indexing on a machine without a GPU; the GPU tests pin the kernel itself."""
import numpy as np
import pytest

from oracle import oracle
from helpers import rel_err
import wave_model

CASES = [(4, 8, 8, 3, 3), (3, 16, 16, 3, 3), (6, 8, 8, 3, 3), (12, 20, 20, 3, 3), (24, 19, 32, 3, 3),
         (16, 12, 12, 2, 2), (4, 10, 12, 5, 5), (2, 10, 16, 3, 5), (5, 7, 4, 3, 3), (8, 40, 8, 3, 3),
         (1, 8, 8, 3, 3), (3, 4, 4, 3, 3), (8, 20, 48, 3, 3), (4, 33, 16, 2, 2), (8, 9, 64, 3, 3),
         (4, 18, 24, 3, 3), (8, 35, 40, 3, 3), (4, 17, 56, 2, 2), (12, 40, 32, 3, 3)]   # W % 16 == 8: a row's odd last piece


@pytest.mark.parametrize("CQ,H,W,KH,KW", CASES)
def test_wave_schedule(CQ, H, W, KH, KW):
    rng = np.random.default_rng(CQ * 1000 + H)
    wc = oracle.make_stored_weights(1, CQ, KH, KW, orient=0, seed=CQ + H)
    x = rng.standard_normal((1, CQ, H, W)).astype(np.float32)
    z = oracle.forward_f32(x, wc, 1, 0)
    xi = wave_model.run(z[0].astype(np.float64), wc, fwd=False)
    assert not np.isnan(xi).any(), "a pixel was never stored"
    assert rel_err(xi, oracle.inverse_f64(z, wc, 1)[0]) < 1e-12


@pytest.mark.parametrize("CQ,H,W,KH,KW", [(24, 19, 32, 3, 3), (3, 16, 16, 3, 3), (12, 20, 20, 3, 3), (16, 12, 12, 2, 2)])
def test_wave_schedule_with_folded_affine(CQ, H, W, KH, KW):
    """the affine map in front of the inverse rides in the bank (z-term columns scaled, accumulators start
    from Linv*shift, idle lanes held at zero): the model on y equals the oracle on scale*y + shift."""
    rng = np.random.default_rng(CQ * 77 + H)
    wc = oracle.make_stored_weights(1, CQ, KH, KW, orient=0, seed=CQ + H)
    y = rng.standard_normal((1, CQ, H, W)).astype(np.float32)
    scale = np.exp(0.3 * rng.standard_normal(CQ))
    shift = rng.standard_normal(CQ)
    z = (y[0].astype(np.float64) * scale[:, None, None] + shift[:, None, None])
    xi = wave_model.run(y[0].astype(np.float64), wc, fwd=False, scale=scale, shift=shift)
    ref = oracle.inverse_f64(z[None], wc, 1)[0]
    assert np.isfinite(xi).all()
    assert rel_err(xi, ref) <= 1e-9
