"""Round 5: the short-step inverse of the small banks (finc_chain.hip) -- every bank, every hand-over form, the widths the
role-split kernel could not take -- through the C ABI against the oracle; the band split under contention and on two streams.

Recurrence under test: cinc_cuda_kernel_level2.cu:59-72; visitation: cinc_cuda_kernel_level2.cu:49-56,98-111.
"""
import numpy as np
import pytest
import torch

from oracle import oracle
from helpers import ORIENT_FASTFLOW, rel_err
from test_gpu_variants import bank_std, run_inverse_case

pytestmark = pytest.mark.gpu

TOL = 1e-5


@pytest.fixture(scope="module")
def dev():
    assert torch.cuda.is_available(), "GPU tests need a ROCm device"
    from fincflow_amd import _lib
    _lib.lib()
    return torch.device("cuda:0")


def t(a, dev):
    return torch.from_numpy(np.ascontiguousarray(a)).to(dev)


# (B, G, Cq, H, W, KH, KW).  The bench shapes (c2, the c4 units); every bank (4, 8, 12, 16 channels; padded counts 1, 3, 6, 7, 11, 13,
# 15) at 3x3 and 2x2; W == P (the recurrence wave writes the rows above the next band itself: 4, 8, 12, 16 columns) and W > P (the I/O
# wave copies them from the FIFO: 20 .. 256 columns, beyond the 72 the role-split kernel's FIFO block holds); one band, partial last
# bands, many bands; a single row; G = 1, 2, 3 (orientation codes other than FastFlow's); 256 problems (one per compute unit)
CHAIN_CASES = [
    (64, 4, 12, 32, 32, 3, 3), (32, 4, 3, 16, 16, 3, 3), (16, 4, 6, 8, 8, 3, 3), (8, 4, 12, 4, 4, 3, 3),
    (2, 4, 4, 33, 20, 3, 3), (3, 4, 8, 17, 24, 3, 3), (2, 4, 16, 40, 36, 3, 3), (2, 4, 1, 9, 12, 3, 3), (5, 1, 7, 50, 16, 3, 3),
    (2, 2, 11, 5, 44, 3, 3), (1, 3, 13, 21, 28, 3, 3), (2, 4, 15, 1, 32, 3, 3), (3, 4, 12, 100, 8, 3, 3), (2, 4, 10, 7, 4, 3, 3),
    (1, 4, 12, 24, 128, 3, 3), (1, 2, 16, 18, 256, 3, 3), (2, 4, 8, 35, 80, 3, 3), (4, 4, 12, 64, 64, 3, 3), (1, 4, 5, 130, 12, 3, 3),
    (4, 4, 2, 6, 8, 2, 2), (2, 4, 8, 23, 32, 2, 2), (3, 1, 12, 16, 16, 2, 2), (2, 4, 15, 37, 20, 2, 2), (1, 4, 16, 12, 96, 2, 2),
    (64, 4, 16, 16, 16, 3, 3),
]


@pytest.mark.parametrize("case", CHAIN_CASES, ids=lambda c: "B%d_G%d_Cq%d_%dx%d_k%dx%d" % c)
def test_short_step_kernel(case, dev):
    """finc_chain.hip against the oracle's fp64 path (<= 1e-5 of the largest entry), the strict kernel bit-exact beside it, repeated
    launches bit-identical, and the library says which kernel it was (form 6: the recurrence wave, one wave per tap with
    a + b == 2, the I/O wave)."""
    from fincflow_amd import _lib, ops
    B, G, Cq, H, W, KH, KW = case
    orient = ORIENT_FASTFLOW if G == 4 else (0x1B & ((1 << (2 * G)) - 1))
    v = _lib.inverse_variant(B, G, Cq, H, W, KH, KW)
    assert v is not None and v["sec"] == 6 and v["nw"] == (5 if KH == 3 else 3) and v["workgroups"] == B * G and v["row"] == -1, v
    e_max, _ = run_inverse_case(dev, B, G, orient, Cq, H, W, KH, KW, seed=17 * Cq + H + W, tag="short_step")
    assert e_max <= TOL
    rng = np.random.default_rng(5)
    ws = oracle.make_stored_weights(G, Cq, KH, KW, orient=orient, seed=4, std=bank_std(Cq, max(KH, KW)))
    wc = ops.canonicalize(t(ws, dev), G, orient)
    z = t(rng.standard_normal((B, G * Cq, H, W)).astype(np.float32), dev)
    first = ops.finc_inverse(z, wc, G, orient)
    for _ in range(10):
        assert torch.equal(ops.finc_inverse(z, wc, G, orient), first)
    assert _lib.hlp_timeouts() == 0 and not _lib.fault_pending()


def test_short_step_kernel_on_sampling_input_and_in_a_graph(dev):
    """z ~ N(0,1) (the sampling distribution, train/losses.py:42-45) through the module at c2's shape, eager and replayed from a
    captured graph; a folded affine map (scale and shift: the shift enters masked, B wave 0) against the two-launch form."""
    from fincflow_amd import FastFlowUnit, _lib, glow
    B, C, H, W = 64, 48, 32, 32
    assert _lib.inverse_variant(B, 4, C // 4, H, W, 3, 3)["sec"] == 6
    torch.manual_seed(11)
    unit = FastFlowUnit(C, C, 3).to(dev)
    an = glow.ActNorm(C).to(dev)
    with torch.no_grad():
        an.log_scale.copy_(0.2 * torch.randn(C, device=dev))
        an.translation.copy_(torch.randn(C, device=dev))
        an.initialized.fill_(1)
    y = torch.randn(B, C, H, W, device=dev)
    with torch.no_grad():
        ref = unit.reverse(y)
        ws = torch.cat(unit._weights()).detach().cpu().numpy()
        wco = oracle.canonicalize(ws, 4, ORIENT_FASTFLOW)
        want = oracle.inverse_via_f64(y[:4].cpu().numpy(), wco, 4, ORIENT_FASTFLOW, nthreads=8)
        assert rel_err(ref[:4].cpu().numpy(), want) <= TOL
        two = unit.reverse(an.reverse(y))
        fused = unit.reverse_affine(y, an.log_scale, an.translation)
        assert fused is not None and rel_err(fused.cpu().numpy(), two.cpu().numpy()) <= TOL
        torch.cuda.synchronize()
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g):
            out = unit.reverse(y)
        for _ in range(5):
            out.zero_()
            g.replay()
            torch.cuda.synchronize()
            assert torch.equal(out, ref)
    assert _lib.hlp_timeouts() == 0 and not _lib.fault_pending()


def test_short_step_kernel_on_a_dword_aligned_view(dev):
    """Activations that are 4-byte but not 16-byte aligned (a view one float into a buffer): the kernel's 16-byte pieces -- LDS-DMA
    requests and stores -- are buffer accesses, which need dword alignment only (as on the role-split kernel)."""
    from fincflow_amd import _lib, ops
    B, G, Cq, H, W, K = 3, 4, 12, 20, 24, 3
    assert _lib.inverse_variant(B, G, Cq, H, W, K, K)["sec"] == 6
    ws = oracle.make_stored_weights(G, Cq, K, K, seed=5)
    wco = oracle.canonicalize(ws, G, ORIENT_FASTFLOW)
    wc = ops.canonicalize(t(ws, dev), G, ORIENT_FASTFLOW)
    rng = np.random.default_rng(1)
    x = rng.standard_normal((B, G * Cq, H, W)).astype(np.float32)
    z = oracle.forward_f32(x, wco, G, ORIENT_FASTFLOW)
    n = z.size
    buf_in = torch.zeros(n + 1, device=dev)
    buf_out = torch.zeros(n + 1, device=dev)
    zin = buf_in[1:].view(B, G * Cq, H, W)
    zin.copy_(t(z, dev))
    out = buf_out[1:].view(B, G * Cq, H, W)
    assert zin.data_ptr() % 16 == 4 and out.data_ptr() % 16 == 4
    ops.finc_inverse(zin, wc, G, ORIENT_FASTFLOW, out=out)
    want = oracle.inverse_via_f64(z, wco, G, ORIENT_FASTFLOW)
    assert rel_err(out.cpu().numpy(), want) <= TOL


# ---------------------------------------------------------------------------------------------------------------------------------
# The band split (finc_split.hip, BSP) in its round-5 form: one workgroup per band, ticket order, a launch owns its progress words
# ---------------------------------------------------------------------------------------------------------------------------------
def _band_split_unit(dev, B, C, H, W, seed):
    from fincflow_amd import FastFlowUnit, _lib
    v = _lib.inverse_variant(B, 4, C // 4, H, W, 3, 3)
    assert v is not None and v["sec"] == 4 and v["workgroups"] == ((H + 15) // 16) * B * 4, v
    torch.manual_seed(seed)
    unit = FastFlowUnit(C, C, 3).to(dev)
    x = torch.randn(B, C, H, W, device=dev)
    with torch.no_grad():
        z, _ = unit(x)
        want = ops_strict(unit, z)
    return unit, z, want


def ops_strict(unit, z):
    from fincflow_amd import ops
    return ops.finc_inverse(z, unit._cache.w_canon, algo="strict")


def test_band_split_launches_on_two_streams_do_not_share_progress_words(dev):
    """launches in flight on different streams must not index the same words.  Two problem sets of different
    sizes, 40 launches each, enqueued far ahead on two streams at once (more launches in flight than there are slots: the ones that
    find none run the chained form) -- every result against the strict kernel's, no wait gave up."""
    from fincflow_amd import _lib
    ua, za, wa = _band_split_unit(dev, 8, 96, 64, 64, 1)
    ub, zb, wb = _band_split_unit(dev, 3, 96, 100, 68, 2)
    with torch.no_grad():
        ua.reverse(za); ub.reverse(zb)                      # (packs the weights outside the streams)
    torch.cuda.synchronize()
    s1, s2 = torch.cuda.Stream(dev), torch.cuda.Stream(dev)
    outs_a, outs_b = [], []
    with torch.no_grad():
        for _ in range(40):
            with torch.cuda.stream(s1):
                outs_a.append(ua.reverse(za))
            with torch.cuda.stream(s2):
                outs_b.append(ub.reverse(zb))
    torch.cuda.synchronize()
    for o in outs_a:
        assert rel_err(o.cpu().numpy(), wa.cpu().numpy()) <= TOL
    for o in outs_b:
        assert rel_err(o.cpu().numpy(), wb.cpu().numpy()) <= TOL
    assert all(torch.equal(o, outs_a[0]) for o in outs_a) and all(torch.equal(o, outs_b[0]) for o in outs_b)
    assert _lib.hlp_timeouts() == 0 and not _lib.fault_pending()


def test_band_split_graph_replays_beside_eager_launches(dev):
    """A captured band-split launch keeps a slot of its own (its memset node zeroes the words at every replay); eager launches on
    another stream take theirs from the event-tracked pool: replays and eager launches run side by side."""
    from fincflow_amd import _lib
    ua, za, wa = _band_split_unit(dev, 8, 96, 64, 64, 3)
    ub, zb, wb = _band_split_unit(dev, 4, 96, 48, 64, 4)
    with torch.no_grad():
        ua.reverse(za); ub.reverse(zb)
        torch.cuda.synchronize()
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g):
            out = ua.reverse(za)
        s2 = torch.cuda.Stream(dev)
        eager = []
        for _ in range(12):
            out.zero_()
            g.replay()
            with torch.cuda.stream(s2):
                for _ in range(3):
                    eager.append(ub.reverse(zb))
            torch.cuda.synchronize()
            assert rel_err(out.cpu().numpy(), wa.cpu().numpy()) <= TOL
    for o in eager:
        assert rel_err(o.cpu().numpy(), wb.cpu().numpy()) <= TOL
    assert _lib.hlp_timeouts() == 0 and not _lib.fault_pending()


def test_band_split_beside_a_kernel_that_fills_the_chip(dev):
    """the band split on one stream while the full-batch c3 inverse (1,024 one-wave problems: every SIMD of every
    compute unit) runs on another, so that the band split's workgroups are placed a few at a time as the other kernel's retire.  A
    workgroup waits only for the band above, whose workgroup drew an earlier ticket and is therefore running or done: the result is
    the oracle's, no wait gives up, no fault is pending."""
    from fincflow_amd import FastFlowUnit, _lib
    ub, zb, wb = _band_split_unit(dev, 16, 96, 64, 64, 5)
    torch.manual_seed(6)
    big = FastFlowUnit(96, 96, 3).to(dev)
    zbig = torch.randn(256, 96, 64, 64, device=dev)
    with torch.no_grad():
        big.reverse(zbig); ub.reverse(zb)
    torch.cuda.synchronize()
    s1, s2 = torch.cuda.Stream(dev), torch.cuda.Stream(dev)
    outs = []
    with torch.no_grad():
        for _ in range(6):
            with torch.cuda.stream(s1):
                for _ in range(4):
                    big.reverse(zbig)
            with torch.cuda.stream(s2):
                for _ in range(6):
                    outs.append(ub.reverse(zb))
    torch.cuda.synchronize()
    ws = torch.cat(ub._weights()).detach().cpu().numpy()
    wco = oracle.canonicalize(ws, 4, ORIENT_FASTFLOW)
    want = oracle.inverse_via_f64(zb[:2].cpu().numpy(), wco, 4, ORIENT_FASTFLOW, nthreads=8)
    for o in outs:
        assert rel_err(o[:2].cpu().numpy(), want) <= TOL
        assert torch.equal(o, outs[0])
    assert rel_err(outs[0].cpu().numpy(), wb.cpu().numpy()) <= TOL
    assert _lib.hlp_timeouts() == 0 and not _lib.fault_pending()


# ---------------------------------------------------------------------------------------------------------------------------------
# Double precision on the matrix cores (finc_f64.hip): FINC_ALGO_AUTO for float64 tensors
# ---------------------------------------------------------------------------------------------------------------------------------
def _oracle_f64_unit(z, wco, G, orient, Cq):
    """The reference-order fp64 solve per group in its own orientation (flip in, flip out: layers/conv.py:113-163)."""
    ref = np.empty_like(z)
    for g in range(G):
        og = (orient >> (2 * g)) & 3
        sl = slice(g * Cq, (g + 1) * Cq)
        zc = z[:, sl]
        if og & 1:
            zc = zc[:, :, :, ::-1]
        if og & 2:
            zc = zc[:, :, ::-1, :]
        xc = oracle.inverse_f64(np.ascontiguousarray(zc), wco[sl], 1)
        if og & 1:
            xc = xc[:, :, :, ::-1]
        if og & 2:
            xc = xc[:, :, ::-1, :]
        ref[:, sl] = xc
    return ref


# (B, G, Cq, H, W, K, orient or None = FastFlow's): every bank of finc_f64.hip, padded channel counts, one band / several / a partial
# last one, W < 16 (fewer rows than lanes), W == 16, wide maps, every orientation, the c2 and c3 banks at their map sizes
F64_CASES = [(2, 4, 12, 32, 32, 3, None), (1, 4, 24, 64, 64, 3, None), (2, 1, 5, 9, 11, 3, 0), (1, 1, 3, 7, 7, 3, 1), (2, 4, 6, 8, 12, 3, None),
             (1, 1, 4, 6, 5, 2, 3), (3, 2, 16, 33, 16, 3, 6), (1, 4, 20, 17, 40, 3, None), (2, 4, 23, 40, 24, 3, None), (1, 3, 8, 50, 100, 3, 0x1B),
             (2, 4, 32, 20, 36, 2, None), (1, 4, 9, 21, 4, 2, None), (4, 4, 1, 5, 8, 3, None)]


@pytest.mark.parametrize("shape", F64_CASES, ids=lambda c: "B%d_G%d_Cq%d_%dx%d_k%d_o%s" % c)
def test_fp64_inverse_and_forward_on_the_matrix_cores(shape, dev):
    """finc_inverse_f64_algo / finc_forward_f64_algo under FINC_ALGO_AUTO (v_mfma_f64_16x16x4_f64, the bank folded in fp64): within
    1e-12 of the reference-order fp64 solve (the oracle's inverse_f64 = solve_parallel_mc.pyx:77-126, bit-equal to the rebuilt
    .pyx; the strict kernel bit-exact beside it) and of the fp64 forward; `mfma` insists on the matrix-core form."""
    from fincflow_amd import ops
    B, G, Cq, H, W, K, o = shape
    orient = ORIENT_FASTFLOW if o is None else o
    rng = np.random.default_rng(sum(shape[:6]))
    ws = oracle.make_stored_weights(G, Cq, K, K, orient=orient, seed=3).astype(np.float64)
    ws += 1e-9 * rng.standard_normal(ws.shape) * (ws != 0) * (ws != 1)      # (bits below fp32: a kernel that narrows anything shows)
    wco64 = oracle.canonicalize(ws.astype(np.float32), G, orient).astype(np.float64)
    wc = ops.canonicalize(t(ws, dev), G, orient)
    wco = wc.cpu().numpy()
    assert wc.dtype == torch.float64 and wco.shape == wco64.shape
    z = rng.standard_normal((B, G * Cq, H, W))
    ref = _oracle_f64_unit(z, wco, G, orient, Cq)
    strict = ops.finc_inverse(t(z, dev), wc, G, orient, algo="strict").cpu().numpy()
    assert np.array_equal(strict, ref)
    fast = ops.finc_inverse(t(z, dev), wc, G, orient, algo="mfma").cpu().numpy()
    auto = ops.finc_inverse(t(z, dev), wc, G, orient).cpu().numpy()
    assert np.array_equal(fast, auto)
    assert rel_err(fast, ref) <= 1e-12, rel_err(fast, ref)
    zf_strict = ops.finc_forward(t(ref, dev), wc, G, orient, algo="strict").cpu().numpy()
    zf = ops.finc_forward(t(ref, dev), wc, G, orient, algo="mfma").cpu().numpy()
    assert rel_err(zf, zf_strict) <= 1e-13 and rel_err(zf, z) <= 1e-12, (rel_err(zf, zf_strict), rel_err(zf, z))


@pytest.mark.parametrize("name", ["unit_c1_B2_C4_8x8_k3", "unit_B2_C8_6x9_k3", "unit_B1_C12_16x16_k3", "unit_B1_C24_8x8_k3", "unit_B2_C48_32x32_k3",
                                  "unit_B1_C96_16x16_k3", "unit_B1_C16_12x12_k2"])
def test_fp64_matrix_core_inverse_on_the_reference_fixtures(name, dev):
    """The reference's own CPU path (reverse_cython: fp32 -> fp64 solve -> fp32, layers/conv.py:113-163) recorded in tests/golden:
    the fp64 matrix-core inverse of the fixture's z, rounded to fp32 as the reference rounds, against x_rev_cython."""
    from fincflow_amd import ops
    from helpers import golden, unit_stored_weights
    g = golden(name)
    ws = unit_stored_weights(g).astype(np.float64)
    wc = ops.canonicalize(t(ws, dev), 4, ORIENT_FASTFLOW)
    x64 = ops.finc_inverse(t(g["z"].astype(np.float64), dev), wc, 4, ORIENT_FASTFLOW, algo="mfma").cpu().numpy()
    x32 = x64.astype(np.float32)
    want = g["x_rev_cython"]
    assert rel_err(x32, want) <= 1e-7, rel_err(x32, want)
    assert np.mean(x32 == want) >= 0.999             # (the same fp32 value wherever the fp64 results do not straddle a rounding boundary)


# (B, G, Cq, H, W): the 28-channel 3x3 bank -- and the counts 25 .. 27 that pad to it -- on the 32-channel bank's packed two-wave kernel
# (finc_mfma.hip, borrowed_cqp): even problem counts up to 512 on narrow and wide maps (16-byte, 32-byte and sector I/O), G = 1, 2, 4, 8
BORROWED_CASES = [(65, 4, 28, 20, 24), (65, 4, 25, 9, 20), (130, 2, 26, 18, 32), (258, 1, 27, 5, 16), (33, 8, 28, 33, 40), (128, 4, 28, 7, 64)]


@pytest.mark.parametrize("case", BORROWED_CASES, ids=lambda c: "B%d_G%d_Cq%d_%dx%d" % c)
def test_the_28_channel_bank_on_the_borrowed_two_wave_kernel(case, dev):
    """The library says it runs the 32-channel bank's packed two-wave kernel (its own bank holds 7 k-steps: no two-wave form), the
    result is the oracle's (<= 1e-5, strict kernel bit-exact beside it), and the next odd count runs the bank's own one-wave kernel
    on the same packed buffer -- both banks live in it."""
    from fincflow_amd import _lib
    B, G, Cq, H, W = case
    orient = ORIENT_FASTFLOW if G == 4 else (0x1B & ((1 << (2 * G)) - 1)) if G < 4 else 0x1BE4
    v = _lib.inverse_variant(B, G, Cq, H, W, 3, 3)
    assert v["cqp"] == 32 and v["nw"] == 2 and v["npw"] == 2 and v["workgroups"] == B * G // 2, v
    run_inverse_case(dev, B, G, orient, Cq, H, W, 3, 3, seed=7 * Cq + H + W, tag="borrowed")
    if G == 1:
        own = _lib.inverse_variant(B + 1, G, Cq, H, W, 3, 3)
        assert own["cqp"] == 28 and own["nw"] == 1, own
        run_inverse_case(dev, B + 1, G, orient, Cq, H, W, 3, 3, seed=5, tag="borrowed_own")


def test_borrowed_bank_through_the_module_with_the_affine_fold(dev):
    """FastFlowUnit at C = 112 (28 channels per group), 64x64: B = 128 runs the borrowed two-wave kernel, B = 129 the bank's own -- one
    cached packed buffer serves both; reverse(forward(x)) = x, the first images against the oracle, and ActNorm folded into the
    borrowed bank (scale and shift) gives what the two layers give."""
    from fincflow_amd import FastFlowUnit, _lib, glow
    C, H, W = 112, 64, 64
    torch.manual_seed(5)
    unit = FastFlowUnit(C, C, 3).to(dev)
    with torch.no_grad():
        for m in (unit.conv_tl, unit.conv_tr, unit.conv_bl, unit.conv_br):
            m.conv.weight.mul_(1 - (1 - bank_std(28, 3) / 0.05) * m.get_mask().to(dev))
    an = glow.ActNorm(C).to(dev)
    with torch.no_grad():
        an.log_scale.copy_(0.3 * torch.randn(C, device=dev))
        an.translation.copy_(torch.randn(C, device=dev))
        an.initialized.fill_(1)
    assert _lib.inverse_variant(128, 4, 28, H, W, 3, 3)["cqp"] == 32 and _lib.inverse_variant(129, 4, 28, H, W, 3, 3)["cqp"] == 28
    assert _lib.inverse_variant(192, 4, 28, H, W, 3, 3)["cqp"] == 28 and _lib.inverse_variant(256, 4, 28, H, W, 3, 3)["cqp"] == 28
    x = torch.randn(129, C, H, W, device=dev)
    wco = oracle.canonicalize(torch.cat(unit._weights()).detach().cpu().numpy(), 4, ORIENT_FASTFLOW)
    with torch.no_grad():
        z = unit(x)[0]
        a = unit.reverse(z[:128])                      # borrowed
        b = unit.reverse(z)                            # own
        assert rel_err(a.cpu().numpy(), x[:128].cpu().numpy()) <= 1e-4 and rel_err(b.cpu().numpy(), x.cpu().numpy()) <= 1e-4
        want = oracle.inverse_via_f64(z[:2].cpu().numpy(), wco, 4, ORIENT_FASTFLOW)
        assert rel_err(a[:2].cpu().numpy(), want) <= TOL and rel_err(b[:2].cpu().numpy(), want) <= TOL
        y = torch.randn(128, C, H, W, device=dev)
        two = unit.reverse(an.reverse(y))
        fused = unit.reverse_affine(y, an.log_scale, an.translation)
        assert fused is not None and rel_err(fused.cpu().numpy(), two.cpu().numpy()) <= TOL
        zz = (y[:2] * torch.exp(an.log_scale).view(1, -1, 1, 1) + an.translation.view(1, -1, 1, 1)).cpu().numpy()
        assert rel_err(fused[:2].cpu().numpy(), oracle.inverse_via_f64(zz, wco, 4, ORIENT_FASTFLOW)) <= TOL


# (B, G, Cq, H, W, K): problem sets of whole rounds of one-wave problems plus a remainder that the library hands to the kernel it would
# pick for the remainder alone (finc_mfma.hip finc_mfma_launch) -- role-split kernel (16 and 256 problems behind 1,024), short-step
# kernel (12 channels: 40 and 512 behind), the small-batch two-wave variant (24 channels: 260 .. 512 behind), G = 1, 2, 4, two rounds
REMAINDER_CASES = [(260, 4, 24, 8, 16, 3, 4), (320, 4, 24, 5, 16, 3, 64), (330, 4, 24, 4, 16, 3, 74), (384, 4, 24, 4, 16, 3, 128), (266, 4, 12, 8, 16, 3, 10),
                   (384, 4, 12, 4, 16, 3, 128), (1100, 1, 24, 4, 16, 3, 76), (522, 4, 24, 3, 16, 3, 10), (300, 4, 16, 6, 16, 2, 44),
                   # 20 channels, 276 problems behind a round: no other kernel would take them (above the role-split kernel's 256, no two-wave
                   # variant, the short-step kernel ends at 16 channels) -- one launch
                   (650, 2, 20, 3, 16, 3, 0),
                   # the packed two-wave kernels (32 channels): rounds of 512
                   (140, 4, 32, 4, 16, 3, 12), (192, 4, 32, 3, 16, 2, 64), (150, 4, 30, 5, 16, 3, 22)]
# (the last number: the images the library says it hands to the second launch)


@pytest.mark.parametrize("case", REMAINDER_CASES, ids=lambda c: "B%d_G%d_Cq%d_%dx%d_k%d" % c[:6])
def test_remainder_of_a_round_runs_on_the_remainders_own_kernel(case, dev):
    """The images behind the whole rounds come from a second launch on another kernel: every image against the oracle (<= 1e-5), the
    strict kernel bit-exact, and the images on either side of the seam equal to what the same images give in a call of their own."""
    from fincflow_amd import _lib, ops
    B, G, Cq, H, W, K, r = case
    orient = ORIENT_FASTFLOW if G == 4 else (0x1B & ((1 << (2 * G)) - 1))
    v = _lib.inverse_variant(B, G, Cq, H, W, K, K)
    assert (v["nw"], v["npw"]) == ((2, 2) if Cq > 28 else (1, 1)) and v["sec"] in (1, 2, 3), v   # the call's main kernel
    rnd = 512 if Cq > 28 else 1024
    assert _lib.inverse_remainder_images(B, G, Cq, H, W, K, K) == r and r in (0, (B * G) % rnd // G)          # the launch's own answer
    assert _lib.inverse_remainder_images(B - r, G, Cq, H, W, K, K) == 0 or r == 0
    run_inverse_case(dev, B, G, orient, Cq, H, W, K, K, seed=B + Cq, tag="remainder")
    rng = np.random.default_rng(B)
    ws = oracle.make_stored_weights(G, Cq, K, K, orient=orient, seed=3, std=bank_std(Cq, K))
    wc = ops.canonicalize(t(ws, dev), G, orient)
    z = t(rng.standard_normal((B, G * Cq, H, W)).astype(np.float32), dev)
    whole = ops.finc_inverse(z, wc, G, orient)
    if r == 0:
        return
    tail = ops.finc_inverse(z[B - r:].contiguous(), wc, G, orient)
    assert r <= 512 // G and torch.equal(whole[B - r:], tail)                 # the remainder's kernel, on the remainder's images
    head = ops.finc_inverse(z[:B - r].contiguous(), wc, G, orient)
    assert torch.equal(whole[:B - r], head)


def test_a_flow_stack_at_a_batch_with_a_remainder_keeps_the_plain_chain(dev):
    """[unit, ActNorm, Conv1x1] reversed at B = 260 (1,040 problems = a round + 16): the premultiplied-input form is one launch or nothing
    (finc_mfma.hip remainder_images), so the container runs the plain inverse -- two launches -- and gives what the layer-by-layer
    chain gives and what went in; at B = 256 it takes the fused path as before."""
    from fincflow_amd import FastFlowUnit, FlowSequential, glow, _lib
    from fincflow_amd.layers import StandardNormal
    C, H, W = 96, 4, 16
    q = _lib.lib().finc_inverse_premultiplied_supported
    assert q(256, 4, 24, H, W, 3, 3) == 1 and q(260, 4, 24, H, W, 3, 3) == 0
    torch.manual_seed(11)
    unit = FastFlowUnit(C, C, 3).to(dev)
    an, mix = glow.ActNorm(C).to(dev), glow.Conv1x1(C).to(dev)
    with torch.no_grad():
        an.log_scale.copy_(0.2 * torch.randn(C, device=dev))
        an.translation.copy_(torch.randn(C, device=dev))
        an.mark_initialized()
    seq = FlowSequential(StandardNormal((C, H, W)), unit, an, mix)
    for B in (256, 260):
        x = torch.randn(B, C, H, W, device=dev)
        with torch.no_grad():
            zz = seq(x)[0]
            a = seq._reverse_chain(zz, None)
            b = seq._reverse_chain(zz, None, fuse=False)
        assert rel_err(a.cpu().numpy(), b.cpu().numpy()) <= TOL and rel_err(a.cpu().numpy(), x.cpu().numpy()) <= 1e-4
    assert _lib.hlp_timeouts() == 0
