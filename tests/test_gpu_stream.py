"""The streaming-bank kernels (finc_stream.hip): the banks no register-resident kernel of the library holds -- 3x3 above 96
channels per group (`CINCFlowUnit` at C = 192, cinc_flow.py:9-30), 5x5 above 48, 2x2 above 32, 4x4 / 6x6 / 7x7 and the
non-square filters above 16 channels (layers/conv.py:30-36 takes any tuple) -- through the C ABI against the oracle.

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


def orient_of(G):
    return ORIENT_FASTFLOW if G == 4 else (0x1B & ((1 << (2 * G)) - 1))


# (B, G, Cq, H, W, KH, KW): every tile count (16, 32, 48 padded channels on one wave; 64, 128, 192, 256 on four), full and padded channel counts, one band / partial
# last band / several bands, maps narrower than a band's 16 rows and than the filter, a single row, widths that are no multiple of
# four, G = 1 / 2 / 4 / 8 (the XCD-aware problem map and the plain one), every filter family the tables leave out
STREAM_CASES = [
    (2, 1, 192, 24, 20, 3, 3), (1, 4, 100, 17, 33, 3, 3), (1, 1, 256, 16, 16, 3, 3), (8, 1, 130, 8, 8, 3, 3), (4, 2, 110, 8, 12, 3, 3),
    (1, 4, 129, 1, 50, 3, 3), (2, 4, 8, 20, 24, 4, 4), (1, 4, 4, 33, 18, 7, 7), (2, 2, 64, 16, 16, 5, 5), (3, 4, 40, 9, 40, 2, 2),
    (2, 4, 32, 12, 28, 3, 5), (2, 4, 20, 5, 3, 6, 6), (1, 8, 50, 35, 7, 5, 5), (2, 4, 24, 40, 19, 5, 3), (3, 3, 17, 18, 64, 1, 7),
    (2, 4, 104, 20, 24, 3, 3), (1, 4, 56, 33, 36, 5, 5), (2, 4, 72, 18, 8, 2, 2),     # 16-byte loads and stores in all four orientations
    # one-wave problems in their per-lane 16-byte form (problems x padded channels > 10,240 and more than one problem per compute
    # unit; the launch keeps the dword form below that), 1 / 2 / 3 tiles -- and the dword form just below the rule's edge
    (161, 4, 12, 8, 8, 4, 4), (129, 4, 20, 20, 8, 3, 5), (65, 8, 40, 5, 12, 2, 2), (160, 4, 12, 8, 8, 4, 4), (64, 4, 40, 8, 8, 2, 2),
]
ONE_WAVE_VEC = {(161, 4, 12, 8, 8, 4, 4), (129, 4, 20, 20, 8, 3, 5), (65, 8, 40, 5, 12, 2, 2)}


def case_id(c):
    return "B%d_G%d_Cq%d_%dx%d_k%dx%d" % c


@pytest.mark.parametrize("case", STREAM_CASES, ids=case_id)
def test_streaming_bank_inverse(case, dev):
    """Inverse on the streaming-bank kernel (form 7) against the oracle's fp64 path (<= 1e-5 of the largest entry, element-wise 1e-3),
    the strict kernel bit-exact beside it, repeated launches bit-identical."""
    from fincflow_amd import _lib, ops
    B, G, Cq, H, W, KH, KW = case
    orient = orient_of(G)
    v = _lib.inverse_variant(B, G, Cq, H, W, KH, KW)
    nw = 1 if Cq <= 48 else 4              # one-wave problems up to 48 channels, a workgroup of four waves beyond
    assert v is not None and v["sec"] == 7 and v["nw"] == nw and v["workgroups"] == B * G, v
    assert v["row"] == (-4 if case in ONE_WAVE_VEC else -3), v          # -4: one-wave problems in their per-lane 16-byte form
    assert v["cqp"] == (Cq + 16 * nw - 1) // (16 * nw) * (16 * nw)
    e_max, _ = run_inverse_case(dev, B, G, orient, Cq, H, W, KH, KW, seed=13 * Cq + H + 3 * W + KH, tag="stream")
    assert e_max <= TOL
    rng = np.random.default_rng(3)
    ws = oracle.make_stored_weights(G, Cq, KH, KW, orient=orient, seed=8, std=bank_std(Cq, max(KH, KW)))
    wc = ops.canonicalize(t(ws, dev), G, orient)
    z = t(rng.standard_normal((B, G * Cq, H, W)).astype(np.float32), dev)
    a = ops.finc_inverse(z, wc, G, orient)
    b = ops.finc_inverse(z, wc, G, orient)
    assert torch.equal(a, b)


@pytest.mark.parametrize("case", STREAM_CASES, ids=case_id)
def test_streaming_bank_forward_and_input_gradient(case, dev):
    """Forward and grad-input (the same kernel on the transposed bank and the flipped image) against the oracle's fp64-accumulated
    forward and CPU fp64 autograd; the weight gradient on the tile-pair kernels, checked entry by entry with its mask."""
    import torch.nn.functional as F
    from fincflow_amd import _lib, ops
    B, G, Cq, H, W, KH, KW = case
    orient = orient_of(G)
    L = _lib.lib()
    assert L.finc_forward_algo_for(Cq, H, W, KH, KW) == _lib.ALGO["mfma"]
    bv = _lib.backward_variant(B, G, Cq, H, W, KH, KW)
    assert bv["conv_form"] == "stream" and bv["gradx_waves"] == (1 if Cq <= 48 else 4), bv
    # the weight gradient: the tile-pair kernels take any number of tiles (finc_gradw.hip) -- Winograd at 3x3 / 5x5 on maps of at least
    # one strip, the direct tile-pair form for the other filters up to 5 columns wide; beyond that (6x6, 7x7, W % 4 != 0) the direct kernel
    want_gw = ("direct" if (W % 4 or KW > 5 or (KH, KW) not in ((2, 2), (3, 3), (4, 4), (5, 5), (2, 3), (3, 2), (3, 5), (5, 3))) else
               "winograd_tiled" if ((KH, KW) == (3, 3) and W >= 32) or ((KH, KW) == (5, 5) and W >= 16) else "tiled")
    assert bv["gradw"] == want_gw, (bv, want_gw)
    rng = np.random.default_rng(11 * Cq + W)
    ws = oracle.make_stored_weights(G, Cq, KH, KW, orient=orient, seed=5, std=bank_std(Cq, max(KH, KW)))
    wco = oracle.canonicalize(ws, G, orient)
    x = rng.standard_normal((B, G * Cq, H, W)).astype(np.float32)
    gz = rng.standard_normal((B, G * Cq, H, W)).astype(np.float32)
    wc = ops.canonicalize(t(ws, dev), G, orient)
    z = ops.finc_forward(t(x, dev), wc, G, orient)
    z_ref = oracle.forward_f32(x, wco, G, orient, accumulate_f64=True)
    assert rel_err(z.cpu().numpy(), z_ref) <= TOL
    zs = ops.finc_forward(t(x, dev), wc, G, orient, algo="strict")
    assert rel_err(z.cpu().numpy(), zs.cpu().numpy()) <= TOL
    gx, gw = ops.finc_backward(t(gz, dev), t(x, dev), wc, G, orient)
    # CPU fp64 autograd of the same map in canonical coordinates: group g's image flipped per its orientation, top-left padded conv
    xd = torch.from_numpy(x).double().requires_grad_(True)
    wd = torch.from_numpy(wco).double().requires_grad_(True)
    outs = []
    for g in range(G):
        o = (orient >> (2 * g)) & 3
        dims = [d for d, bit in ((2, 2), (3, 1)) if o & bit]
        xg = xd[:, g * Cq:(g + 1) * Cq]
        xg = torch.flip(xg, dims) if dims else xg
        y = F.conv2d(F.pad(xg, (KW - 1, 0, KH - 1, 0)), wd[g * Cq:(g + 1) * Cq])
        outs.append(torch.flip(y, dims) if dims else y)
    ref = torch.cat(outs, 1)
    assert rel_err(z.cpu().numpy(), ref.detach().numpy()) <= TOL
    ref.backward(torch.from_numpy(gz).double())
    assert rel_err(gx.cpu().numpy(), xd.grad.numpy()) <= TOL
    want = wd.grad.numpy().copy()
    corner = want[:, :, KH - 1, KW - 1].reshape(G, Cq, Cq)
    corner[:, np.triu_indices(Cq)[0], np.triu_indices(Cq)[1]] = 0.0           # PaddedConv2d.reset_gradients (layers/conv.py:98-99)
    assert rel_err(gw.cpu().numpy(), want) <= 2e-5


def test_cinc_unit_at_192_channels(dev):
    """`CINCFlowUnit(192, 192, 3)` (cinc_flow.py:9-30, groups = 1): forward, reverse and the round trip on the streaming-bank kernels."""
    from fincflow_amd import CINCFlowUnit, _lib
    torch.manual_seed(7)
    C, H, W, B = 192, 32, 32, 3
    unit = CINCFlowUnit(C, C, 3).to(dev)
    with torch.no_grad():      # (the reference's init std 0.05 has no stable inverse at 192 channels: the bank's own scale, as everywhere)
        unit.conv_tl.conv.weight.copy_(t(oracle.make_stored_weights(1, C, 3, 3, orient=0, seed=3, std=bank_std(C, 3)), dev))
    L = _lib.lib()
    assert L.finc_inverse_algo_for(C, H, W, 3, 3) == _lib.ALGO["mfma"] and L.finc_forward_algo_for(C, H, W, 3, 3) == _lib.ALGO["mfma"]
    v = _lib.inverse_variant(B, 1, C, H, W, 3, 3)
    assert v["sec"] == 7, v
    x = torch.randn(B, C, H, W, device=dev)
    with torch.no_grad():
        z, logdet = unit(x)
        xr = unit.reverse(z)
    assert logdet == 0.0
    w = unit.conv_tl.conv.weight.detach().cpu().numpy()
    wco = oracle.canonicalize(w, 1, 0)
    z_ref = oracle.forward_f32(x.cpu().numpy(), wco, 1, 0, accumulate_f64=True)
    x_ref = oracle.inverse_via_f64(z.cpu().numpy(), wco, 1, 0)
    assert rel_err(z.cpu().numpy(), z_ref) <= TOL
    assert rel_err(xr.cpu().numpy(), x_ref) <= TOL
    assert rel_err(xr.cpu().numpy(), x.cpu().numpy()) <= 1e-4


@pytest.mark.parametrize("case", [(2, 4, 40, 18, 24, 2, 2), (2, 1, 160, 20, 16, 3, 3), (130, 4, 12, 8, 8, 4, 4), (129, 4, 36, 18, 8, 2, 2)], ids=case_id)
def test_streaming_bank_carries_the_affine_folds(case, dev):
    """ActNorm folded into the streaming banks: scale into the z-term's columns and Linv * shift as the accumulators' start (inverse,
    layers/actnorm.py:39-52), filter rows scaled and accumulators started from the shift (forward)."""
    from fincflow_amd import ops
    B, G, Cq, H, W, KH, KW = case
    orient = orient_of(G)
    rng = np.random.default_rng(77)
    ws = oracle.make_stored_weights(G, Cq, KH, KW, orient=orient, seed=2, std=bank_std(Cq, max(KH, KW)))
    wst = t(ws, dev)
    per = ws.shape[0] // G
    weights = [wst[i * per:(i + 1) * per].clone() for i in range(G)]
    log_scale = t((0.1 * rng.standard_normal((1, G * Cq, 1, 1))).astype(np.float32), dev)
    translation = t((0.2 * rng.standard_normal((1, G * Cq, 1, 1))).astype(np.float32), dev)
    cache = ops.PackedWeights()
    y = t(rng.standard_normal((B, G * Cq, H, W)).astype(np.float32), dev)
    got = cache.inverse_affine(y, weights, G, orient, log_scale, translation)
    assert got is not None, "the streaming bank carries scale and shift"
    want = cache.inverse(torch.exp(log_scale) * y + translation, weights, G, orient)
    assert rel_err(got.cpu().numpy(), want.cpu().numpy()) <= TOL
    gotf = cache.forward_affine(y, weights, G, orient, log_scale, translation)
    assert gotf is not None
    wantf = (cache.forward(y, weights, G, orient) - translation) * torch.exp(-log_scale)
    assert rel_err(gotf.cpu().numpy(), wantf.cpu().numpy()) <= TOL


@pytest.mark.parametrize("case", [(3, 160, 12, 16, (2, 2)), (2, 32, 9, 12, (4, 4)), (2, 448, 8, 8, (3, 3))], ids=lambda c: "B%d_C%d_%dx%d_k%dx%d" % (c[0], c[1], c[2], c[3], c[4][0], c[4][1]))
def test_fastflowunit_module_on_streaming_banks(case, dev):
    """The layer API (fastflow.py:15-55) on banks that only the streaming-bank kernels hold: forward under autograd, backward (masked weight
    gradients in the four stored orientations), reverse -- against CPU fp64 autograd of the reference's own formulation (pad + conv2d per chunk)."""
    import torch.nn.functional as F
    from fincflow_amd import FastFlowUnit, _lib
    B, C, H, W, (KH, KW) = case
    Cq = C // 4
    assert _lib.inverse_variant(B, 4, Cq, H, W, KH, KW)["sec"] == 7
    torch.manual_seed(C + H)
    unit = FastFlowUnit(C, C, (KH, KW)).to(dev)
    convs = (unit.conv_tl, unit.conv_tr, unit.conv_bl, unit.conv_br)
    with torch.no_grad():          # the free taps at the bank's own scale (init std 0.05 has no stable inverse at these widths), corner tap kept
        for m in convs:
            m.conv.weight.mul_(1 - (1 - bank_std(Cq, max(KH, KW)) / 0.05) * m.get_mask().to(dev))
    x = torch.randn(B, C, H, W, device=dev)
    gz = torch.randn(B, C, H, W, device=dev)
    xg = x.clone().requires_grad_(True)
    z, logdet = unit(xg)
    z.backward(gz)
    assert logdet == 0.0
    with torch.no_grad():
        xr = unit.reverse(z.detach())
    assert rel_err(xr.cpu().numpy(), x.cpu().numpy()) <= 5e-5
    xd = x.cpu().double().requires_grad_(True)
    wds = [m.conv.weight.detach().cpu().double().requires_grad_(True) for m in convs]
    ref = torch.cat([F.conv2d(F.pad(c, m.pad), w) for m, c, w in zip(convs, torch.chunk(xd, 4, 1), wds)], 1)
    ref.backward(gz.cpu().double())
    assert rel_err(z.detach().cpu().numpy(), ref.detach().numpy()) <= TOL
    assert rel_err(xg.grad.cpu().numpy(), xd.grad.numpy()) <= TOL
    for m, w in zip(convs, wds):
        want = (w.grad * m.get_mask().cpu().double()).numpy()
        assert rel_err(m.conv.weight.grad.cpu().numpy(), want) <= 2e-5
    x_ref = oracle.inverse_via_f64(z.detach().cpu().numpy(), oracle.canonicalize(torch.cat(unit._weights()).detach().cpu().numpy(), 4, ORIENT_FASTFLOW), 4, ORIENT_FASTFLOW)
    assert rel_err(xr.cpu().numpy(), x_ref) <= TOL


@pytest.mark.parametrize("case", [(3, 448, 20, 24, (3, 3)), (164, 32, 8, 8, (4, 4))], ids=lambda c: "B%d_C%d_%dx%d_k%dx%d" % (c[0], c[1], c[2], c[3], c[4][0], c[4][1]))
def test_streaming_bank_launch_in_a_captured_graph_and_repeated(case, dev):
    """A sampling loop captures its launches (fincflow_amd/glow.py): the streaming-bank inverse replayed from a HIP graph
    gives the eager result bit for bit, and 200 eager launches give it every time (the kernel has no state between launches: no
    progress words, no workspace) -- four-wave problems with 16-byte I/O and 656 one-wave problems in their per-lane form."""
    from fincflow_amd import FastFlowUnit, _lib
    B, C, H, W, (KH, KW) = case
    Cq = C // 4
    v = _lib.inverse_variant(B, 4, Cq, H, W, KH, KW)
    assert v["sec"] == 7 and v["row"] == (-4 if Cq <= 48 else -3), v
    torch.manual_seed(3 * C + H)
    unit = FastFlowUnit(C, C, (KH, KW)).to(dev)
    with torch.no_grad():
        for m in (unit.conv_tl, unit.conv_tr, unit.conv_bl, unit.conv_br):
            m.conv.weight.mul_(1 - (1 - bank_std(Cq, max(KH, KW)) / 0.05) * m.get_mask().to(dev))
    y = torch.randn(B, C, H, W, device=dev)                  # the sampling distribution (train/losses.py:42-45)
    with torch.no_grad():
        ref = unit.reverse(y)
        wco = oracle.canonicalize(torch.cat(unit._weights()).detach().cpu().numpy(), 4, ORIENT_FASTFLOW)
        want = oracle.inverse_via_f64(y[:2].cpu().numpy(), wco, 4, ORIENT_FASTFLOW)
        assert rel_err(ref[:2].cpu().numpy(), want) <= TOL
        torch.cuda.synchronize()
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g):
            out = unit.reverse(y)
        for _ in range(5):
            out.zero_()
            g.replay()
            torch.cuda.synchronize()
            assert torch.equal(out, ref)
        o2 = torch.empty_like(ref)
        for i in range(200):
            o2.fill_(float("nan"))
            unit._cache.inverse(y, unit._weights(), 4, ORIENT_FASTFLOW, out=o2)
            if i % 20 == 19:
                assert torch.equal(o2, ref), i
    assert _lib.hlp_timeouts() == 0 and not _lib.fault_pending()
