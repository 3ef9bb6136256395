"""Round-4 parity cases (GPU; through the C ABI; nothing here reads /root/reference).

* every kernel family of the 3x3 forward / grad-input -- the direct strip kernel, Winograd F(2,3), Winograd F(4,3) --
  against REFERENCE-GENERATED data: the golden forward outputs (tests/golden/make_golden.py) and the golden input gradients
  (tests/golden/make_golden_r4.py: torch autograd through the reference's own layers), including the trained-like fixtures
  (F(4,3), the headline forward kernel, had only met the oracle and fp64 conv2d at init-scale weights);
* the affine fold on the maps finc_big.hip takes over from the 33..64-channel banks (a bank packed with a
  shift had an all-zero wide-map half and the launch computed with it);
* the packed-weight cache: an entry created by the training path is validated by the first inference call.
"""
import ctypes

import numpy as np
import pytest
import torch

from oracle import oracle
from helpers import ORDER_BITS, ORIENT_FASTFLOW, elem_rel_err, golden, golden_names, rel_err, report, unit_stored_weights

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


GRAD_CASES = [n[len("grad_"):] for n in golden_names("grad_")]
FORMS = {1: ("strip", "strip16"), 2: ("winograd",), 4: ("winograd4",)}


@pytest.mark.parametrize("name", GRAD_CASES)
def test_every_forward_form_against_reference_outputs_and_gradients(name, dev):
    """layers/conv.py:102-107 (forward) and its autograd (grad-input), each kernel family pinned in turn.  Max-normalised
    error <= 1e-5 (BASELINE.json) asserted; the element-wise error (every element against its own magnitude, down to 1e-3 of
    the largest) goes on record in gpurun_out/parity_report.jsonl."""
    from fincflow_amd import _lib, ops
    g, gg = golden(name), golden("grad_" + name)
    if name.startswith("unit_"):
        ws, G, orient = unit_stored_weights(g), 4, ORIENT_FASTFLOW
    else:
        ws, G, orient = g["w"], 1, ORDER_BITS[str(g["order"])]
    assert ws.shape[2:] == (3, 3) and g["x"].shape[3] % 4 == 0
    B, C, H, W = g["x"].shape
    wc = ops.canonicalize(t(ws, dev), G, orient)
    x, gz = t(g["x"], dev), t(gg["gz_times16"].astype(np.float32) / 16.0, dev)
    try:
        for form, names in FORMS.items():
            _lib.set_forward_form(form)
            assert _lib.backward_variant(B, G, C // G, H, W, 3, 3)["conv_form"] in names, (form, name)
            z = ops.finc_forward(x, wc, G, orient).cpu().numpy()
            gx, _ = ops.finc_backward(gz, x, wc, G, orient, need_gx=True, need_gw=False)
            gx = gx.cpu().numpy()
            e_f, e_g = rel_err(z, g["z"]), rel_err(gx, gg["grad_x"])
            report("forward_form_vs_reference", fixture=name, form=form, forward=e_f, grad_input=e_g,
                   forward_elementwise=elem_rel_err(z, g["z"]), grad_input_elementwise=elem_rel_err(gx, gg["grad_x"]))
            assert e_f <= TOL and e_g <= TOL, (name, form, e_f, e_g)
    finally:
        _lib.set_forward_form(0)


def test_f43_on_trained_like_weights_at_a_shape_the_library_sends_there(dev):
    """B = 64, C = 96, 64x64 (a 4-way strong split's share of configs[2]) with the free taps x 1.5: AUTO picks F(4,3); three
    images against the pinned oracle (fp64 accumulation) and against fp64 conv2d autograd for grad-input."""
    import torch.nn.functional as F
    from fincflow_amd import FastFlowUnit, _lib
    B, C, H, W = 64, 96, 64, 64
    assert _lib.backward_variant(B, 4, C // 4, H, W, 3, 3)["conv_form"] == "winograd4"
    torch.manual_seed(404)
    unit = FastFlowUnit(C, C, 3)
    with torch.no_grad():
        for m in (unit.conv_tl, unit.conv_tr, unit.conv_bl, unit.conv_br):
            m.conv.weight.mul_(1 + 0.5 * m.get_mask())                      # make_golden.unit_case's `heavy`, 1.5
    unit = unit.to(dev)
    x = torch.randn(B, C, H, W, device=dev)
    gz = torch.randn(B, C, H, W, device=dev)
    xg = x.clone().requires_grad_(True)
    z, logdet = unit(xg)
    z.backward(gz)
    assert logdet == 0.0
    pick = [0, 31, 63]
    ws = torch.cat(unit._weights()).detach().cpu().numpy()
    wco = oracle.canonicalize(ws, 4, ORIENT_FASTFLOW)
    z_ref = oracle.forward_f32(x[pick].cpu().numpy(), wco, 4, ORIENT_FASTFLOW, accumulate_f64=True)
    xd = x[pick].cpu().double().requires_grad_(True)
    ref = torch.cat([F.conv2d(F.pad(c, m.pad), m.conv.weight.detach().cpu().double()) for m, c in
                     zip((unit.conv_tl, unit.conv_tr, unit.conv_bl, unit.conv_br), torch.chunk(xd, 4, 1))], 1)
    ref.backward(gz[pick].cpu().double())
    e_f = rel_err(z[pick].detach().cpu().numpy(), z_ref)
    e_g = rel_err(xg.grad[pick].cpu().numpy(), xd.grad.numpy())
    report("f43_trained_like", shape=[B, C, H, W], forward=e_f, grad_input=e_g,
           forward_elementwise=elem_rel_err(z[pick].detach().cpu().numpy(), z_ref),
           grad_input_elementwise=elem_rel_err(xg.grad[pick].cpu().numpy(), xd.grad.numpy()))
    assert e_f <= TOL and e_g <= TOL, (e_f, e_g)
    # (no inverse here: free taps x 1.5 at 24 channels per group make a 64x64 system lose its digits in the reference's own
    # fp64 solver -- tests/golden/make_golden_r4.py found the same on 16x64 at x 2; the inverse meets trained-like weights in
    # the `heavy` fixtures, at sizes the reference itself can still solve)


# (B, C, H, W): FastFlowUnit shapes whose inverse runs on finc_big.hip's wide-map form of a 33..64-channel bank
WIDE_AFFINE_CASES = [(1, 192, 18, 256), (1, 200, 40, 256)]


@pytest.mark.parametrize("shape", WIDE_AFFINE_CASES, ids=lambda c: "B%d_C%d_%dx%d" % c)
def test_affine_fold_declines_where_the_kernel_cannot_carry_a_shift(shape, dev):
    from fincflow_amd import FastFlowUnit, FlowSequential, _lib, glow
    from fincflow_amd.layers import StandardNormal
    B, C, H, W = shape
    Cq = C // 4
    L = _lib.lib()
    v = _lib.inverse_variant(B, 4, Cq, H, W, 3, 3)
    assert v is not None and v["sec"] == 5, v                        # the wide takeover: form 5
    assert L.finc_inverse_affine_supported(B, 4, Cq, H, W, 3, 3) == 0
    assert L.finc_inverse_affine_supported(B, 4, Cq, 16, 32, 3, 3) == 1   # the same bank on a narrow map carries the shift
    torch.manual_seed(sum(shape))
    unit = FastFlowUnit(C, C, 3)
    with torch.no_grad():
        for m in (unit.conv_tl, unit.conv_tr, unit.conv_bl, unit.conv_br):
            m.conv.weight.mul_(1 - 0.6 * m.get_mask())               # (a wide map amplifies: keep the system well conditioned)
    unit = unit.to(dev)
    an = glow.ActNorm(C).to(dev)
    with torch.no_grad():
        an.log_scale.copy_(0.3 * torch.randn(C, device=dev))
        an.translation.copy_(torch.randn(C, device=dev))
        an.initialized.fill_(1)
    y = torch.randn(B, C, H, W, device=dev)
    with torch.no_grad():
        assert unit.reverse_affine(y, an.log_scale, an.translation) is None
        two = unit.reverse(an.reverse(y))
        seq = FlowSequential(StandardNormal((C, H, W)), unit, an)
        assert seq.fuse_affine
        chain = seq._reverse_chain(y, None)
    ws = torch.cat(unit._weights()).detach().cpu().numpy()
    wco = oracle.canonicalize(ws, 4, ORIENT_FASTFLOW)
    zin = (y * torch.exp(an.log_scale).view(1, -1, 1, 1) + an.translation.view(1, -1, 1, 1)).detach().cpu().numpy()
    want = oracle.inverse_via_f64(zin, wco, 4, ORIENT_FASTFLOW, nthreads=8)
    tol = max(TOL, 2.0 * rel_err(oracle.inverse_f32(zin, wco, 4, ORIENT_FASTFLOW, nthreads=8), want))   # (as test_wide_maps_...)
    assert rel_err(two.cpu().numpy(), want) <= tol
    assert torch.equal(chain, two)                                   # the container took the two-launch path
    # at the C ABI: a bank packed WITH a shift serves the narrow map and is refused -- not computed with -- on the wide one
    from fincflow_amd import ops
    wc = ops.canonicalize(torch.cat(unit._weights()).detach().contiguous(), 4, ORIENT_FASTFLOW)
    packed = torch.empty(L.finc_workspace_bytes(4, Cq, 3, 3), dtype=torch.uint8, device=dev)
    scale, shift = torch.exp(an.log_scale).contiguous(), an.translation.detach().contiguous()
    st = torch.cuda.current_stream().cuda_stream
    assert L.finc_pack_inverse_weights_affine_f32(wc.data_ptr(), scale.data_ptr(), shift.data_ptr(), packed.data_ptr(), 4, Cq, 3, 3, st) == 0
    out = torch.empty_like(y)
    assert L.finc_inverse_packed_f32(y.data_ptr(), packed.data_ptr(), out.data_ptr(), B, 4, Cq, H, W, 3, 3, ORIENT_FASTFLOW, st) == 3
    ys = y[:, :, :16, :32].contiguous()
    outs = torch.empty_like(ys)
    assert L.finc_inverse_packed_f32(ys.data_ptr(), packed.data_ptr(), outs.data_ptr(), B, 4, Cq, 16, 32, 3, 3, ORIENT_FASTFLOW, st) == 0
    want_s = oracle.inverse_via_f64(np.ascontiguousarray(zin[:, :, :16, :32]), wco, 4, ORIENT_FASTFLOW)
    assert rel_err(outs.cpu().numpy(), want_s) <= TOL
    # a scale alone rides on the wide map too; and re-packing the same buffer without a shift revives it
    assert L.finc_pack_inverse_weights_affine_f32(wc.data_ptr(), scale.data_ptr(), None, packed.data_ptr(), 4, Cq, 3, 3, st) == 0
    assert L.finc_inverse_packed_f32(y.data_ptr(), packed.data_ptr(), out.data_ptr(), B, 4, Cq, H, W, 3, 3, ORIENT_FASTFLOW, st) == 0
    want_scale = oracle.inverse_via_f64((y * torch.exp(an.log_scale).view(1, -1, 1, 1)).detach().cpu().numpy(), wco, 4, ORIENT_FASTFLOW, nthreads=8)
    assert rel_err(out.cpu().numpy(), want_scale) <= tol


def test_an_entry_the_training_path_created_is_validated_by_the_first_inference_call(dev):
    from fincflow_amd import FastFlowUnit
    unit = FastFlowUnit(16, 16, 3).to(dev)
    x = torch.randn(2, 16, 8, 8, device=dev, requires_grad=True)
    with torch.no_grad():
        unit.conv_tl.conv.weight[1, 1, -1, -1] = 0.9                # (what weight decay does to the unit diagonal)
    z, _ = unit(x)                                                   # training path: cached, not validated
    with pytest.raises(RuntimeError):
        unit.reverse(z.detach())                                     # same weight version: the check must still run
    with torch.no_grad():
        unit.conv_tl.conv.weight[1, 1, -1, -1] = 1.0
    z, _ = unit(x)
    xr = unit.reverse(z.detach())
    assert rel_err(xr.cpu().numpy(), x.detach().cpu().numpy()) <= 1e-4


def test_fault_gate_is_consulted_by_every_launching_entry_point(dev):
    """after a protocol fault, forward / mix / backward must refuse too -- not only the next inverse.  No fault
    is injected here (tests/test_gpu_variants.py does that with an experiment build): this checks the word is armed by a
    packing call and that a clean device passes every gate."""
    from fincflow_amd import FastFlowUnit, _lib, ops
    unit = FastFlowUnit(48, 48, 3).to(dev)
    x = torch.randn(72, 48, 16, 16, device=dev)
    with torch.no_grad():
        z, _ = unit(x)
        unit.reverse(z)
    torch.cuda.synchronize()
    assert not _lib.fault_pending() and _lib.hlp_timeouts() == 0
    _lib.raise_if_faulted("test")
    m = torch.eye(48, device=dev)
    assert torch.equal(ops.finc_mix(x, m), x)


# (B, C, H, W, K): problem sets the band split takes (2 * B * 4 workgroups <= compute units, >= 2 bands, W >= 64): 2 .. 8 bands,
# partial last bands, an odd number of bands, a padded bank (Cq = 22), 2x2 taps, widths beyond 64
# (round 5: the banks of up to 12 channels left the band split for the short-step kernel, finc_chain.hip -- C = 48 became C = 80)
BAND_SPLIT_CASES = [(32, 96, 64, 64, 3), (4, 96, 64, 64, 3), (3, 96, 48, 64, 3), (2, 96, 33, 64, 3), (8, 80, 17, 72, 3),
                    (1, 128, 40, 72, 3), (7, 88, 31, 64, 3), (5, 64, 128, 64, 3), (6, 64, 50, 64, 2),
                    # W = 68: 17 store windows per row -- the producer's loop ends before the stores of its last window are said
                    # complete (scripts/stress_bands.py found the consumer of a workgroup's last band waiting for them forever)
                    (3, 80, 26, 68, 3), (5, 64, 85, 68, 2), (18, 96, 82, 68, 3)]


@pytest.mark.parametrize("shape", BAND_SPLIT_CASES, ids=lambda c: "B%d_C%d_%dx%d_k%d" % c)
def test_band_split_of_the_role_split_inverse(shape, dev):
    """finc_split.hip, BSP: the bands of a problem dealt out to two workgroups on different compute units, the rows above a band
    handed over through memory (progress words, write-through stores, loads past the caches).  Same visitation inside a band
    (cinc_cuda_kernel_level2.cu:49-56) -- against the oracle's fp64 path, against the chained form (FINC_SPLIT_BANDS=0 is another
    process: here the strict kernel stands in as the independent answer), repeated launches bit-identical (the epoch of the
    progress words advances on the device), no wait gave up."""
    from fincflow_amd import FastFlowUnit, _lib, ops
    B, C, H, W, K = shape
    v = _lib.inverse_variant(B, 4, C // 4, H, W, K, K)
    assert v is not None and v["sec"] == 4 and v["workgroups"] == ((H + 15) // 16) * B * 4, v   # (round 5: one workgroup per band)
    torch.manual_seed(sum(shape))
    unit = FastFlowUnit(C, C, K).to(dev)
    x = torch.randn(B, C, H, W, device=dev)
    with torch.no_grad():
        z, _ = unit(x)
        got = unit.reverse(z)
        strict = ops.finc_inverse(z, unit._cache.w_canon, algo="strict")
        again = [unit.reverse(z) for _ in range(4)]
    torch.cuda.synchronize()
    ws = torch.cat(unit._weights()).detach().cpu().numpy()
    wco = oracle.canonicalize(ws, 4, ORIENT_FASTFLOW)
    pick = sorted({0, B // 2, B - 1})
    want = oracle.inverse_via_f64(z[pick].cpu().numpy(), wco, 4, ORIENT_FASTFLOW, nthreads=8)
    e = rel_err(got[pick].cpu().numpy(), want)
    report("band_split", shape=list(shape), err_max_norm=e, err_elementwise=elem_rel_err(got[pick].cpu().numpy(), want))
    assert e <= TOL, e
    assert rel_err(got.cpu().numpy(), strict.cpu().numpy()) <= TOL
    assert all(torch.equal(a, got) for a in again)
    assert _lib.hlp_timeouts() == 0 and not _lib.fault_pending()


def test_band_split_in_a_captured_graph_and_with_a_folded_affine_map(dev):
    """A captured launch keeps its slot of progress words; the epoch lives on the device, so every replay starts clean.  The
    affine fold rides on the band split like on the chained form (same packed bank)."""
    from fincflow_amd import FastFlowUnit, _lib, glow
    B, C, H, W = 8, 96, 64, 64
    assert _lib.inverse_variant(B, 4, C // 4, H, W, 3, 3)["workgroups"] == (H // 16) * B * 4
    torch.manual_seed(5)
    unit = FastFlowUnit(C, C, 3).to(dev)
    an = glow.ActNorm(C).to(dev)
    with torch.no_grad():
        an.log_scale.copy_(0.2 * torch.randn(C, device=dev))
        an.translation.copy_(torch.randn(C, device=dev))
        an.initialized.fill_(1)
    y = torch.randn(B, C, H, W, device=dev)
    with torch.no_grad():
        two = unit.reverse(an.reverse(y))
        fused = unit.reverse_affine(y, an.log_scale, an.translation)
        assert fused is not None and rel_err(fused.cpu().numpy(), two.cpu().numpy()) <= TOL
        ref = unit.reverse(y)
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


# (B, C, H, W): every bank of finc_wino5.hip (Cq = 4 .. 48, padded channel counts among them: Cq = 20 on the 24-channel two-wave
# bank, 40 on the 48-channel one), one and several strips of 32 columns, a partial last strip, maps shorter than the four-row
# prologue, row chunks (few strips), the c5 bank at its per-GPU width
FORWARD_55_CASES = [(1, 16, 12, 20), (2, 32, 9, 32), (2, 48, 3, 34), (3, 64, 17, 64), (2, 80, 20, 24), (2, 96, 33, 40), (1, 128, 16, 96),
                    (2, 160, 7, 30), (2, 192, 24, 128), (1, 192, 40, 36)]


@pytest.mark.parametrize("shape", FORWARD_55_CASES, ids=lambda c: "B%d_C%d_%dx%d" % c)
def test_5x5_forward_with_fewer_multiplies(shape, dev):
    """finc_wino5.hip: Winograd F(2,5) along W for the 5x5 banks (0.6 x the MFMAs of the direct sum; layers/conv.py:102-107 is free
    to run any exact reformulation) -- forward, grad-input (the same kernel on transposed fragments) and the output-side affine fold
    against fp64 F.pad + F.conv2d autograd on the CPU, all four corner orientations of a FastFlowUnit; the direct strip kernels
    (finc_debug_set_forward_form(1)) beside it on the same data."""
    import torch.nn.functional as F
    from fincflow_amd import FastFlowUnit, _lib
    B, C, H, W = shape
    torch.manual_seed(sum(shape))
    unit = FastFlowUnit(C, C, 5)
    with torch.no_grad():
        for m in (unit.conv_tl, unit.conv_tr, unit.conv_bl, unit.conv_br):
            m.conv.weight.mul_(1 - 0.6 * m.get_mask())               # N(0, 0.02^2) free taps, as bench.py's c5
    unit = unit.to(dev)
    x = torch.randn(B, C, H, W, device=dev)
    gz = torch.randn(B, C, H, W, device=dev)
    xd = x.detach().cpu().double().requires_grad_(True)
    ref = torch.cat([F.conv2d(F.pad(c, m.pad), m.conv.weight.detach().cpu().double()) for m, c in
                     zip((unit.conv_tl, unit.conv_tr, unit.conv_bl, unit.conv_br), torch.chunk(xd, 4, 1))], 1)
    ref.backward(gz.cpu().double())
    log_scale, translation = 0.3 * torch.randn(C, device=dev), torch.randn(C, device=dev)
    ref_aff = (ref.detach() - translation.cpu().double().view(1, -1, 1, 1)) * torch.exp(-log_scale).cpu().double().view(1, -1, 1, 1)
    worst = {}
    try:
        for form, want in ((0, "winograd25"), (1, ("strip", "strip16"))):
            _lib.set_forward_form(form)
            got_form = _lib.backward_variant(B, 4, C // 4, H, W, 5, 5)["conv_form"]
            assert got_form == want or got_form in want, (form, got_form)
            xg = x.clone().requires_grad_(True)
            z, logdet = unit(xg)
            z.backward(gz)
            assert logdet == 0.0
            with torch.no_grad():
                fused = unit.forward_affine(x, log_scale, translation)
            assert fused is not None
            e = (rel_err(z.detach().cpu().numpy(), ref.detach().numpy()), rel_err(xg.grad.cpu().numpy(), xd.grad.numpy()),
                 rel_err(fused.cpu().numpy(), ref_aff.numpy()))
            worst[form] = max(e)
            assert worst[form] <= TOL, (form, e)
    finally:
        _lib.set_forward_form(0)
    report("forward_5x5_forms", shape=list(shape), f25=worst[0], strip=worst[1])


def test_5x5_forward_on_the_reference_fixture(dev):
    """unit_B1_C16_12x20_k5 (reference-generated, tests/golden/make_golden.py): the F(2,5) kernel's forward against the reference's z."""
    from fincflow_amd import _lib, ops
    g = golden("unit_B1_C16_12x20_k5")
    assert _lib.backward_variant(1, 4, 4, 12, 20, 5, 5)["conv_form"] == "winograd25"
    wc = ops.canonicalize(t(unit_stored_weights(g), dev), 4, ORIENT_FASTFLOW)
    z = ops.finc_forward(t(g["x"], dev), wc, 4, ORIENT_FASTFLOW).cpu().numpy()
    assert rel_err(z, g["z"]) <= TOL


# (B, C, H, W, (KH, KW)): non-square filters (PaddedConv2d takes a tuple, layers/conv.py:30-36; the reference's fixtures have 3x5
# and 2x3) on the MFMA kernels -- every bank of the two shapes, padded channel counts, widths the inverse streams and widths it pads
NONSQUARE_CASES = [(2, 16, 12, 16, (2, 3)), (3, 20, 9, 9, (2, 3)), (2, 32, 20, 24, (2, 3)), (1, 64, 33, 32, (2, 3)), (70, 32, 8, 16, (2, 3)),
                   (2, 8, 10, 14, (3, 5)), (2, 32, 18, 20, (3, 5)), (1, 64, 24, 40, (3, 5)), (2, 52, 7, 12, (3, 5)), (66, 64, 9, 16, (3, 5))]


@pytest.mark.parametrize("case", NONSQUARE_CASES, ids=lambda c: "B%d_C%d_%dx%d_k%dx%d" % (c[0], c[1], c[2], c[3], c[4][0], c[4][1]))
def test_non_square_filters_run_on_mfma(case, dev):
    """VERDICT r3 missing 5: K_H != K_W ran the scalar kernels (100x slower).  Inverse against the oracle's fp64 path (the strict
    kernel beside it), forward and its gradients against the oracle / CPU autograd; the library's answer about the kernel asserted."""
    import torch.nn.functional as F
    from fincflow_amd import FastFlowUnit, _lib, ops
    B, C, H, W, (KH, KW) = case
    L = _lib.lib()
    Cq = C // 4
    assert L.finc_forward_algo_for(Cq, H, W, KH, KW) == _lib.ALGO["mfma"]
    Wp = W if W % 4 == 0 else (W + 7) // 8 * 8
    assert L.finc_inverse_algo_for(Cq, H, Wp, KH, KW) == _lib.ALGO["mfma"]
    torch.manual_seed(sum(case[:4]) + KH)
    unit = FastFlowUnit(C, C, (KH, KW)).to(dev)
    x = torch.randn(B, C, H, W, device=dev)
    gz = torch.randn(B, C, H, W, device=dev)
    xg = x.clone().requires_grad_(True)
    z, logdet = unit(xg)
    z.backward(gz)
    assert logdet == 0.0
    with torch.no_grad():
        xr = unit.reverse(z.detach())
        xs = ops.finc_inverse(z.detach(), unit._cache.w_canon, algo="strict")
    ws = torch.cat(unit._weights()).detach().cpu().numpy()
    wco = oracle.canonicalize(ws, 4, ORIENT_FASTFLOW)
    pick = sorted({0, B // 2, B - 1})
    z_ref = oracle.forward_f32(x[pick].cpu().numpy(), wco, 4, ORIENT_FASTFLOW, accumulate_f64=True)
    x_ref = oracle.inverse_via_f64(z[pick].detach().cpu().numpy(), wco, 4, ORIENT_FASTFLOW)
    assert rel_err(z[pick].detach().cpu().numpy(), z_ref) <= TOL
    assert rel_err(xr[pick].cpu().numpy(), x_ref) <= TOL
    assert rel_err(xr.cpu().numpy(), xs.cpu().numpy()) <= TOL
    xd = x[pick].cpu().double().requires_grad_(True)
    wds = [m.conv.weight.detach().cpu().double().requires_grad_(True) for m in (unit.conv_tl, unit.conv_tr, unit.conv_bl, unit.conv_br)]
    ref = torch.cat([F.conv2d(F.pad(c, m.pad), w) for m, c, w in
                     zip((unit.conv_tl, unit.conv_tr, unit.conv_bl, unit.conv_br), torch.chunk(xd, 4, 1), wds)], 1)
    ref.backward(gz[pick].cpu().double())
    assert rel_err(xg.grad[pick].cpu().numpy(), xd.grad.numpy()) <= TOL
    if len(pick) == B:                                                # (the weight gradient sums over the batch)
        for m, w in zip((unit.conv_tl, unit.conv_tr, unit.conv_bl, unit.conv_br), wds):
            want = (w.grad * m.get_mask().cpu().double()).numpy()
            assert rel_err(m.conv.weight.grad.cpu().numpy(), want) <= 2e-5


# ------------------------------------------------------------------ Winograd weight gradients (finc_gradw.hip)
ONE_GROUP_GRADW_CASES = [
    # (B, Cq, H, W, K, order, grad-weight kernel): a single PaddedConv2d (G = 1: its waves spread over the whole chip, not a
    # quarter of it), every storage order -- the mirrored strips and the row walk from the bottom
    (3, 24, 9, 64, 3, "TL", "winograd"), (2, 24, 7, 36, 3, "TR", "winograd"), (2, 20, 5, 32, 3, "BL", "winograd"),
    (2, 32, 6, 48, 3, "BR", "winograd"), (2, 16, 4, 16, 3, "TR", "winograd"),
    (2, 48, 5, 40, 3, "BR", "winograd_tiled"), (1, 80, 4, 32, 3, "TR", "winograd_tiled"),
    (2, 48, 6, 24, 5, "TR", "winograd_tiled"), (2, 16, 5, 20, 5, "BL", "winograd_tiled"), (1, 40, 3, 16, 5, "BR", "winograd_tiled"),
]


@pytest.mark.parametrize("case", ONE_GROUP_GRADW_CASES, ids=lambda c: "B%d_Cq%d_%dx%d_k%d_%s_%s" % c)
def test_winograd_weight_gradient_of_one_padded_conv(case, dev):
    """The transposed-Winograd grad-weight kernels with G = 1 (`PaddedConv2d`, layers/conv.py:30-107), entry by entry against
    CPU fp64 autograd through F.pad + F.conv2d times the gradient mask (layers/conv.py:98-99)."""
    import torch.nn.functional as F
    from fincflow_amd import PaddedConv2d, _lib
    B, Cq, H, W, K, order, want = case
    assert _lib.backward_variant(B, 1, Cq, H, W, K, K)["gradw"] == want
    torch.manual_seed(sum(case[:5]))
    m = PaddedConv2d(Cq, Cq, (K, K), order=order).to(dev)
    x = torch.randn(B, Cq, H, W, device=dev, requires_grad=True)
    z, _ = m(x)
    gz = torch.randn_like(z)
    z.backward(gz)
    xc = x.detach().cpu().double().requires_grad_(True)
    w = m.conv.weight.detach().cpu().double().requires_grad_(True)
    F.conv2d(F.pad(xc, m.pad), w).backward(gz.cpu().double())
    expect = (w.grad * m.mask.double()).numpy()
    got = m.conv.weight.grad.cpu().numpy()
    ew = rel_err(got, expect)
    report("grad_w_" + want, case="B%d_Cq%d_%dx%d_k%d_%s" % case[:6], max_normalised=ew)
    assert ew <= 1e-5, ew
    assert np.all(got[m.mask.numpy() == 0] == 0)
    assert rel_err(x.grad.cpu().numpy(), xc.grad.numpy()) <= TOL


# (B, C, H, W): the banks of finc_wino4m.hip (32, 48, 64 channels per group; 28, 40 and 52 padded onto them), strips of 64 columns
# whole and three quarters full, a second strip, maps shorter than the prologue, row chunks (few strips) and none
FORWARD_4M_CASES = [(8, 128, 12, 64), (8, 112, 9, 48), (8, 192, 20, 64), (9, 160, 3, 64), (4, 208, 17, 128), (8, 256, 24, 64),
                    (40, 192, 8, 64), (2, 192, 40, 112)]


@pytest.mark.parametrize("shape", FORWARD_4M_CASES, ids=lambda c: "B%d_C%d_%dx%d" % c)
def test_3x3_forward_of_the_wide_banks_with_fewer_multiplies(shape, dev):
    """finc_wino4m.hip: Winograd F(4,3) along W, M-split over the waves of a workgroup, for 3x3 banks of 28 .. 64 channels per group
    (layers/conv.py:102-107 is free to run any exact reformulation) -- forward, grad-input (the same kernel on transposed
    fragments) and the output-side affine fold against fp64 F.pad + F.conv2d autograd on the CPU, all four corner orientations;
    the direct strip kernels (finc_debug_set_forward_form(1)) beside it on the same data."""
    import torch.nn.functional as F
    from fincflow_amd import FastFlowUnit, _lib
    B, C, H, W = shape
    torch.manual_seed(sum(shape))
    unit = FastFlowUnit(C, C, 3).to(dev)
    x = torch.randn(B, C, H, W, device=dev)
    gz = torch.randn(B, C, H, W, device=dev)
    xd = x.detach().cpu().double().requires_grad_(True)
    ref = torch.cat([F.conv2d(F.pad(c, m.pad), m.conv.weight.detach().cpu().double()) for m, c in
                     zip((unit.conv_tl, unit.conv_tr, unit.conv_bl, unit.conv_br), torch.chunk(xd, 4, 1))], 1)
    ref.backward(gz.cpu().double())
    log_scale, translation = 0.3 * torch.randn(C, device=dev), torch.randn(C, device=dev)
    ref_aff = (ref.detach() - translation.cpu().double().view(1, -1, 1, 1)) * torch.exp(-log_scale).cpu().double().view(1, -1, 1, 1)
    worst = {}
    try:
        for form, want in ((0, ("winograd4m",)), (1, ("strip", "strip16"))):
            _lib.set_forward_form(form)
            got_form = _lib.backward_variant(B, 4, C // 4, H, W, 3, 3)["conv_form"]
            assert got_form in want, (form, got_form)
            xg = x.clone().requires_grad_(True)
            z, logdet = unit(xg)
            z.backward(gz)
            assert logdet == 0.0
            with torch.no_grad():
                fused = unit.forward_affine(x, log_scale, translation)
            assert fused is not None
            e = (rel_err(z.detach().cpu().numpy(), ref.detach().numpy()), rel_err(xg.grad.cpu().numpy(), xd.grad.numpy()),
                 rel_err(fused.cpu().numpy(), ref_aff.numpy()))
            worst[form] = max(e)
            assert worst[form] <= TOL, (form, e)
    finally:
        _lib.set_forward_form(0)
    report("forward_3x3_wide_bank_forms", shape=list(shape), f43_msplit=worst[0], strip=worst[1])


@pytest.mark.parametrize("C", [48, 64, 40])
def test_cinc_unit_forward_on_the_msplit_winograd_kernel(C, dev):
    """`CINCFlowUnit` (cinc_flow.py:9-80: one TL conv over all channels, G = 1) at 40 .. 64 channels: forward and backward through
    finc_wino4m.hip / the tile-pair Winograd grad-weight against CPU fp64 autograd, and the inverse round trip."""
    import torch.nn.functional as F
    from fincflow_amd import CINCFlowUnit, _lib
    B, H, W = 40, 16, 64
    v = _lib.backward_variant(B, 1, C, H, W, 3, 3)
    assert v["conv_form"] == "winograd4m" and v["gradw"] == "winograd_tiled", v
    torch.manual_seed(C)
    u = CINCFlowUnit(C, C, 3).to(dev)
    x = torch.randn(B, C, H, W, device=dev, requires_grad=True)
    z, ld = u(x)
    gz = torch.randn_like(z)
    z.backward(gz)
    m = u.conv_tl
    xc = x.detach().cpu().double().requires_grad_(True)
    w = m.conv.weight.detach().cpu().double().requires_grad_(True)
    zc = F.conv2d(F.pad(xc, m.pad), w)
    zc.backward(gz.cpu().double())
    assert ld == 0.0 and rel_err(z.detach().cpu().numpy(), zc.detach().numpy()) <= TOL
    assert rel_err(x.grad.cpu().numpy(), xc.grad.numpy()) <= TOL
    assert rel_err(m.conv.weight.grad.cpu().numpy(), (w.grad * m.mask.double()).numpy()) <= 1e-5
    if C <= 48:     # (64 channels at the init's N(0, 0.05^2) on a 64-wide map: the triangular system amplifies fp32 rounding past any
                    #  tolerance -- in the oracle's fp32 order as well; the forward / backward kernels are what this test is about)
        with torch.no_grad():
            assert rel_err(u.reverse(z.detach()).cpu().numpy(), x.detach().cpu().numpy()) <= TOL
