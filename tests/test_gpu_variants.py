"""Every compiled variant of the MFMA inverse, configs[4] at its per-GPU batch, a bounded random sweep, the fp64 entry
points, the biased layer and the unaligned-view fallback -- all through the C ABI, against the oracle.

Each case appends its achieved errors (max-normalised AND element-wise) to gpurun_out/parity_report.jsonl so the
cases that needed more than 1e-5 are on record (summary committed under profiles/).
Recurrence under test: cinc_cuda_kernel_level2.cu:59-72.
"""
import numpy as np
import pytest
import torch

from oracle import oracle
from helpers import (ORIENT_FASTFLOW, elem_rel_err, fuzz_case, problem_counts_for_row, rel_err, report, split_problems)

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


def _rows():
    from fincflow_amd import _lib
    return _lib.inverse_table()


def bank_std(Cq, K):
    return (0.05 if K < 5 else 0.02) * min(1.0, (24.0 / Cq) ** 0.5)


def run_inverse_case(dev, B, G, orient, Cq, H, W, KH, KW, seed, tag, expect_row=None, expect_sec=None):
    """One problem set through auto (MFMA) and strict; returns the achieved errors."""
    from fincflow_amd import _lib, ops
    K = max(KH, KW)
    v = _lib.inverse_variant(B, G, Cq, H, W, KH, KW)
    if expect_row is not None:
        assert v is not None and v["row"] == expect_row, (v, expect_row)
        # (form 3 = form 2 with helper waves: taken when the problems come in fours and the bank leaves room)
        assert v["sec"] == expect_sec or (expect_sec == 2 and v["sec"] == 3 and (B * G) % 4 == 0)
    rng = np.random.default_rng(seed)
    ws = oracle.make_stored_weights(G, Cq, KH, KW, orient=orient, seed=seed, std=bank_std(Cq, K))
    wco = oracle.canonicalize(ws, G, orient)
    x = rng.standard_normal((B, G * Cq, H, W)).astype(np.float32)
    nthr = min(oracle.max_threads(), 16)
    z = oracle.forward_f32(x, wco, G, orient, nthreads=nthr)
    ref = oracle.inverse_via_f64(z, wco, G, orient, nthreads=nthr)
    ref32 = oracle.inverse_f32(z, wco, G, orient, nthreads=nthr)
    wc = ops.canonicalize(t(ws, dev), G, orient)
    zt = t(z, dev)
    auto = ops.finc_inverse(zt, wc, G, orient, algo="auto").cpu().numpy()
    strict = ops.finc_inverse(zt, wc, G, orient, algo="strict").cpu().numpy()
    gap = rel_err(ref32, ref)                       # the reference's own fp32-order vs fp64 difference on this bank
    tol = max(TOL, 2.0 * gap)
    e_max, e_elem = rel_err(auto, ref), elem_rel_err(auto, ref)
    gap_elem = elem_rel_err(ref32, ref)             # ... and the same difference judged entry by entry
    report(tag, B=B, G=G, Cq=Cq, H=H, W=W, K=[KH, KW], orient=orient, variant=v, err_max_norm=e_max, err_elementwise=e_elem,
           reference_fp32_vs_fp64=gap, reference_fp32_vs_fp64_elementwise=gap_elem, tol=tol, needed_more_than_1e5=bool(e_max > TOL))
    assert np.array_equal(strict, ref32), "strict kernel must be bit-exact with the fp32 reference order"
    assert e_max <= tol, (e_max, tol, v)
    # element-wise (every entry of at least a thousandth of the largest judged against ITSELF): 1e-3 holds on every case on record
    # (worst 5.4e-4, profiles/r04/parity_errors.json) -- scaled like `tol` for the banks whose own fp32-fp64 gap exceeds 1e-5
    # -- or, on the widest banks (256 channels: 1.2e-3 measured), within twice what the reference's own fp32 order shows entry by entry
    assert e_elem <= max(1e-3 * (tol / TOL), 2.0 * gap_elem), (e_elem, gap_elem, tol, v)
    return e_max, e_elem


@pytest.mark.parametrize("row", range(34))
def test_every_row_of_the_instantiation_table(row, dev):
    """Walks g_insts (finc_mfma.hip): each row is launched in its 64-byte sector-pairing form (W % 16 == 0, one-wave rows),
    its 32-byte-I/O form (W % 8 == 0) and its 16-byte form (W % 8 == 4), at problem counts on each side of max_problems and at odd and even counts (problems per workgroup),
    with the full channel count and with padded channels; the test asserts WHICH variant the library picked."""
    rows = _rows()
    if row >= len(rows):
        pytest.skip("table has fewer rows")
    assert len(rows) <= 34, "extend the parametrisation: the table grew"
    i = rows[row]
    counts = problem_counts_for_row(rows, row)
    assert counts, f"no problem count selects row {row}: {i}"
    one_wave = i["nw"] == 1 and i["npw"] == 1
    for n in counts:
        B, G, orient = split_problems(n)
        big = n > 64
        # I/O form by width: W % 16 == 0 -> 64-byte sector pairing where the row has it (one wave per problem): with helper
        # waves (3) when the problems come in fours, without (2) otherwise; else W % 8 == 0 -> 32-byte pieces (1), else
        # 16-byte groups (0)
        s64 = 2 if one_wave else 1
        shapes = ((s64, (7, 16) if big else (10, 32)), (1, (5, 24) if big else (19, 24)),
                  (0, (5, 12) if big else (9, 20)))
        for sec, (H, W) in shapes:
            Cq = i["cqp"] if sec else max(i["cqp"] - 1, 1)          # the 16-byte form also carries padded channels
            run_inverse_case(dev, B, G, orient, Cq, H, W, i["kh"], i["kw"], seed=1000 * row + n + sec,
                             tag="variant_table", expect_row=row, expect_sec=sec)


SPLIT_CASES = [
    # (B, G, Cq, H, W, KH, KW): the bench shapes first -- c3's per-GPU share of an 8- and a 4-way split, c2, the c4 units
    (32, 4, 24, 64, 64, 3, 3), (64, 4, 24, 64, 64, 3, 3), (64, 4, 12, 32, 32, 3, 3), (32, 4, 3, 16, 16, 3, 3), (16, 4, 6, 8, 8, 3, 3),
    (8, 4, 12, 4, 4, 3, 3),
    # every bank of the kernel, padded channels (Cq not a multiple of 4), ragged heights (partial last band), widths on both
    # sides of one band of 16 columns, a single row, the widest map the FIFO block holds (72), all four orientations (G = 4)
    (3, 4, 1, 9, 12, 3, 3), (2, 4, 4, 20, 20, 3, 3), (2, 4, 7, 17, 28, 3, 3), (5, 1, 8, 33, 16, 3, 3), (2, 4, 11, 5, 40, 3, 3),
    (1, 4, 16, 40, 24, 3, 3), (3, 1, 19, 21, 36, 3, 3), (2, 4, 20, 1, 32, 3, 3), (1, 4, 23, 35, 72, 3, 3), (2, 2, 28, 18, 20, 3, 3),
    (1, 4, 27, 30, 8, 3, 3), (2, 4, 32, 19, 44, 3, 3), (1, 3, 30, 50, 4, 3, 3),
    (4, 4, 2, 6, 8, 2, 2), (2, 4, 8, 23, 32, 2, 2), (3, 1, 12, 16, 16, 2, 2), (2, 4, 15, 37, 20, 2, 2), (1, 4, 24, 12, 64, 2, 2),
    (2, 4, 31, 9, 28, 2, 2),
]


@pytest.mark.parametrize("case", SPLIT_CASES, ids=lambda c: "B%d_G%d_Cq%d_%dx%d_k%dx%d" % c)
def test_role_split_kernel(case, dev):
    """finc_split.hip (the kernel of the under-filled chip: one wave carries the recurrence, three prepare everything else
    one step ahead) against the oracle and, bit for bit, over repeated launches.  Recurrence: cinc_cuda_kernel_level2.cu:59-72."""
    from fincflow_amd import _lib, ops
    B, G, Cq, H, W, KH, KW = case
    orient = ORIENT_FASTFLOW if G == 4 else (0x1B & ((1 << (2 * G)) - 1))
    v = _lib.inverse_variant(B, G, Cq, H, W, KH, KW)
    # (round 4: with compute units to spare -- 2 B G <= 256 -- on a map of >= 2 bands and >= 64 columns the bands of a problem are
    # dealt out to two workgroups: tests/test_gpu_round4.py walks that form)
    # (round 5: the banks of up to 16 channels run its short-step form, finc_chain.hip: form 6 -- the recurrence wave, one wave per
    # tap with a + b == 2, the I/O wave; one workgroup per problem)
    chain = Cq <= 16 and not (Cq > 12 and 2 * B * G <= 256 and H > 16 and W >= 64 and KH > 1)
    per = (H + 15) // 16 if (2 * B * G <= 256 and H > 16 and W >= 64 and KH > 1 and not chain) else 1   # (round 5: one workgroup per band)
    assert v is not None and v["sec"] == (6 if chain else 4) and v["nw"] == ((5 if KH == 3 else 3) if chain else 4), v
    assert v["workgroups"] == per * B * G, v
    e_max, _ = run_inverse_case(dev, B, G, orient, Cq, H, W, KH, KW, seed=31 * Cq + H + W, tag="role_split")
    assert e_max <= TOL or Cq > 24                                          # (run_inverse_case holds the wider banks to 2x the reference's own fp32-fp64 gap)
    rng = np.random.default_rng(9)
    ws = oracle.make_stored_weights(G, Cq, KH, KW, orient=orient, seed=4, std=bank_std(Cq, max(KH, KW)))
    wc = ops.canonicalize(t(ws, dev), G, orient)
    z = t(rng.standard_normal((B, G * Cq, H, W)).astype(np.float32), dev)
    first = ops.finc_inverse(z, wc, G, orient)
    for _ in range(10):
        assert torch.equal(ops.finc_inverse(z, wc, G, orient), first)


def test_role_split_kernel_with_the_affine_fold(dev):
    """The folded affine map (z = s*y + t in front of the inverse, layers/actnorm.py:39-52) rides in the bank of
    the role-split kernel as in the wavefront kernel's: Linv*diag(s) as the z-term, Linv*t as the start of the accumulators."""
    from fincflow_amd import FastFlowUnit, _lib
    torch.manual_seed(3)
    for (B, C, H, W) in ((8, 96, 24, 32), (16, 48, 32, 32), (4, 12, 16, 16)):
        unit = FastFlowUnit(C, C, 3).to(dev)
        assert _lib.inverse_variant(B, 4, C // 4, H, W, 3, 3)["sec"] == (4 if C // 4 > 16 else 6)
        y = torch.randn(B, C, H, W, device=dev)
        log_scale = 0.2 * torch.randn(C, device=dev)
        translation = torch.randn(C, device=dev)
        fused = unit.reverse_affine(y, log_scale, translation)
        assert fused is not None
        plain = unit.reverse(torch.exp(log_scale).view(1, -1, 1, 1) * y + translation.view(1, -1, 1, 1))
        assert rel_err(fused.cpu().numpy(), plain.cpu().numpy()) <= TOL


def test_wide_map_at_a_full_problem_count(dev):
    """A map wider than 64 columns with more problems than the small-batch rows take: the packed two-wave form (find_inst's
    LDS rule), against the oracle."""
    e_max, _ = run_inverse_case(dev, 130, 4, ORIENT_FASTFLOW, 24, 5, 80, 3, 3, seed=4242, tag="wide_map")
    from fincflow_amd import _lib
    v = _lib.inverse_variant(130, 4, 24, 5, 80, 3, 3)
    assert (v["nw"], v["npw"]) == (2, 2)
    assert e_max <= 1e-5


def test_c5_at_the_per_gpu_batch(dev):
    """BASELINE configs[4]: 5x5, C=192, 128x128 at the per-GPU batch of 64 (512 over 8 GPUs).  Forward and inverse of the
    whole batch on the K-split MFMA kernels; images 0, 31 and 63 against the oracle (fp64 solve / fp32 forward), the
    whole batch through the round trip and the sampling residual.  Weights: std 0.02 (at the init std 0.05
    the 5x5/Cq=48 inverse itself is unstable -- tests/test_oracle.py pins that on the CPU)."""
    from fincflow_amd import _lib, ops
    B, C, H, W, K = 64, 192, 128, 128, 5
    v = _lib.inverse_variant(B, 4, C // 4, H, W, K, K)
    assert v is not None and v["nw"] == 4 and v["cqp"] == 48
    ws = oracle.make_stored_weights(4, C // 4, K, K, std=0.02)
    wco = oracle.canonicalize(ws, 4, ORIENT_FASTFLOW)
    wc = ops.canonicalize(t(ws, dev), 4, ORIENT_FASTFLOW)
    torch.manual_seed(11)
    x = torch.randn(B, C, H, W, device=dev)
    z = ops.finc_forward(x, wc)
    xr = ops.finc_inverse(z, wc)
    e_rt = float((xr - x).abs().max() / x.abs().max())
    assert e_rt <= TOL, e_rt
    pick = [0, 31, 63]
    nthr = min(oracle.max_threads(), 16)
    zp = z[pick].cpu().numpy()
    ref_x = oracle.inverse_via_f64(zp, wco, nthreads=nthr)
    ref_z = oracle.forward_f32(x[pick].cpu().numpy(), wco, nthreads=nthr)
    e_inv, e_fwd = rel_err(xr[pick].cpu().numpy(), ref_x), rel_err(zp, ref_z)
    report("c5_full_batch", B=B, variant=v, round_trip=e_rt, inverse_vs_oracle=e_inv, forward_vs_oracle=e_fwd,
           inverse_elementwise=elem_rel_err(xr[pick].cpu().numpy(), ref_x))
    assert e_inv <= TOL and e_fwd <= TOL, (e_inv, e_fwd)
    del xr
    zs = torch.randn(B, C, H, W, device=dev)                      # the sampling distribution
    xs = ops.finc_inverse(zs, wc)
    resid = float((ops.finc_forward(xs, wc) - zs).abs().max() / zs.abs().max())
    ref_s = oracle.inverse_via_f64(zs[[5]].cpu().numpy(), wco, nthreads=nthr)
    e_s = rel_err(xs[[5]].cpu().numpy(), ref_s)
    report("c5_full_batch_sampling", residual=resid, inverse_vs_oracle=e_s, xmax=float(xs.abs().max()))
    assert resid <= TOL and e_s <= TOL, (resid, e_s)


@pytest.mark.parametrize("seed", [0, 1, 2])
def test_bounded_fuzz_sweep(seed, dev):
    """scripts/fuzz_parity.py's generator, 40 cases per seed: random (B, G, Cq, H, W, K, orientation) through inverse
    auto/strict and the forward, against the oracle."""
    from fincflow_amd import ops, _lib
    rng = np.random.default_rng(seed)
    worst = 0.0
    for case in range(40):
        c = fuzz_case(rng, case)
        B, G, Cq, H, W, K, orient = c["B"], c["G"], c["Cq"], c["H"], c["W"], c["K"], c["orient"]
        ws = oracle.make_stored_weights(G, Cq, K, K, orient=orient, seed=case, std=c["std"])
        wco = oracle.canonicalize(ws, G, orient)
        x = rng.standard_normal((B, G * Cq, H, W)).astype(np.float32)
        z = oracle.forward_f32(x, wco, G, orient)
        ref = oracle.inverse_via_f64(z, wco, G, orient)
        ref32 = oracle.inverse_f32(z, wco, G, orient)
        wc = ops.canonicalize(t(ws, dev), G, orient)
        auto = ops.finc_inverse(t(z, dev), wc, G, orient, algo="auto").cpu().numpy()
        strict = ops.finc_inverse(t(z, dev), wc, G, orient, algo="strict").cpu().numpy()
        fwd = ops.finc_forward(t(x, dev), wc, G, orient).cpu().numpy()
        gap = rel_err(ref32, ref)
        tol = max(TOL, 2.0 * gap)
        e_inv, e_fwd = rel_err(auto, ref), rel_err(fwd, z)
        if _lib.lib().finc_inverse_premultiplied_supported(B, G, Cq, H, W, K, K):   # the form without the z-term, where it exists
            lead = np.linalg.inv(wco.reshape(G, Cq, Cq, K, K)[:, :, :, -1, -1].astype(np.float64))
            zp = np.einsum("gok,bgkhw->bgohw", lead, z.reshape(B, G, Cq, H, W).astype(np.float64)).astype(np.float32).reshape(z.shape)
            packed = torch.empty(_lib.lib().finc_workspace_bytes(G, Cq, K, K), dtype=torch.uint8, device=dev)
            _lib.check(_lib.lib().finc_pack_inverse_weights_f32(wc.data_ptr(), packed.data_ptr(), G, Cq, K, K, None), "pack")
            zpt, xo = t(zp, dev), torch.empty(z.shape, dtype=torch.float32, device=dev)
            _lib.check(_lib.lib().finc_inverse_packed_premultiplied_f32(zpt.data_ptr(), packed.data_ptr(), xo.data_ptr(), B, G, Cq,
                                                                         H, W, K, K, orient, None), "premultiplied")
            torch.cuda.synchronize()
            assert rel_err(xo.cpu().numpy(), ref) <= tol, (c, "premultiplied")
        report("fuzz", seed=seed, **c, err_max_norm=e_inv, err_elementwise=elem_rel_err(auto, ref), forward=e_fwd,
               reference_fp32_vs_fp64=gap, needed_more_than_1e5=bool(e_inv > TOL))
        assert np.array_equal(strict, ref32), c
        assert e_inv <= tol and e_fwd <= TOL, (c, e_inv, e_fwd, tol)
        worst = max(worst, e_inv)
    assert worst < 1e-3


# ------------------------------------------------------------------ forward: the staged (16-byte piece) form
STAGED_FORWARD_SHAPES = [
    # (B, G, Cq, H, W, K, orient)   W % 16 == 0 -> rows move as dwordx4 pieces through LDS
    (2, 4, 24, 9, 16, 3, None),     # one strip: no neighbour on either side (halo piece invalid for both flips)
    (2, 4, 24, 7, 64, 3, None),     # c3's width, all four flips in one call
    (1, 4, 12, 32, 32, 3, None),    # c2: few strips -> the host cuts the walk into row chunks (r0 > 0)
    (3, 1, 23, 5, 48, 3, 0), (3, 1, 23, 5, 48, 3, 1), (3, 1, 23, 5, 48, 3, 2), (3, 1, 23, 5, 48, 3, 3),  # padded channels, each flip
    (2, 4, 3, 16, 16, 3, None), (2, 4, 6, 8, 32, 3, None), (1, 4, 20, 6, 32, 3, None),
    (1, 4, 32, 6, 32, 3, None),     # 144 fragments: one wave per SIMD, three load instructions per row
    (2, 4, 13, 6, 32, 2, None), (2, 1, 24, 5, 16, 2, 3),
    (1, 4, 12, 9, 32, 5, None), (2, 1, 8, 7, 48, 5, 1), (1, 1, 4, 6, 16, 5, 2),   # 5x5: the halo is the whole piece
    (1, 4, 28, 6, 32, 3, None), (1, 4, 16, 6, 32, 5, None),   # banks the staged form has no room for: the dword form, same answers
    (1, 1, 4, 1, 16, 3, 0), (1, 1, 4, 2, 16, 3, 3), (4, 4, 24, 3, 16, 3, None),   # fewer rows than the filter is tall
]


@pytest.mark.parametrize("shape", STAGED_FORWARD_SHAPES)
def test_staged_forward_against_the_oracle(shape, dev):
    """finc_conv_kernel<..., WIDE>: every flip, padded channels, single strips, row chunks, each filter size -- against
    oracle.forward_f32 (fastflow.py:31-50's four padded convs), plus the affine fold on the same shapes."""
    from fincflow_amd import ops
    B, G, Cq, H, W, K, orient = shape
    orient = ORIENT_FASTFLOW if orient is None else orient
    rng = np.random.default_rng(B * 1000 + Cq * 10 + K)
    std = (0.05 if K < 5 else 0.02) * min(1.0, (24.0 / Cq) ** 0.5)
    ws = oracle.make_stored_weights(G, Cq, K, K, orient=orient, seed=Cq + K, std=std)
    wco = oracle.canonicalize(ws, G, orient)
    x = rng.standard_normal((B, G * Cq, H, W)).astype(np.float32)
    z = oracle.forward_f32(x, wco, G, orient)
    wc = ops.canonicalize(t(ws, dev), G, orient)
    got = ops.finc_forward(t(x, dev), wc, G, orient).cpu().numpy()
    e = rel_err(got, z)
    report("staged_forward", B=B, G=G, Cq=Cq, H=H, W=W, K=K, orient=orient, err_max_norm=e, err_elementwise=elem_rel_err(got, z))
    assert e <= TOL, (shape, e)
    # grad_input runs the same kernel on the flipped image with transposed fragments: adjoint identity <gz, A x'> = <A^T gz, x'>
    gz = rng.standard_normal(z.shape).astype(np.float32)
    x2 = rng.standard_normal(x.shape).astype(np.float32)
    xt = t(x, dev).requires_grad_(True)
    ops.conv_forward(xt, [t(ws, dev)], G, orient, ops.PackedWeights()).backward(t(gz, dev))
    lhs = float(np.sum(gz.astype(np.float64) * oracle.forward_f32(x2, wco, G, orient)))
    rhs = float(np.sum(xt.grad.cpu().numpy().astype(np.float64) * x2))
    assert abs(lhs - rhs) <= 1e-4 * max(1.0, abs(lhs)), (shape, lhs, rhs)


# ------------------------------------------------------------------ fp64 entry points
@pytest.mark.parametrize("shape", [(2, 1, 5, 9, 11, 3, 0), (1, 1, 3, 7, 7, 3, 1), (2, 4, 6, 8, 12, 3, None), (1, 1, 4, 6, 5, 2, 3)])
def test_fp64_inverse_is_bit_exact_with_the_reference_solver(shape, dev):
    """The reference op's double arm (cinc_cuda_kernel_level2.cu:117) = the arithmetic of solve_parallel_mc.pyx:77-126:
    finc_inverse_f64 against the oracle's fp64 solve (itself bit-equal to the rebuilt .pyx), bit for bit, in every
    orientation (the oracle solves the canonical system: flip in, flip out, as layers/conv.py:113-163 does)."""
    from fincflow_amd import ops
    B, G, Cq, H, W, K, o = shape
    orient = ORIENT_FASTFLOW if o is None else o
    rng = np.random.default_rng(sum(shape[:6]))
    ws = oracle.make_stored_weights(G, Cq, K, K, orient=orient, seed=3).astype(np.float64)
    z = rng.standard_normal((B, G * Cq, H, W))
    wc = ops.canonicalize(t(ws, dev), G, orient)
    assert wc.dtype == torch.float64
    wco = oracle.canonicalize(ws.astype(np.float32), G, orient).astype(np.float64)
    assert np.array_equal(wc.cpu().numpy(), wco)
    out = ops.finc_inverse(t(z, dev), wc, G, orient, algo="strict").cpu().numpy()   # (auto: the matrix-core form, tests/test_gpu_round5.py)
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
    assert np.array_equal(out, ref)
    # forward in fp64 closes the loop to ~1e-15
    back = ops.finc_forward(t(out, dev), wc, G, orient, algo="strict").cpu().numpy()
    assert rel_err(back, z) <= 1e-12
    # the drop-in op dispatches on dtype like the reference (float / double), canonical orientation
    if orient == 0:
        y = torch.empty_like(t(z, dev))
        res = ops.inverse(t(z, dev), wc, y)
        # (the op runs FINC_ALGO_AUTO: for this bank the fp64 matrix-core form, which adds a pixel's terms in another order)
        assert res[0].data_ptr() == y.data_ptr() and rel_err(y.cpu().numpy(), ref) <= 1e-12
    with pytest.raises(ValueError):
        ops.finc_inverse(t(z, dev), wc.float(), G, orient)        # dtype mismatch is an error, not a silent cast


# ------------------------------------------------------------------ PaddedConv2d(bias=True)
@pytest.mark.parametrize("order", ["TL", "BR"])
def test_padded_conv_with_bias(order, dev):
    """layers/conv.py:60,102-117: conv(pad(x)) + b forward, reverse subtracts b first.  The holder conv stays bias-free as in
    the reference (its nn.Conv2d is built with bias=False whatever the argument), the bias is the layer's own zero-initialised
    parameter: the state dict is the reference's key plus `bias`."""
    import torch.nn.functional as F
    from fincflow_amd import PaddedConv2d
    torch.manual_seed(4)
    m = PaddedConv2d(6, 6, (3, 3), bias=True, order=order)
    assert sorted(m.state_dict().keys()) == ["bias", "conv.weight"] and m.conv.bias is None
    assert torch.equal(m.bias.detach(), torch.zeros(6))
    with torch.no_grad():
        m.bias.copy_(torch.randn(6))
    x = torch.randn(2, 6, 9, 12)
    ref = F.conv2d(F.pad(x, m.pad), m.conv.weight.detach(), m.bias.detach())
    m = m.to(dev)
    with torch.no_grad():
        z, ld = m(x.to(dev))
        assert ld == 0.0 and rel_err(z.cpu().numpy(), ref.numpy()) <= TOL
        xr, ld2 = m.reverse(z)
    assert ld2 == 0 and rel_err(xr.cpu().numpy(), x.numpy()) <= TOL
    xg = x.to(dev).requires_grad_(True)                           # autograd path: bias gets its gradient too
    zz, _ = m(xg)
    zz.sum().backward()
    assert m.bias.grad is not None and torch.allclose(m.bias.grad.cpu(), torch.full((6,), 2.0 * 9 * 12))


# ------------------------------------------------------------------ unaligned views through the module
def test_unaligned_view_through_fastflowunit_reverse(dev):
    """16-byte alignment is needed for the MFMA path, "else it falls back".  A contiguous view that
    starts 4 bytes into an allocation goes through FastFlowUnit.reverse / reverse_affine / PaddedConv2d.reverse."""
    from fincflow_amd import FastFlowUnit, glow, ops
    torch.manual_seed(6)
    B, C, H, W = 160, 16, 8, 8             # (more problems than the role-split kernel takes: that one has no alignment rule)
    unit = FastFlowUnit(C, C, 3).to(dev)
    x = torch.randn(B, C, H, W, device=dev)
    with torch.no_grad():
        z, _ = unit(x)
        n = z.numel()
        buf = torch.zeros(n + 8, device=dev)
        buf[1:n + 1] = z.flatten()
        zv = buf[1:n + 1].view(B, C, H, W)
        assert zv.is_contiguous() and zv.data_ptr() % 16 == 4
        xr = unit.reverse(zv)
        assert rel_err(xr.cpu().numpy(), x.cpu().numpy()) <= TOL
        # bit-equal to the strict kernel: that is what the fallback runs
        wc = ops.canonicalize(torch.cat(unit._weights()).detach().contiguous(), 4, ORIENT_FASTFLOW)
        assert torch.equal(xr, ops.finc_inverse(z, wc, algo="strict"))
        an = glow.ActNorm(C).to(dev)
        an.initialized.fill_(1)
        assert unit.reverse_affine(zv, an.log_scale, an.translation) is None     # caller runs the two layers separately
        y, _ = unit.conv_tl.reverse(zv[:, :4].contiguous())
        assert y.shape == (B, 4, H, W)


def test_unaligned_view_through_the_forward(dev):
    """The staged forward moves 16-byte pieces; a float-aligned view (W % 16 == 0, so the staged form would be picked)
    must take the dword form and give the same numbers -- input view, output view, and both."""
    from fincflow_amd import FastFlowUnit
    torch.manual_seed(7)
    B, C, H, W = 2, 48, 6, 32
    unit = FastFlowUnit(C, C, 3).to(dev)
    x = torch.randn(B, C, H, W, device=dev)
    with torch.no_grad():
        z, _ = unit(x)                                # aligned: the staged form
        n = x.numel()
        buf = torch.zeros(n + 8, device=dev)
        buf[1:n + 1] = x.flatten()
        xv = buf[1:n + 1].view(B, C, H, W)
        assert xv.is_contiguous() and xv.data_ptr() % 16 == 4
        zv, _ = unit(xv)
        assert rel_err(zv.cpu().numpy(), z.cpu().numpy()) <= 1e-6
        obuf = torch.zeros(n + 8, device=dev)
        ov = obuf[3:n + 3].view(B, C, H, W)
        unit._cache.forward(xv, unit._weights(), 4, ORIENT_FASTFLOW, out=ov)
        assert rel_err(ov.cpu().numpy(), z.cpu().numpy()) <= 1e-6
        assert float(obuf[:3].abs().sum()) == 0.0 and float(obuf[n + 3:].abs().sum()) == 0.0   # nothing written outside the view


# ------------------------------------------------------------------ the 1x1 conv next to the unit
@pytest.mark.parametrize("shape", [(3, 96, 20, 24), (2, 12, 16, 16), (5, 24, 8, 8), (4, 48, 4, 4), (2, 192, 9, 8), (1, 4, 7, 7),
                                   (2, 16, 5, 3), (3, 64, 6, 10), (1, 128, 3, 5), (2, 8, 1, 1)])
def test_mix_kernel_is_the_1x1_conv(shape, dev):
    """finc_mix_f32 against F.conv2d on the CPU in fp64 (layers/conv1x1.py:29-43), with and without bias, in place and
    out of place, even and odd pixel counts (two pixels per lane / one)."""
    import torch.nn.functional as F
    from fincflow_amd import ops
    B, C, H, W = shape
    assert ops.mix_supported(C)
    torch.manual_seed(sum(shape))
    x = torch.randn(B, C, H, W)
    M = torch.randn(C, C) / C ** 0.5
    b = torch.randn(C)
    ref = F.conv2d(x.double(), M.double().view(C, C, 1, 1), b.double()).numpy()
    ref0 = F.conv2d(x.double(), M.double().view(C, C, 1, 1)).numpy()
    xd, Md, bd = x.to(dev), M.to(dev), b.to(dev)
    out = ops.finc_mix(xd, Md, bd)
    e = rel_err(out.cpu().numpy(), ref)
    report("mix", shape=list(shape), err_max_norm=e, err_elementwise=elem_rel_err(out.cpu().numpy(), ref))
    assert e <= TOL
    assert rel_err(ops.finc_mix(xd, Md).cpu().numpy(), ref0) <= TOL
    y = xd.clone()
    assert ops.finc_mix(y, Md, bd, out=y).data_ptr() == y.data_ptr() and torch.equal(y, out)      # in place
    assert not ops.mix_supported(20) and not ops.mix_supported(7)
    from fincflow_amd import _lib
    with pytest.raises(_lib.FincError):
        ops.finc_mix(torch.randn(1, 20, 4, 4, device=dev), torch.eye(20, device=dev))


def test_conv1x1_module_runs_on_the_mix_kernel(dev):
    """glow.Conv1x1 (layers/conv1x1.py:8-49): under no_grad both directions are one finc_mix_f32 launch; reverse(forward(x))
    == x; the autograd path (training) still matches; and [Conv1x1, ActNorm] in a reverse chain is the affine-folded call."""
    from fincflow_amd import glow
    torch.manual_seed(3)
    np.random.seed(3)
    c = glow.Conv1x1(48).to(dev)
    x = torch.randn(4, 48, 8, 8, device=dev)
    with torch.no_grad():
        z, ldj = c(x)
        z_ref = torch.nn.functional.conv2d(x, c.W.view(48, 48, 1, 1))
        assert rel_err(z.cpu().numpy(), z_ref.cpu().numpy()) <= TOL
        assert abs(float(ldj) - 64 * float(torch.slogdet(c.W)[1])) < 1e-3
        assert rel_err(c.reverse(z).cpu().numpy(), x.cpu().numpy()) <= TOL
        an = glow.ActNorm(48).to(dev)
        an.log_scale.copy_(0.2 * torch.randn(48, device=dev))
        an.translation.copy_(torch.randn(48, device=dev))
        an.initialized.fill_(1)
        fused = c.reverse_then_affine(z, an.log_scale, an.translation)
        assert rel_err(fused.cpu().numpy(), an.reverse(c.reverse(z)).cpu().numpy()) <= TOL
        c.W.mul_(1.01)                                      # in-place update: the cached inverse must follow
        assert rel_err(c.reverse(c(x)[0]).cpu().numpy(), x.cpu().numpy()) <= TOL
    xg = x.clone().requires_grad_(True)
    zg, _ = c(xg)
    zg.sum().backward()
    assert xg.grad is not None and c.W.grad is not None


@pytest.mark.parametrize("shape", [(3, 96, 20, 24, 3), (2, 48, 16, 16, 3), (2, 12, 9, 8, 3), (2, 192, 12, 16, 3), (1, 64, 10, 12, 5)])
def test_affine_fold_behind_the_forward(shape, dev):
    """SURVEY 8 f3, forward direction: ActNorm.forward(FastFlowUnit.forward(x)) in one launch (filter rows scaled, accumulators
    started from the shift; K-split banks add the shift once).  Against the two layers one after the other and the oracle;
    FlowSequential.forward takes the fused path by itself and returns the same log-probabilities."""
    from fincflow_amd import FastFlowUnit, FlowSequential, glow
    from fincflow_amd.layers import StandardNormal
    B, C, H, W, K = shape
    torch.manual_seed(sum(shape))
    unit = FastFlowUnit(C, C, K).to(dev)
    an = glow.ActNorm(C).to(dev)
    with torch.no_grad():
        an.log_scale.copy_(0.3 * torch.randn(C, device=dev))
        an.translation.copy_(torch.randn(C, device=dev))
        an.initialized.fill_(1)
        x = torch.randn(B, C, H, W, device=dev)
        two = an(unit(x)[0])[0]
        fused = unit.forward_affine(x, an.log_scale, an.translation)
    assert fused is not None
    assert rel_err(fused.cpu().numpy(), two.cpu().numpy()) <= TOL
    ws = torch.cat(unit._weights()).detach().cpu().numpy()
    wco = oracle.canonicalize(ws, 4, ORIENT_FASTFLOW)
    z = oracle.forward_f32(x.cpu().numpy(), wco)
    ref = (z - an.translation.detach().cpu().numpy()[None, :, None, None]) * np.exp(-an.log_scale.detach().cpu().numpy())[None, :, None, None]
    assert rel_err(fused.cpu().numpy(), ref) <= TOL
    seq = FlowSequential(StandardNormal((C, H, W)), unit, an)
    with torch.no_grad():
        seq.fuse_affine = True
        za, lpa = seq(x)
        seq.fuse_affine = False
        zb, lpb = seq(x)
        assert rel_err(za.cpu().numpy(), zb.cpu().numpy()) <= TOL and torch.allclose(lpa, lpb, rtol=1e-5, atol=1e-3)
        an.translation.add_(0.5)                        # in-place update: the cached bank must follow
        seq.fuse_affine = True
        zc, _ = seq(x)
        assert rel_err(zc.cpu().numpy(), an(unit(x)[0])[0].cpu().numpy()) <= TOL
        an.reset_initialization()                        # not initialised: no fold (ActNorm.forward must see the data)
        assert an.forward_affine_params() is None


def test_helper_wave_protocol_never_times_out(dev):
    """Form 3 of the inverse pairs each compute wave with a helper wave through progress words in LDS; every wait is bounded
    and a wait that gives up is counted.  Many launches at several shapes: bit-identical results, zero timeouts."""
    from fincflow_amd import FastFlowUnit, _lib
    torch.manual_seed(1)
    for (B, C, H, W) in ((256, 96, 64, 64), (129, 96, 16, 32), (130, 96, 40, 48), (512, 64, 16, 16), (260, 48, 32, 32)):
        v = _lib.inverse_variant(B, 4, C // 4, H, W, 3, 3)
        assert v is not None and v["sec"] == 3, v
        unit = FastFlowUnit(C, C, 3).to(dev)
        x = torch.randn(B, C, H, W, device=dev)
        with torch.no_grad():
            z, _ = unit(x)
            ref = unit.reverse(z).clone()
            assert rel_err(ref.cpu().numpy(), x.cpu().numpy()) <= TOL
            for _ in range(300):
                assert torch.equal(unit.reverse(z), ref)
    assert _lib.hlp_timeouts() == 0


FAULT_SCRIPT = r"""
import sys, torch
sys.path.insert(0, %(repo)r)
from fincflow_amd import FastFlowUnit, _lib
assert _lib.build_flags() != 0                       # a test-only build: the knobs announce themselves
dev = torch.device("cuda:0")
unit = FastFlowUnit(96, 96, 3).to(dev)
z = torch.randn(129, 96, 32, 32, device=dev)         # 516 problems: the helper-wave form (beyond the small-batch forms)
assert _lib.inverse_variant(129, 4, 24, 32, 32, 3, 3)["sec"] == 3
with torch.no_grad():
    unit.reverse(z)                                  # the injected fault: the helper never announces a landing -> waits give up
    torch.cuda.synchronize()
    n = _lib.hlp_timeouts()
    assert n > 0, "the injected fault did not show in the counter"
    try:
        unit.reverse(z)
        print("NOT-RAISED")
    except _lib.FincError as e:
        print("RAISED", "gave up" in str(e))
    _lib.clear_fault()
    unit.reverse(z)                                  # the gate is open again (the build still faults: the word is set anew)
    torch.cuda.synchronize()
    try:
        unit.reverse(z)
        print("NOT-RAISED-2")
    except _lib.FincError:
        print("RAISED-2")
"""


def test_a_protocol_timeout_is_an_error_not_a_silent_wrong_answer(dev, tmp_path):
    """ADVICE r2 / VERDICT r2 weak 4: a helper-wave wait that gives up used to bump a counter nobody read.  Now the kernel
    also sets a word in mapped host memory and the next finc_* call on the device returns FINC_ERR_LAUNCH (sticky until
    finc_clear_fault()).  Exercised with a TEST-ONLY build (-DFINC_EXPERIMENT -DFINC_HLP_INJECT_TIMEOUT: the helper stops
    announcing its landings) in a process of its own; the product library never carries the knob (finc_build_flags() == 0)."""
    import os
    import shutil
    import subprocess
    import sys
    from helpers import REPO
    if not shutil.which("hipcc"):
        pytest.skip("no hipcc on this box")
    csrc = os.path.join(REPO, "fincflow_amd", "csrc")
    objs = [os.path.join(csrc, o) for o in ("finc_abi.o", "finc_generic.o", "finc_chain.o", "finc_f64.o", "finc_big.o", "finc_conv.o", "finc_wino.o", "finc_gradw.o", "finc_mix.o", "finc_probe.o", "finc_wino5.o", "finc_wino4m.o", "finc_stream.o")]
    if not all(os.path.exists(o) for o in objs):
        pytest.skip("object files of the product build are not in the tree")
    lib = str(tmp_path / "libfinc_faulty.so")
    flags = ["-O3", "-fPIC", "--offload-arch=gfx950", "-std=c++20", "-mllvm", "-amdgpu-mfma-vgpr-form", "-DFINC_EXPERIMENT", "-DFINC_ONLY_C3",
             "-DFINC_HLP_INJECT_TIMEOUT", "-DFINC_HLP_BUDGET_LOG2=8"]
    for name in ("finc_mfma", "finc_split"):
        subprocess.run(["hipcc"] + flags + ["-c", os.path.join(csrc, name + ".hip"), "-o", str(tmp_path / (name + ".o"))], check=True,
                       capture_output=True, timeout=600)
    subprocess.run(["hipcc", "--offload-arch=gfx950", "-shared", "-fPIC", "-o", lib] + objs +
                   [str(tmp_path / "finc_mfma.o"), str(tmp_path / "finc_split.o")], check=True, capture_output=True, timeout=300)
    r = subprocess.run([sys.executable, "-c", FAULT_SCRIPT % {"repo": REPO}], env=dict(os.environ, FINCFLOW_LIB=lib),
                       capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stderr[-2000:]
    assert "RAISED True" in r.stdout and "RAISED-2" in r.stdout and "NOT-RAISED" not in r.stdout, r.stdout
    from fincflow_amd import _lib
    assert _lib.build_flags() == 0 and _lib.hlp_timeouts() == 0           # this process runs the product library
