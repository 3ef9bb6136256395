"""Host-side operators over the C ABI (include/finc.h).

`inverse(input, kernel, output)` mirrors the reference's one native op
(fastflow/utils/fastflow_cuda_inverse/cinc_cuda_level2.cpp:19-32): same name,
same argument meaning, same in-place-and-return-alias behaviour, same
RuntimeError for non-device / non-contiguous tensors.  `finc_inverse` /
`finc_forward` are the orientation-aware calls FastFlowUnit uses (no flips, no
chunk/cat copies: fastflow.py:78-100 collapses into one launch).
"""
import threading

import torch

from . import _lib

ORDER_BITS = {"TL": 0, "TR": 1, "BL": 2, "BR": 3}
ORIENT_FASTFLOW = 0xE4  # TL,TR,BL,BR (fastflow.py:24-27)

_workspaces = {}


def _stream_ptr(t):
    return torch.cuda.current_stream(t.device).cuda_stream


def _require_device(t, name, dtype=torch.float32):
    # same conditions, same exception type as CHECK_INPUT (cinc_cuda_level2.cpp:15-17)
    if not t.is_cuda:
        raise RuntimeError(f"{name} must be a CUDA tensor (fincflow_amd has no CPU fallback)")
    if not t.is_contiguous():
        raise RuntimeError(f"{name} must be contiguous")
    if t.dtype != dtype:
        raise ValueError(f"{name} must be {dtype}, got {t.dtype}")


def _float_dtype(t, name):
    """float and double, the reference op's dispatch (AT_DISPATCH_FLOATING_TYPES, cinc_cuda_kernel_level2.cu:117)."""
    if t.dtype not in (torch.float32, torch.float64):
        raise ValueError(f"{name} must be float32 or float64, got {t.dtype}")
    return t.dtype


def release_workspaces():
    """Drop every cached per-(device, stream) scratch buffer (they are re-created on demand).  For long-lived
    processes that cycle through many streams; call it at a synchronisation point."""
    _workspaces.clear()


def _workspace(device, nbytes):
    key = (device.index, torch.cuda.current_stream(device).cuda_stream)
    ws = _workspaces.get(key)
    if ws is None or ws.numel() < nbytes:
        ws = torch.empty(max(nbytes, 1 << 20), dtype=torch.uint8, device=device)
        _workspaces[key] = ws
    return ws


def _dims(act, w, G):
    if act.dim() != 4 or w.dim() != 4:
        raise ValueError("expected activations [B,C,H,W] and weights [G*Cq,Cq,KH,KW]")
    B, C, H, W = act.shape
    if C % G != 0:
        raise ValueError(f"channels {C} not divisible by groups {G}")
    Cq = C // G
    if w.shape[0] != C or w.shape[1] != Cq:
        raise ValueError(f"weights {tuple(w.shape)} do not match C={C}, Cq={Cq}")
    if act.device != w.device:
        raise ValueError("activations and weights on different devices")
    return B, Cq, H, W, w.shape[2], w.shape[3]


def canonicalize(w_stored, G, orient):
    """State-dict form -> TL-canonical (fastflow.py:79-84).  The flip is an involution, so the same call
    maps canonical gradients back to stored form."""
    dt = _float_dtype(w_stored, "weights")
    _require_device(w_stored, "weights", dt)
    out = torch.empty_like(w_stored)
    Cq = w_stored.shape[0] // G
    fn = "finc_canonicalize_weights_f32" if dt == torch.float32 else "finc_canonicalize_weights_f64"
    with torch.cuda.device(w_stored.device):
        st = getattr(_lib.lib(), fn)(w_stored.data_ptr(), out.data_ptr(), G, Cq, w_stored.shape[2],
                                     w_stored.shape[3], orient, _stream_ptr(w_stored))
    _lib.check(st, fn)
    return out


def check_invariant(w_canon, G):
    """Raises if the corner tap is not unit lower triangular (layers/conv.py:63-70).  Synchronises."""
    _require_device(w_canon, "weights")
    with torch.cuda.device(w_canon.device):
        st = _lib.lib().finc_check_invariant_f32(w_canon.data_ptr(), G, w_canon.shape[0] // G, w_canon.shape[2],
                                                 w_canon.shape[3], _stream_ptr(w_canon))
    _lib.check(st, "finc_check_invariant_f32")


def _run(fn_name, act, w_canon, G, orient, algo, out):
    dt = _float_dtype(act, "input")
    _require_device(act, "input", dt)
    _require_device(w_canon, "kernel", dt)
    B, Cq, H, W, KH, KW = _dims(act, w_canon, G)
    if out is None:
        out = torch.empty_like(act)
    else:
        _require_device(out, "output", dt)
        if out.shape != act.shape or out.device != act.device:
            raise ValueError("output must match input in shape and device")
    if act.numel() == 0:
        return out
    L = _lib.lib()
    if dt == torch.float64:
        # strict: the reference-order fp64 kernels (bit-exact with the reference's Cython solver; no packed form, no workspace);
        # auto / mfma: the matrix-core form where the bank has one (finc_f64.hip: Cq <= 24 at 3x3, <= 32 at 2x2), else strict
        fn64 = fn_name.replace("_f32", "_f64")
        with torch.cuda.device(act.device):
            if algo == "strict":
                st = getattr(L, fn64)(act.data_ptr(), w_canon.data_ptr(), out.data_ptr(), B, G, Cq, H, W, KH, KW, orient,
                                      _stream_ptr(act))
            else:
                ws = _workspace(act.device, L.finc_f64_workspace_bytes(G, Cq, KH, KW))
                fn64 += "_algo"
                st = getattr(L, fn64)(act.data_ptr(), w_canon.data_ptr(), out.data_ptr(), B, G, Cq, H, W, KH, KW, orient,
                                      _lib.ALGO[algo], ws.data_ptr(), ws.numel(), _stream_ptr(act))
        _lib.check(st, fn64)
        return out
    if fn_name == "finc_inverse_f32":      # room for the zero-padded copy an odd width is solved on
        nbytes = L.finc_inverse_workspace_bytes(B, G, Cq, H, W, KH, KW)
    else:
        nbytes = L.finc_workspace_bytes(G, Cq, KH, KW)
    with torch.cuda.device(act.device):
        ws = _workspace(act.device, nbytes)
        st = getattr(L, fn_name)(act.data_ptr(), w_canon.data_ptr(), out.data_ptr(), B, G, Cq, H, W, KH, KW, orient,
                                 _lib.ALGO[algo], ws.data_ptr(), ws.numel(), _stream_ptr(act))
    _lib.check(st, fn_name)
    return out


def finc_inverse(z, w_canon, G=4, orient=ORIENT_FASTFLOW, algo="auto", out=None):
    """x = inverse(z) for G groups with per-group orientation; one asynchronous launch on the current stream."""
    return _run("finc_inverse_f32", z, w_canon, G, orient, algo, out)


def finc_forward(x, w_canon, G=4, orient=ORIENT_FASTFLOW, algo="auto", out=None):
    """z = forward(x); the layer's logdet is identically 0 (layers/conv.py:106)."""
    return _run("finc_forward_f32", x, w_canon, G, orient, algo, out)


def finc_backward(grad_z, x, w_canon, G, orient, need_gx=True, need_gw=True):
    _require_device(grad_z, "grad_output")
    B, Cq, H, W, KH, KW = _dims(grad_z, w_canon, G)
    gx = torch.empty_like(grad_z) if need_gx else None
    gw = torch.empty_like(w_canon) if need_gw else None
    if grad_z.numel() == 0:
        if gw is not None:
            gw.zero_()
        return gx, gw
    L = _lib.lib()
    with torch.cuda.device(grad_z.device):
        ws = _workspace(grad_z.device, L.finc_backward_workspace_bytes(B, G, Cq, H, W, KH, KW))
        st = L.finc_backward_f32(grad_z.data_ptr(), x.data_ptr() if x is not None else None,
                                 w_canon.data_ptr(), gx.data_ptr() if gx is not None else None,
                                 gw.data_ptr() if gw is not None else None, B, G, Cq, H, W, KH, KW, orient,
                                 ws.data_ptr(), ws.numel(), _stream_ptr(grad_z))
    _lib.check(st, "finc_backward_f32")
    return gx, gw


def mix_supported(C):
    return bool(_lib.lib().finc_mix_supported_f32(int(C)))


def finc_mix(x, mat, bias=None, out=None):
    """out[b, :, h, w] = mat @ x[b, :, h, w] + bias: the 1x1 convolution of a flow step (layers/conv1x1.py:29-43) with
    whatever per-channel affine neighbour the caller folded into `mat` / `bias`, as one streaming HIP launch.
    x [B,C,H,W] fp32 contiguous on the device, mat [C,C] (out, in), bias [C] or None.  `out` may be `x`."""
    _require_device(x, "input")
    _require_device(mat, "matrix")
    if x.dim() != 4 or mat.shape != (x.shape[1], x.shape[1]) or mat.device != x.device:
        raise ValueError("expected activations [B,C,H,W] and a [C,C] matrix on the same device")
    if bias is not None:
        _require_device(bias, "bias")
        if bias.numel() != x.shape[1]:
            raise ValueError("bias must have one entry per channel")
    if out is None:
        out = torch.empty_like(x)
    else:
        _require_device(out, "output")
        if out.shape != x.shape or out.device != x.device:
            raise ValueError("output must match input in shape and device")
    if x.numel() == 0:
        return out
    B, C, H, W = x.shape
    with torch.cuda.device(x.device):
        st = _lib.lib().finc_mix_f32(x.data_ptr(), mat.data_ptr(), bias.data_ptr() if bias is not None else None,
                                     out.data_ptr(), B, C, H * W, _stream_ptr(x))
    _lib.check(st, "finc_mix_f32")
    return out


def inverse(input, kernel, output):
    """Drop-in for the reference extension's `inverse` (cinc_cuda_level2.cpp:19-32).

    input  [B,C,H,W]        already flipped to TL-canonical per group by the caller (fastflow.py:85-90)
    kernel [G*Cq,Cq,KH,KW]  TL-canonical (fastflow.py:79-84); G = kernel.shape[0] // kernel.shape[1]
    output [B,C,H,W]        written in place; a one-element list aliasing it is returned.
    """
    dt = _float_dtype(input, "input")
    for t, n in ((input, "input"), (kernel, "kernel"), (output, "output")):
        _require_device(t, n, dt)
    G = kernel.shape[0] // kernel.shape[1]
    finc_inverse(input, kernel, G=G, orient=0, out=output)
    return [output]


class _DeviceBank:
    """What PackedWeights holds for ONE device."""
    __slots__ = ("key", "keep", "validated", "w_canon", "packed_inv", "packed_fwd", "packed_aff", "aff_key", "packed_faff",
                 "faff_key", "linv")

    def __init__(self):
        self.key = None
        self.keep = None          # the source tensors' storages, kept alive while the entry is (see PackedWeights._get)
        self.validated = False    # check_invariant has run on THIS weight version
        self.w_canon = None
        self.packed_inv = None
        self.packed_fwd = None
        self.packed_aff = None
        self.aff_key = None
        self.packed_faff = None
        self.faff_key = None
        self.linv = None


class PackedWeights:
    """Sampling-time cache for one weight version: the TL-canonical bank (fastflow.py:79-84, done once instead of
    every call), its invariant check, and the packed MFMA fragments (finc_pack_inverse_weights_f32), so that a
    sampling step is exactly one kernel launch (finc_inverse_packed_f32).  Rebuilt when a source tensor changes.

    State is kept PER DEVICE: the reference wraps its models in nn.DataParallel (fastflow_cifar_multi_gpu.py:439-440),
    whose replicas are shallow copies that share this object while their weights live on different devices and their
    forwards run on different threads."""

    def __init__(self):
        self._banks = {}
        self._lock = threading.Lock()

    def _bank(self, device):
        b = self._banks.get(device)
        if b is None:
            with self._lock:
                b = self._banks.setdefault(device, _DeviceBank())
        return b

    def invalidate(self):
        """Force a rebuild on the next call.  Needed after writes that do not bump Tensor._version
        (torch.distributed collectives, writes through `.data`)."""
        with self._lock:
            self._banks = {}

    # what single-device callers and tests read
    @property
    def w_canon(self):
        banks = list(self._banks.values())
        return banks[0].w_canon if len(banks) == 1 else None

    def get(self, weights, G, orient):
        return self._get(weights, G, orient).w_canon

    def _get(self, weights, G, orient, validate=True):
        """`validate=False`: the training path -- the gradient mask keeps the corner tap unit triangular (layers/conv.py:98-99,
        applied inside the HIP backward), and the check is a device->host synchronisation per layer and step."""
        bank = self._bank(weights[0].device)
        # The entry is keyed on (address, version counter) of every source tensor AND holds their storages alive: a weight
        # rebound through `.data` to a fresh tensor keeps its version counter, and the address of a freed tensor is the first
        # one the allocator hands out again -- with the old storage still referenced here the new one cannot land on it.
        key = tuple((w.data_ptr(), w._version) for w in weights) + (orient,)
        if key != bank.key:
            ws = torch.cat([w.detach() for w in weights], dim=0).contiguous() if len(weights) > 1 else weights[0].detach().contiguous()
            bank.w_canon = canonicalize(ws, G, orient)
            bank.validated = False
            bank.packed_inv = None
            bank.packed_fwd = None
            bank.packed_aff = None
            bank.packed_faff = None
            bank.linv = None
            bank.key = key
            bank.keep = tuple(w.untyped_storage() for w in weights)
        # An entry the training path created (validate=False) is NOT validated: the first inference call on the same weight
        # version runs the check, so an optimiser effect outside the in-kernel gradient mask (weight decay on the diagonal,
        # a manual edit followed by a forward under grad) cannot reach the inverse unnoticed.
        if validate and not bank.validated:
            check_invariant(bank.w_canon, G)
            bank.validated = True
        return bank

    def forward(self, x, weights, G, orient, out=None, validate=True):
        """Forward on the cached canonical bank + cached strip-kernel fragments (also the forward of the autograd path: a
        weight version that has not changed since the last call -- evaluation under grad, several micro-batches per
        optimiser step -- costs no cat / canonicalise / pack launch)."""
        bank = self._get(weights, G, orient, validate)
        w_canon = bank.w_canon
        _require_device(x, "input")
        B, Cq, H, W, KH, KW = _dims(x, w_canon, G)
        L = _lib.lib()
        if x.numel() == 0 or L.finc_forward_algo_for(Cq, H, W, KH, KW) != _lib.ALGO["mfma"]:
            return finc_forward(x, w_canon, G, orient, out=out)
        with torch.cuda.device(x.device):
            if bank.packed_fwd is None:
                bank.packed_fwd = torch.empty(L.finc_workspace_bytes(G, Cq, KH, KW), dtype=torch.uint8, device=x.device)
                _lib.check(L.finc_pack_forward_weights_f32(w_canon.data_ptr(), bank.packed_fwd.data_ptr(), G, Cq, KH, KW,
                                                           _stream_ptr(x)), "finc_pack_forward_weights_f32")
            if out is None:
                out = torch.empty_like(x)
            _lib.check(L.finc_forward_packed_f32(x.data_ptr(), bank.packed_fwd.data_ptr(), out.data_ptr(), B, G, Cq, H, W,
                                                 KH, KW, orient, _stream_ptr(x)), "finc_forward_packed_f32")
        return out

    def forward_affine(self, x, weights, G, orient, log_scale, translation, out=None):
        """(forward(x) - translation) * exp(-log_scale) in ONE launch: the per-channel affine layer BEHIND the unit in the
        model (ActNorm.forward, layers/actnorm.py:39-46) folded into the forward bank -- filter rows scaled, accumulators
        started from the shift.  Returns None when the shape has no MFMA strip kernel (the caller runs the two layers)."""
        bank = self._get(weights, G, orient)
        w_canon = bank.w_canon
        _require_device(x, "input")
        B, Cq, H, W, KH, KW = _dims(x, w_canon, G)
        L = _lib.lib()
        if x.numel() == 0 or L.finc_forward_algo_for(Cq, H, W, KH, KW) != _lib.ALGO["mfma"]:
            return None
        key = (log_scale.data_ptr(), log_scale._version, translation.data_ptr(), translation._version)
        with torch.cuda.device(x.device):
            if bank.packed_faff is None or bank.faff_key != key:
                scale = torch.exp(-log_scale.detach().float()).contiguous()
                shift = (-translation.detach().float() * scale).contiguous()
                if scale.numel() != G * Cq:
                    raise ValueError("affine parameters must have one entry per channel")
                bank.packed_faff = torch.empty(L.finc_workspace_bytes(G, Cq, KH, KW), dtype=torch.uint8, device=x.device)
                _lib.check(L.finc_pack_forward_weights_affine_f32(w_canon.data_ptr(), scale.data_ptr(), shift.data_ptr(),
                                                                  bank.packed_faff.data_ptr(), G, Cq, KH, KW, _stream_ptr(x)),
                           "finc_pack_forward_weights_affine_f32")
                bank.faff_key = key
            if out is None:
                out = torch.empty_like(x)
            _lib.check(L.finc_forward_packed_f32(x.data_ptr(), bank.packed_faff.data_ptr(), out.data_ptr(), B, G, Cq, H, W,
                                                 KH, KW, orient, _stream_ptr(x)), "finc_forward_packed_f32")
        return out

    @staticmethod
    def _packed_path_ok(L, t, out, Cq, H, W, KH, KW):
        """The packed launch streams 16-byte pieces: it needs an MFMA instantiation AND 16-byte aligned activations
        (a view into a larger allocation may be only 4-byte aligned; INTEGRATION.md: such calls fall back)."""
        if t.numel() == 0 or L.finc_inverse_algo_for(Cq, H, W, KH, KW) != _lib.ALGO["mfma"]:
            return False
        ptrs = t.data_ptr() | (out.data_ptr() if out is not None else 0)
        return (ptrs & 15) == 0

    def inverse(self, z, weights, G, orient, out=None):
        bank = self._get(weights, G, orient)
        w_canon = bank.w_canon
        _require_device(z, "input")
        B, Cq, H, W, KH, KW = _dims(z, w_canon, G)
        L = _lib.lib()
        if out is None and z.numel():
            out = torch.empty_like(z)
        if not self._packed_path_ok(L, z, out, Cq, H, W, KH, KW):
            return finc_inverse(z, w_canon, G, orient, out=out)
        with torch.cuda.device(z.device):
            if bank.packed_inv is None:
                bank.packed_inv = torch.empty(L.finc_workspace_bytes(G, Cq, KH, KW), dtype=torch.uint8, device=z.device)
                _lib.check(L.finc_pack_inverse_weights_f32(w_canon.data_ptr(), bank.packed_inv.data_ptr(), G, Cq, KH, KW,
                                                           _stream_ptr(z)), "finc_pack_inverse_weights_f32")
            _lib.check(L.finc_inverse_packed_f32(z.data_ptr(), bank.packed_inv.data_ptr(), out.data_ptr(), B, G, Cq, H, W,
                                                 KH, KW, orient, _stream_ptr(z)), "finc_inverse_packed_f32")
        return out

    def lead_inverse(self, weights, G, orient):
        """Linv_g = inverse of the unit lower triangular tap of the pixel itself (canonical tap [KH-1, KW-1],
        layers/conv.py:63-70), [G, Cq, Cq] fp32 (solved in fp64), cached per weight version: what a channel mix in front of
        the unit multiplies into its matrix so that `inverse_premultiplied` can skip the z-term."""
        bank = self._get(weights, G, orient)
        if bank.linv is None:
            wc = bank.w_canon
            Cq = wc.shape[1]
            lead = wc.view(G, Cq, Cq, wc.shape[2], wc.shape[3])[:, :, :, -1, -1].double()
            eye = torch.eye(Cq, dtype=torch.float64, device=wc.device).expand(G, Cq, Cq)
            bank.linv = torch.linalg.solve_triangular(lead, eye, upper=False, unitriangular=True).float().contiguous()
        return bank.linv

    def premultiplied_supported(self, shape, weights, G, orient):
        """Does `inverse_premultiplied` exist for activations of this shape (the helper-wave form of the inverse: a problem
        set that fills the chip, W % 16 == 0)?"""
        w_canon = self._get(weights, G, orient).w_canon
        B, C, H, W = shape
        Cq, KH, KW = w_canon.shape[1], w_canon.shape[2], w_canon.shape[3]
        return C == G * Cq and bool(_lib.lib().finc_inverse_premultiplied_supported(B, G, Cq, H, W, KH, KW))

    def inverse_premultiplied(self, zp, weights, G, orient, out=None):
        """inverse(z) given zp = blockdiag(Linv) z (the channel mix in front of the unit applied Linv for free),
        ONE launch without the z-term's MFMAs.  None when the shape has no such kernel or the activations are not 16-byte
        aligned (the caller runs the plain chain)."""
        bank = self._get(weights, G, orient)
        w_canon = bank.w_canon
        _require_device(zp, "input")
        B, Cq, H, W, KH, KW = _dims(zp, w_canon, G)
        L = _lib.lib()
        if zp.numel() == 0 or not L.finc_inverse_premultiplied_supported(B, G, Cq, H, W, KH, KW):
            return None
        if out is None:
            out = torch.empty_like(zp)
        if (zp.data_ptr() | out.data_ptr()) & 15:
            return None
        with torch.cuda.device(zp.device):
            if bank.packed_inv is None:
                bank.packed_inv = torch.empty(L.finc_workspace_bytes(G, Cq, KH, KW), dtype=torch.uint8, device=zp.device)
                _lib.check(L.finc_pack_inverse_weights_f32(w_canon.data_ptr(), bank.packed_inv.data_ptr(), G, Cq, KH, KW,
                                                           _stream_ptr(zp)), "finc_pack_inverse_weights_f32")
            _lib.check(L.finc_inverse_packed_premultiplied_f32(zp.data_ptr(), bank.packed_inv.data_ptr(), out.data_ptr(), B, G,
                                                               Cq, H, W, KH, KW, orient, _stream_ptr(zp)),
                       "finc_inverse_packed_premultiplied_f32")
        return out

    def inverse_affine(self, y, weights, G, orient, log_scale, translation, out=None):
        """inverse(exp(log_scale) * y + translation) in ONE launch: the per-channel affine layer in front
        of the unit in the reverse chain (ActNorm.reverse, layers/actnorm.py:39-52) is folded into the packed bank.
        Returns None when the shape has no MFMA instantiation or the activations are not 16-byte aligned (the caller
        then runs the two layers one after the other)."""
        bank = self._get(weights, G, orient)
        w_canon = bank.w_canon
        _require_device(y, "input")
        B, Cq, H, W, KH, KW = _dims(y, w_canon, G)
        L = _lib.lib()
        if out is None and y.numel():
            out = torch.empty_like(y)
        if not self._packed_path_ok(L, y, out, Cq, H, W, KH, KW):
            return None
        # the shift rides on the wavefront / role-split kernels only: the big banks and the wide maps that finc_big.hip takes
        # over from the 33..64-channel banks (Cq = 50 at 256 columns) carry a scale and nothing else -> two launches there
        if not L.finc_inverse_affine_supported(B, G, Cq, H, W, KH, KW):
            return None
        key = (log_scale.data_ptr(), log_scale._version, translation.data_ptr(), translation._version)
        with torch.cuda.device(y.device):
            if bank.packed_aff is None or bank.aff_key != key:
                scale = torch.exp(log_scale.detach().float()).contiguous()
                shift = translation.detach().float().contiguous()
                if scale.numel() != G * Cq or shift.numel() != G * Cq:
                    raise ValueError("affine parameters must have one entry per channel")
                packed = torch.empty(L.finc_workspace_bytes(G, Cq, KH, KW), dtype=torch.uint8, device=y.device)
                st = L.finc_pack_inverse_weights_affine_f32(w_canon.data_ptr(), scale.data_ptr(), shift.data_ptr(), packed.data_ptr(),
                                                            G, Cq, KH, KW, _stream_ptr(y))
                if st == 3:        # FINC_ERR_UNSUPPORTED: a bank whose kernel cannot carry the shift (the big banks, finc_big.hip)
                    return None
                _lib.check(st, "finc_pack_inverse_weights_affine_f32")
                bank.packed_aff = packed
                bank.aff_key = key
            st = L.finc_inverse_packed_f32(y.data_ptr(), bank.packed_aff.data_ptr(), out.data_ptr(), B, G, Cq, H, W, KH, KW, orient,
                                           _stream_ptr(y))
            if st == 3:            # (the launch itself refuses a shift-carrying bank on a map it cannot serve)
                return None
            _lib.check(st, "finc_inverse_packed_f32")
        return out


class _FincConvFunction(torch.autograd.Function):
    """The autograd.Function underneath FastFlowUnit / PaddedConv2d.forward.  Backward applies the
    corner-tap mask in-kernel, so `model.apply(clear_grad)` (train/experiment.py:16-18) is a no-op on it.
    The stored weights are inputs (one per group, so every parameter gets its own gradient without a cat node in the
    graph); the canonical bank and the forward fragments come from the layer's PackedWeights cache."""

    @staticmethod
    def forward(ctx, x, cache, G, orient, *weights):
        x = x.contiguous()
        out = cache.forward(x, list(weights), G, orient, validate=False)
        ctx.save_for_backward(x, cache._get(list(weights), G, orient, validate=False).w_canon)   # (same entry: no check, no synchronisation)
        ctx.G, ctx.orient, ctx.nw = G, orient, len(weights)
        return out

    @staticmethod
    def backward(ctx, grad_z):
        x, w_canon = ctx.saved_tensors
        need_gw = any(ctx.needs_input_grad[4:])
        gx, gw = finc_backward(grad_z.contiguous(), x, w_canon, ctx.G, ctx.orient,
                               need_gx=ctx.needs_input_grad[0], need_gw=need_gw)
        gws = (None,) * ctx.nw
        if gw is not None:
            gw = canonicalize(gw, ctx.G, ctx.orient)  # canonical -> stored orientation
            gws = tuple(gw.chunk(ctx.nw, dim=0)) if ctx.nw > 1 else (gw,)
        return (gx, None, None, None) + gws


def conv_forward(x, weights, G, orient, cache):
    """z = forward(x) under autograd.  `weights`: the stored (state-dict form) banks of the G groups, one tensor per group or
    one tensor for all; `cache`: the layer's PackedWeights."""
    return _FincConvFunction.apply(x, cache, G, orient, *weights)
