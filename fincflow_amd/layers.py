"""Drop-in modules for the reference's hot-path layers.

Same constructor signatures, attribute names, state-dict keys and
forward -> (out, logdet) / reverse protocol as

  * fastflow/layers/flowlayer.py:8-30      FlowLayer (ABC)
  * fastflow/layers/conv.py:21-221          PaddedConv2d
  * fastflow/fastflow.py:13-100             FastFlowUnit

so a reference `FlowSequential` (layers/flowsequential.py:21-44,89-115) or
`FastFlowStep` (fastflow_cifar_multi_gpu.py:224-256) can hold them unchanged.
The arithmetic runs in libfinc_hip.so; there is no CPU path.
"""
from abc import ABCMeta, abstractmethod

import torch
import torch.nn as nn

from . import _lib, ops


class FlowLayer(nn.Module, metaclass=ABCMeta):
    """layers/flowlayer.py:8-30."""

    @abstractmethod
    def forward(self, input, context=None):
        pass

    @abstractmethod
    def reverse(self, input, context=None):
        pass

    @abstractmethod
    def logdet(self, input, context=None):
        pass

    def reconstruct_forward(self, input, context=None):
        return self.forward(input)

    def reconstruct_reverse(self, input, context=None):
        return self.reverse(input)


class PaddedConv2d(FlowLayer):
    """One-corner zero-padded, bias-free, unit-triangular conv (layers/conv.py:21-221).

    `self.conv.weight` is stored flipped per `order` exactly like the reference
    (layers/conv.py:72-79), so reference checkpoints load with `load_state_dict`.
    """

    def __init__(self, in_channels, out_channels, kernel_size, bias=False, order='TL'):
        super().__init__()
        assert len(kernel_size) == 2
        assert order in {'TL', 'TR', 'BL', 'BR'}, 'unknown order: {}'.format(order)
        if in_channels != out_channels:
            raise ValueError("an invertible conv needs in_channels == out_channels")
        self.kernel_size = kernel_size
        self.order = order
        K_H, K_W = kernel_size[0], kernel_size[1]
        # (left, right, top, bottom), layers/conv.py:41-55
        self.pad = {'TL': (K_W - 1, 0, K_H - 1, 0), 'TR': (0, K_W - 1, K_H - 1, 0),
                    'BL': (K_W - 1, 0, 0, K_H - 1), 'BR': (0, K_W - 1, 0, K_H - 1)}[order]
        # Parameter holder only, and bias-free WHATEVER the argument says -- exactly as the reference builds it
        # (layers/conv.py:60 hard-codes bias=False), so the state dict has the one key `conv.weight` and reference
        # checkpoints load strictly.  The reference then carries a dead `conv.bias is not None` branch in its reverse
        # (layers/conv.py:113-117); here `bias=True` makes that branch live through a parameter of the layer's OWN,
        # `bias` (zero-initialised, as the reference's reset would leave a bias): conv(x) + b, and the reverse subtracts b
        # first -- a per-channel add around the same two launches.  The FInC unit never sets it (fastflow.py:24-27).
        self.conv = nn.Conv2d(in_channels, out_channels, kernel_size, bias=False)
        self.bias = nn.Parameter(torch.zeros(out_channels)) if bias else None
        self._cache = ops.PackedWeights()
        self.reset_parameters()

    @property
    def _orient(self):
        return ops.ORDER_BITS[self.order]

    def _flip(self, t):
        if self.order == 'TR':
            return torch.flip(t, [3])
        if self.order == 'BL':
            return torch.flip(t, [2])
        if self.order == 'BR':
            return torch.flip(t, [2, 3])
        return t

    def reset_parameters(self):
        """layers/conv.py:63-79."""
        nn.init.normal_(self.conv.weight.data, mean=0.0, std=0.05)
        self.mask = self.get_mask()
        w = self.conv.weight.data
        for c_out in range(w.shape[0]):
            w[c_out, c_out, -1, -1] = 1.0
            w[c_out, c_out + 1:, -1, -1] = 0.0
        self.conv.weight.data = self._flip(w).contiguous()

    def get_mask(self):
        """layers/conv.py:81-96."""
        mask = torch.ones_like(self.conv.weight.data)
        for c_out in range(mask.shape[0]):
            mask[c_out, c_out:, -1, -1] = 0.0
        return self._flip(mask).contiguous()

    def reset_gradients(self):
        """layers/conv.py:98-99.  The HIP backward already writes masked gradients; kept for the runner's
        `model.apply(clear_grad)` (train/experiment.py:16-18)."""
        if self.conv.weight.grad is not None:
            self.conv.weight.grad = self.conv.weight.grad * self.mask.to(self.conv.weight.grad.device)

    def forward(self, x, context=None, compute_expensive=None):
        if torch.is_grad_enabled() and (x.requires_grad or self.conv.weight.requires_grad):
            out = ops.conv_forward(x, [self.conv.weight], 1, self._orient, self._cache)
        else:  # density evaluation / sampling checks: cached fragments, one launch
            out = self._cache.forward(x.contiguous(), [self.conv.weight], 1, self._orient)
        b = self.bias if self.bias is not None else self.conv.bias          # (conv.bias: set by hand, the reference's dead branch)
        if b is not None:
            out = out + b.view(1, -1, 1, 1)
        return out, 0.0

    def reverse(self, x, context=None, compute_expensive=None):
        with torch.no_grad():
            b = self.bias if self.bias is not None else self.conv.bias
            if b is not None:
                x = x - b.reshape(-1, x.shape[1], 1, 1)
            y = self._cache.inverse(x.contiguous(), [self.conv.weight], 1, self._orient)
        return y, 0

    def logdet(self, x=None, context=None):
        return 0.0


class FastFlowUnit(nn.Module):
    """Four PaddedConv2d (TL, TR, BL, BR), one per channel quarter (fastflow.py:13-100), evaluated as ONE
    grouped launch in each direction."""

    def __init__(self, in_channels, out_channels, kernel_size):
        super().__init__()
        if isinstance(kernel_size, int) or len(kernel_size) == 1:
            kernel_size = (kernel_size, kernel_size) if isinstance(kernel_size, int) else (kernel_size[0],) * 2
        assert in_channels % 4 == 0, "Input channels have to be a multiple of 4"
        out_channels = in_channels // 4  # fastflow.py:21: the argument is overridden
        self.conv_tl = PaddedConv2d(out_channels, out_channels, kernel_size, order='TL')
        self.conv_tr = PaddedConv2d(out_channels, out_channels, kernel_size, order='TR')
        self.conv_bl = PaddedConv2d(out_channels, out_channels, kernel_size, order='BL')
        self.conv_br = PaddedConv2d(out_channels, out_channels, kernel_size, order='BR')
        self._cache = ops.PackedWeights()

    def _weights(self):
        return [self.conv_tl.conv.weight, self.conv_tr.conv.weight, self.conv_bl.conv.weight,
                self.conv_br.conv.weight]

    def forward(self, x, context=None):
        if torch.is_grad_enabled() and (x.requires_grad or any(w.requires_grad for w in self._weights())):
            out = ops.conv_forward(x, self._weights(), 4, ops.ORIENT_FASTFLOW, self._cache)
        else:  # density evaluation / sampling checks: cached fragments, one launch
            out = self._cache.forward(x.contiguous(), self._weights(), 4, ops.ORIENT_FASTFLOW)
        return out, 0.0

    def reverse(self, x, context=None):
        return self.reverse_level2(x)

    def reverse_level2(self, x):
        """fastflow.py:78-100 without the 6 flip/cat/zeros copies: the kernel indexes the flipped pixel."""
        with torch.no_grad():
            return self._cache.inverse(x.contiguous(), self._weights(), 4, ops.ORIENT_FASTFLOW)

    def reverse_affine(self, y, log_scale, translation):
        """reverse(exp(log_scale) * y + translation): the ActNorm that follows the unit in the model precedes it in the
        reverse chain, and its affine map rides in the inverse's filter bank at no cost per step.
        Returns None if this shape cannot take the fused path."""
        with torch.no_grad():
            return self._cache.inverse_affine(y.contiguous(), self._weights(), 4, ops.ORIENT_FASTFLOW, log_scale,
                                              translation)

    def reverse_after_mix(self, u, mix, affine=None):
        """reverse(affine(mix.reverse(u))) in TWO launches that cost less than the plain two: the channel mix
        in front of the unit in the reverse chain (Conv1x1.reverse, with the ActNorm between them as a row scaling and a
        bias) also applies blockdiag(Linv_g) of this unit -- the same dense C x C product per pixel -- and the inverse
        runs without its z-term.  `mix` exposes `reverse_premultiplied(u, lead, log_scale, translation)`.  None when this
        call cannot take that path (shape without the kernel, autograd on, CPU tensors)."""
        if torch.is_grad_enabled() and (u.requires_grad or any(w.requires_grad for w in self._weights())):
            return None
        if not u.is_cuda or u.dtype != torch.float32 or u.dim() != 4:
            return None
        with torch.no_grad():
            if not self._cache.premultiplied_supported(tuple(u.shape), self._weights(), 4, ops.ORIENT_FASTFLOW):
                return None
            lead = self._cache.lead_inverse(self._weights(), 4, ops.ORIENT_FASTFLOW)
            zp = mix.reverse_premultiplied(u, lead, *(affine if affine is not None else (None, None)))
            if zp is None:
                return None
            return self._cache.inverse_premultiplied(zp, self._weights(), 4, ops.ORIENT_FASTFLOW)

    def forward_affine(self, x, log_scale, translation):
        """(forward(x) - translation) * exp(-log_scale): the ActNorm behind the unit rides in the forward bank (SURVEY 8
        f3, forward direction).  Inference only; None if this call cannot take the fused path."""
        if torch.is_grad_enabled() and (x.requires_grad or any(w.requires_grad for w in self._weights())):
            return None
        with torch.no_grad():
            return self._cache.forward_affine(x.contiguous(), self._weights(), 4, ops.ORIENT_FASTFLOW, log_scale,
                                              translation)

    def reverse_level1(self, x):
        """fastflow.py:57-76: one solve per group."""
        chunks = torch.chunk(x, 4, dim=1)
        outs = [m.reverse(c.contiguous())[0] for m, c in
                zip((self.conv_tl, self.conv_tr, self.conv_bl, self.conv_br), chunks)]
        return torch.cat(outs, dim=1)


class CINCFlowUnit(nn.Module):
    """The groups=1 (CInC) unit of cinc_flow.py:9-80: ONE top-left PaddedConv2d over all channels
    (`out_channels` is overridden with `in_channels`, cinc_flow.py:17), state-dict key `conv_tl.conv.weight`."""

    def __init__(self, in_channels, out_channels, kernel_size):
        super().__init__()
        if isinstance(kernel_size, int) or len(kernel_size) == 1:
            kernel_size = (kernel_size, kernel_size) if isinstance(kernel_size, int) else (kernel_size[0],) * 2
        out_channels = in_channels
        self.conv_tl = PaddedConv2d(out_channels, out_channels, kernel_size, order='TL')

    def forward(self, x, context=None):
        out, logdet = self.conv_tl.forward(x)
        return out, 0.0 + logdet

    def reverse(self, x, context=None):
        return self.reverse_level1(x)

    def reverse_level1(self, x):
        return self.conv_tl.reverse(x)[0]


def load_reference_checkpoint(model, checkpoint, strict=True, validate=True, trust_pickle=False):
    """Load a reference training checkpoint (train/experiment.py:400-427: a dict holding 'model_state_dict',
    written by `torch.save`) or a bare state-dict into `model`.

    The reference stores every PaddedConv2d weight flipped per its `order` (layers/conv.py:72-79) under
    `<prefix>.conv_{tl,tr,bl,br}.conv.weight`; the modules here keep the same storage and the same keys, so
    this is `load_state_dict` plus what a checkpoint written by `nn.DataParallel` needs (the 'module.' prefix,
    fastflow_cifar_multi_gpu.py wraps the model) and what trained weights need: every packed-fragment cache is
    dropped and, with `validate`, each layer's unit-triangular corner tap is checked once on the device
    (a violated invariant raises RuntimeError instead of silently solving a different system).
    A path is read with `torch.load(weights_only=True)` (tensors and plain containers only).  The reference's files
    also pickle its config dict; if that holds arbitrary objects the safe load fails, and the caller opts in to full
    unpickling -- which executes code from the file -- with `trust_pickle=True`.
    Returns the (missing_keys, unexpected_keys) of `load_state_dict`.
    """
    if isinstance(checkpoint, (str, bytes)) or hasattr(checkpoint, "__fspath__"):
        try:
            checkpoint = torch.load(checkpoint, map_location="cpu", weights_only=True)
        except Exception as e:
            if not trust_pickle:
                raise RuntimeError(f"checkpoint {checkpoint!r} needs full unpickling ({type(e).__name__}); pass "
                                   "trust_pickle=True only for files you trust") from e
            checkpoint = torch.load(checkpoint, map_location="cpu", weights_only=False)
    state = checkpoint.get("model_state_dict", checkpoint) if isinstance(checkpoint, dict) else checkpoint
    if state and all(k.startswith("module.") for k in state):
        state = {k[len("module."):]: v for k, v in state.items()}
    # a PaddedConv2d built with bias=True owns a `bias` the reference never wrote (its conv is bias-free, layers/conv.py:60):
    # that key may be missing from a reference checkpoint -- the zero initialisation stands -- everything else is strict
    own_bias = {name + ".bias" for name, m in model.named_modules() if isinstance(m, PaddedConv2d) and m.bias is not None}
    own_bias |= {"bias"} if isinstance(model, PaddedConv2d) and model.bias is not None else set()
    result = model.load_state_dict(state, strict=False)
    missing = [k for k in result.missing_keys if k not in own_bias]
    if strict and (missing or result.unexpected_keys):
        raise RuntimeError(f"checkpoint does not match the model: missing {missing}, unexpected {list(result.unexpected_keys)}")
    if any(p.is_cuda for p in model.parameters()):
        _lib.raise_if_faulted("load_reference_checkpoint")   # (the invariant check below synchronises: a natural reporting point)
    for m in model.modules():
        cache = getattr(m, "_cache", None)
        if isinstance(cache, ops.PackedWeights):
            cache.invalidate()
        if validate and isinstance(m, PaddedConv2d):
            w = m.conv.weight.detach()
            if w.is_cuda:
                ops.check_invariant(ops.canonicalize(w.contiguous(), 1, m._orient), 1)
            else:  # host-side check of the same rule, for a model validated before .cuda()
                corner = m._flip(w)[:, :, -1, -1]
                if not torch.equal(torch.triu(corner), torch.eye(corner.shape[0], dtype=corner.dtype)):
                    raise RuntimeError("checkpoint violates the unit-triangular corner tap (layers/conv.py:63-70)")
    return result


class StandardNormal(nn.Module):
    """Base density with the runner's interface (train/losses.py:17-45): log_prob(z) -> [B], sample(n)."""

    def __init__(self, size):
        super().__init__()
        self.size = tuple(size)
        self.register_buffer("_anchor", torch.zeros(1), persistent=False)  # not in reference checkpoints

    def log_prob(self, z, context=None):
        return -0.5 * (z ** 2 + torch.log(torch.tensor(2 * torch.pi, device=z.device))).flatten(1).sum(1)

    def sample(self, n_samples, context=None):
        return torch.randn(n_samples, *self.size, device=self._anchor.device), None


class FlowSequential(nn.Module):
    """Protocol-compatible stand-in for layers/flowsequential.py:9-115 (the reference file cannot travel to
    the GPU box and hard-imports wandb + a cuDNN-only extension).  Only what the hot path's callers use."""

    def __init__(self, base_distribution, *modules):
        super().__init__()
        self.base_distribution = base_distribution
        for i, module in enumerate(modules):
            self.add_module(str(i), module)
        self.sequence_modules = modules

    def __iter__(self):
        yield from self.sequence_modules

    def forward(self, input, context=None, compute_expensive=False):
        logdet = 0
        mods = list(self.sequence_modules)
        i = 0
        while i < len(mods):
            module = mods[i]
            # density evaluation: a FastFlowUnit followed by an (initialised) per-channel affine layer is one launch
            if (self.fuse_affine and i + 1 < len(mods) and isinstance(module, FastFlowUnit)
                    and hasattr(mods[i + 1], "forward_affine_params") and not torch.is_grad_enabled()):
                params = mods[i + 1].forward_affine_params()
                fused = module.forward_affine(input, *params) if params is not None else None
                if fused is not None:
                    logdet += mods[i + 1].logdet(input, context)      # (the unit's own logdet is 0)
                    input = output = fused
                    i += 2
                    continue
            output, layer_logdet = module(input, context)
            logdet += layer_logdet
            input = output
            i += 1
        logprob = self.base_distribution.log_prob(input)
        return output, logprob + logdet

    def log_prob(self, input, context=None, compute_expensive=True):
        return self.forward(input, context, compute_expensive)[1]

    #: fold a per-channel affine layer (a module exposing `reverse_affine_params()`, e.g. glow.ActNorm) into the
    #: FastFlowUnit that follows it in the reverse chain: one launch instead of two, same result
    fuse_affine = True
    #: let the channel mix in front of a FastFlowUnit (reverse chain) apply the unit's blockdiag(Linv) as well
    fuse_lead = True

    def _reverse_chain(self, input, context, fuse=None):
        fuse = self.fuse_affine if fuse is None else fuse
        mods = list(reversed(self.sequence_modules))
        i = 0
        while i < len(mods):
            m = mods[i]
            if (fuse and i + 1 < len(mods) and hasattr(m, "reverse_affine_params")
                    and isinstance(mods[i + 1], FastFlowUnit) and not torch.is_grad_enabled()):
                fused = mods[i + 1].reverse_affine(input, *m.reverse_affine_params())
                if fused is not None:
                    input = fused
                    i += 2
                    continue
            # channel mix -> [per-channel affine] -> unit: the mix also applies the unit's blockdiag(Linv), the unit's inverse
            # runs without its z-term (full-chip problem sets only; otherwise the folds below)
            if fuse and self.fuse_lead and hasattr(m, "reverse_premultiplied") and not torch.is_grad_enabled():
                j, affine = i + 1, None
                if j < len(mods) and hasattr(mods[j], "reverse_affine_params"):
                    affine = mods[j].reverse_affine_params()
                    j += 1
                if j < len(mods) and isinstance(mods[j], FastFlowUnit) and (j == i + 1 or affine is not None):
                    fused = mods[j].reverse_after_mix(input, m, affine)
                    if fused is not None:
                        input = fused
                        i = j + 1
                        continue
            # a channel mix followed by a per-channel affine layer that no FastFlowUnit takes: one mixing launch
            if (fuse and i + 1 < len(mods) and hasattr(m, "reverse_then_affine") and hasattr(mods[i + 1], "reverse_affine_params")
                    and not (i + 2 < len(mods) and isinstance(mods[i + 2], FastFlowUnit)) and not torch.is_grad_enabled()):
                fused = m.reverse_then_affine(input, *mods[i + 1].reverse_affine_params())
                if fused is not None:
                    input = fused
                    i += 2
                    continue
            output = m.reverse(input, context)
            input = output[0] if isinstance(output, tuple) else output
            i += 1
        return input

    def sample(self, n_samples, context=None, compute_expensive=False, also_true_inverse=False):
        """layers/flowsequential.py:89-115: returns `(input, input_true)` -- the runner unpacks two values
        (train/experiment.py:311-335).  No layer of the hot path is a ModifiedGradFlowLayer, so the "true inverse"
        differs from the regular sample only in how it is evaluated: it re-runs the reverse chain from the same z
        layer by layer (no affine fold), an independent check of the fused launch.  Without `also_true_inverse` (or with
        `compute_expensive`) the second value IS the first, as in the reference (`input_true = input`)."""
        z, _ = self.base_distribution.sample(n_samples, context)
        x = self._reverse_chain(z, context)
        if not compute_expensive and also_true_inverse:
            x_true = self._reverse_chain(z, context, fuse=False)
        else:
            x_true = x
        # A helper-wave launch whose protocol wait gave up has returned FINC_OK (launches are asynchronous); every later
        # launching call on the device refuses, but a chain that ENDS on such a launch would hand its garbage out.  The fault
        # word lives in host memory: reading it here costs nothing and catches every launch that has finished by now (the
        # caller's own synchronisation + the next call catch the rest).
        if x.is_cuda:
            _lib.raise_if_faulted("FlowSequential.sample")
        return x, x_true

    def reconstruct(self, input, context=None, compute_expensive=False):
        z = self.forward(input, context)[0]
        return self._reverse_chain(z, context)
