"""The layers either side of the hot path in the reference's CIFAR Glow stack (BASELINE
configs[3]): thin PyTorch modules with the reference's constructor arguments, parameter names and
forward -> (out, logdet[B]) / reverse -> out protocol, so that `create_model` below builds the topology of
fastflow/fastflow_cifar.py:35-63 around fincflow_amd.FastFlowUnit.  None of this is a HIP kernel: these are
per-pixel / 1x1 / small-conv ops that PyTorch-ROCm already runs; the point of this file is that a whole
sampling pass (96 units at 16x16 / 8x8 / 4x4) can run and be captured in one HIP graph.

Reference semantics followed:
  Squeeze        layers/squeeze.py:5-41          space-to-depth, channel order (c, dy, dx)
  ActNorm        layers/actnorm.py:5-66          data-dependent init on first forward; out = (x - t) * exp(-log_scale)
  Conv1x1        layers/conv1x1.py:8-49          orthogonal init, ldj = H*W*log|det W|
  Coupling       layers/coupling.py:46-113       affine, net = conv3x3-ReLU-conv1x1-ReLU-Conv2dZero, log_s = 2*tanh(h/2)
  SplitPrior     layers/splitprior.py:7-41       Coupling + factor out the second half under a standard normal
  Normalization, LogitTransform, Dequantization  layers/normalize.py, transforms.py:6-19, dequantize.py
  GaussianPrior  train/losses.py:17-45           standard-normal base (log_prob per sample, sample(n))
"""
import math

import numpy as np
import torch
import torch.nn as nn
import torch.nn.functional as F

from .layers import FastFlowUnit, FlowLayer, FlowSequential


class Squeeze(FlowLayer):
    def forward(self, input, context=None):
        b, c, h, w = input.shape
        x = input.reshape(b, c, h // 2, 2, w // 2, 2).permute(0, 1, 3, 5, 2, 4)
        return x.reshape(b, c * 4, h // 2, w // 2), self.logdet(input, context)

    def reverse(self, input, context=None):
        b, c, h, w = input.shape
        x = input.reshape(b, c // 4, 2, 2, h, w).permute(0, 1, 4, 2, 5, 3)
        return x.reshape(b, c // 4, h * 2, w * 2)

    def logdet(self, input, context=None):
        return input.new_zeros(len(input))


class ActNorm(FlowLayer):
    def __init__(self, n_dims):
        super().__init__()
        self.n_dims = n_dims
        self.translation = nn.Parameter(torch.zeros(n_dims))
        self.log_scale = nn.Parameter(torch.zeros(n_dims))
        self.register_buffer('initialized', torch.tensor(0))
        # host-side mirror of `initialized` (None: unknown, read the buffer once): the buffer lives on the device, and
        # testing it costs a device->host sync per layer and call -- 96 per pass of the CIFAR stack, and no HIP graph
        self._init_known = None

    def _is_initialized(self):
        if self._init_known is None:
            self._init_known = bool(self.initialized)      # one sync, then cached
        return self._init_known

    def _load_from_state_dict(self, *args, **kwargs):
        self._init_known = None                            # a checkpoint brings its own flag
        return super()._load_from_state_dict(*args, **kwargs)

    def mark_initialized(self):
        """Skip the data-dependent initialisation (parameters set by hand or by a checkpoint): no sync, graph-safe."""
        self.initialized.fill_(1)
        self._init_known = True

    def reset_initialization(self):
        """Ask for the data-dependent initialisation again (the next forward computes it).  Writing the `initialized` buffer
        directly is not seen by the host-side mirror once it is known."""
        self.initialized.fill_(0)
        self._init_known = False

    def _shaped(self, input):
        shape = (1, -1) + (1,) * (input.dim() - 2)
        return self.translation.view(shape), self.log_scale.view(shape)

    def forward(self, input, context=None):
        if not self._is_initialized():
            with torch.no_grad():
                dims = [d for d in range(input.dim()) if d != 1]
                self.translation.copy_(input.mean(dim=dims))
                self.log_scale.copy_(torch.log(input.std(dim=dims) + 1e-8))
                self.initialized.fill_(1)
                self._init_known = True
        t, ls = self._shaped(input)
        return (input - t) * torch.exp(-ls), self.logdet(input, context)

    def reverse(self, input, context=None):
        t, ls = self._shaped(input)
        return input * torch.exp(ls) + t

    def forward_affine_params(self):
        """forward(x) = (x - translation) * exp(-log_scale): what FlowSequential folds into the FastFlowUnit in front of
        this layer (only once the data-dependent initialisation has happened)."""
        if not self._is_initialized():
            return None
        return self.log_scale, self.translation

    def reverse_affine_params(self):
        """reverse(y) = exp(log_scale) * y + translation, per channel: what FlowSequential folds into the FastFlowUnit
        that comes next in the reverse chain."""
        return self.log_scale, self.translation

    def logdet(self, input, context=None):
        pixels = int(np.prod(input.shape[2:])) if input.dim() > 2 else 1
        return -self.log_scale.sum().expand(input.size(0)) * pixels


class Conv1x1(FlowLayer):
    """layers/conv1x1.py:8-49.  Inference / sampling (no autograd graph, device tensors, a channel count the library
    instantiates) runs on the HIP mixing kernel (`ops.finc_mix`, one streaming launch); training keeps F.conv2d."""

    def __init__(self, n_channels):
        super().__init__()
        self.n_channels = n_channels
        q = np.linalg.qr(np.random.randn(n_channels, n_channels))[0]
        self.W = nn.Parameter(torch.from_numpy(q.astype('float32')))

    def _hip(self, x):
        from . import ops
        return (not (torch.is_grad_enabled() and (self.W.requires_grad or x.requires_grad)) and x.is_cuda
                and x.dtype == torch.float32 and x.dim() == 4 and ops.mix_supported(self.n_channels))

    def forward(self, x, context=None):
        h, w = x.shape[2:]
        ldj = h * w * torch.slogdet(self.W)[1]
        if self._hip(x):
            from . import ops
            return ops.finc_mix(x.contiguous(), self.W.detach().contiguous()), ldj
        return F.conv2d(x, self.W.view(self.n_channels, self.n_channels, 1, 1)), ldj

    def _inverse_matrix(self):
        # the reference inverts W on every call (layers/conv1x1.py:37-39); cache it per weight version so that a
        # sampling pass has no LU factorisation (and no host sync) in it and can be captured in a HIP graph
        key = (self.W.data_ptr(), self.W._version, self.W.device)
        if getattr(self, "_inv_key", None) != key:
            with torch.no_grad():
                self._w_inv = torch.inverse(self.W.detach()).contiguous()
            self._inv_key = key
            self._aff_key = None
        return self._w_inv

    def reverse(self, z, context=None):
        if self._hip(z):
            from . import ops
            return ops.finc_mix(z.contiguous(), self._inverse_matrix())
        if torch.is_grad_enabled() and self.W.requires_grad:
            w_inv = torch.inverse(self.W)
        else:
            w_inv = self._inverse_matrix()
        return F.conv2d(z, w_inv.view(self.n_channels, self.n_channels, 1, 1))

    def reverse_then_affine(self, z, log_scale, translation):
        """exp(log_scale) * reverse(z) + translation in ONE launch: the ActNorm that precedes this layer in
        the model follows it in the reverse chain (layers/actnorm.py:47-52), and a per-channel affine map after a channel
        mix is a row scaling of the matrix plus a bias.  None when this call cannot take the HIP path."""
        if not self._hip(z):
            return None
        from . import ops
        w_inv = self._inverse_matrix()
        key = (log_scale.data_ptr(), log_scale._version, translation.data_ptr(), translation._version)
        if getattr(self, "_aff_key", None) != key:
            with torch.no_grad():
                self._m_aff = (torch.exp(log_scale.detach().float()).view(-1, 1) * w_inv).contiguous()
                self._b_aff = translation.detach().float().contiguous()
            self._aff_key = key
        return ops.finc_mix(z.contiguous(), self._m_aff, self._b_aff)

    def reverse_premultiplied(self, z, lead, log_scale=None, translation=None):
        """blockdiag(lead) (exp(log_scale) * reverse(z) + translation) in ONE launch: `lead` [G, Cq, Cq] is the inverse of
        the unit triangular tap of the FastFlowUnit that follows in the reverse chain, which then runs
        without its z-term.  The product with a block-diagonal matrix is folded into the mix's matrix and bias on the host,
        once per weight version.  None when this call cannot take the HIP path."""
        if not self._hip(z):
            return None
        from . import ops
        w_inv = self._inverse_matrix()
        # (`lead` and `w_inv` are compared by identity and kept alive here: both are cache entries that are REPLACED when their
        # weights change, and the address of a freed tensor is the first one the allocator hands out again)
        key = () if log_scale is None else (log_scale.data_ptr(), log_scale._version, translation.data_ptr(), translation._version)
        if (getattr(self, "_lead_key", None) != key or getattr(self, "_lead_inv", None) is not w_inv
                or getattr(self, "_lead_obj", None) is not lead):
            with torch.no_grad():
                m = w_inv.double()
                b = None
                if log_scale is not None:
                    m = torch.exp(log_scale.detach().double()).view(-1, 1) * m
                    b = translation.detach().double()
                blk = torch.block_diag(*lead.double().unbind(0))
                self._m_lead = (blk @ m).float().contiguous()
                self._b_lead = None if b is None else (blk @ b).float().contiguous()
            self._lead_key, self._lead_inv, self._lead_obj = key, w_inv, lead
        return ops.finc_mix(z.contiguous(), self._m_lead, self._b_lead)

    def logdet(self, input, context=None):
        raise NotImplementedError


class Conv2dZero(nn.Module):
    """Zero-initialised 3x3 conv with a learned per-channel log-scale (layers/coupling.py:10-43)."""

    def __init__(self, in_channels, out_channels, logscale_factor=3):
        super().__init__()
        self.logscale_factor = logscale_factor
        self.weight = nn.Parameter(torch.zeros(out_channels, in_channels, 3, 3))
        self.bias = nn.Parameter(torch.zeros(out_channels))
        self.logs = nn.Parameter(torch.zeros(out_channels))

    def forward(self, input):
        out = F.conv2d(input, self.weight, self.bias, padding=1)
        return out * torch.exp(self.logs * self.logscale_factor).view(1, -1, 1, 1)


class Coupling(FlowLayer):
    def __init__(self, input_size, width=512, n_context=None):
        super().__init__()
        self.n_channels = input_size[0]
        self.half_channels = self.n_channels // 2
        self.width = width
        self.uses_context = n_context is not None
        in_channels = self.half_channels + (n_context or 0)
        self.net = nn.Sequential(nn.Conv2d(in_channels, width, kernel_size=(3, 3), padding=(1, 1)), nn.ReLU(),
                                 nn.Conv2d(width, width, (1, 1)), nn.ReLU(),
                                 Conv2dZero(width, self.n_channels))

    def _params(self, x, context):
        assert (context is not None) == self.uses_context
        x1, x2 = x[:, :self.half_channels], x[:, self.half_channels:]
        h = self.net(x1 if context is None else torch.cat([x1, context], dim=1))
        log_s = 2.0 * torch.tanh(h[:, ::2] / 2.0)
        return x1, x2, log_s, h[:, 1::2]

    def forward(self, input, context=None):
        x1, x2, log_s, t = self._params(input, context)
        return torch.cat([x1, x2 * torch.exp(log_s) + t], dim=1), log_s.flatten(start_dim=1).sum(-1)

    def reverse(self, input, context=None):
        x1, x2, log_s, t = self._params(input, context)
        return torch.cat([x1, (x2 - t) * torch.exp(-log_s)], dim=1)

    def logdet(self, input, context=None):
        return self.forward(input, context)[1]


class GaussianPrior(nn.Module):
    """Standard-normal base with the interface of train/losses.py:17-45 (device follows the module)."""

    def __init__(self, size):
        super().__init__()
        self.size = tuple(size)
        self.dim = int(np.prod(size))
        self.register_buffer('_anchor', torch.zeros(1), persistent=False)  # not in reference checkpoints

    def log_prob(self, input, context=None, sum=True):
        z = input.reshape(-1, self.dim)
        return -0.5 * (z * z).sum(-1) - 0.5 * self.dim * math.log(2 * math.pi)

    def forward(self, input, context=None):
        return -self.log_prob(input, context).sum(-1)

    def sample(self, n_samples, context=None):
        x = torch.randn(n_samples, *self.size, device=self._anchor.device)
        return x, self.log_prob(x, context)


class SplitPrior(FlowLayer):
    def __init__(self, input_size, distribution, width=512):
        super().__init__()
        assert len(input_size) == 3
        self.n_channels = input_size[0]
        self.transform = Coupling(input_size, width=width)
        self.base = distribution((self.n_channels // 2, input_size[1], input_size[2]))

    def forward(self, input, context=None):
        x, ldj = self.transform(input, context)
        half = self.n_channels // 2
        return x[:, :half], self.base.log_prob(x[:, half:]) + ldj

    def reverse(self, input, context=None):
        x2, _ = self.base.sample(input.shape[0], context)
        return self.transform.reverse(torch.cat([input, x2], dim=1), context)

    def logdet(self, input, context=None):
        return self.forward(input, context)[1]


class Normalization(FlowLayer):
    def __init__(self, translation, scale, learnable=False):
        super().__init__()
        if learnable:
            self.translation = nn.Parameter(torch.Tensor([translation]))
            self.scale = nn.Parameter(torch.Tensor([scale]))
        else:
            self.register_buffer('translation', torch.Tensor([translation]))
            self.register_buffer('scale', torch.Tensor([scale]))

    def forward(self, input, context=None):
        return (input - self.translation) / self.scale, self.logdet(input, context)

    def reverse(self, input, context=None):
        return input * self.scale + self.translation

    def logdet(self, input, context=None):
        n, c, h, w = input.shape
        return (-c * h * w * torch.log(self.scale)).expand(n)


class LogitTransform(FlowLayer):
    def forward(self, input, context=None):
        return torch.log(input) - torch.log(1 - input), self.logdet(input, context)

    def reverse(self, input, context=None):
        return torch.sigmoid(input)

    def logdet(self, input, context=None):
        return (-torch.log(input) - torch.log(1 - input)).flatten(start_dim=1).sum(-1)


class Dequantization(FlowLayer):
    """Uniform dequantisation (layers/dequantize.py + distributions/uniform.py): forward adds U[0,1) noise."""

    def __init__(self, size=None):
        super().__init__()
        self.size = size

    def forward(self, input, context=None):
        return input + torch.rand_like(input.float()), input.new_zeros(len(input), dtype=torch.float32)

    def reverse(self, input, context=None):
        return input.floor()

    def logdet(self, input, context=None):
        raise NotImplementedError


def create_model(num_blocks=3, block_size=32, actnorm=False, split_prior=False, image_size=(3, 32, 32),
                 preprocess=True, coupling_width=512):
    """fastflow/fastflow_cifar.py:35-63: Squeeze -> [FastFlowUnit, (ActNorm), Conv1x1, Coupling] x block_size
    (-> SplitPrior) per block, under a standard-normal base."""
    size = tuple(image_size)
    layers = []
    if preprocess:
        alpha = 1e-6
        layers += [Dequantization(size), Normalization(translation=0, scale=256),
                   Normalization(translation=-alpha, scale=1 / (1 - 2 * alpha)), LogitTransform()]
    for level in range(num_blocks):
        layers.append(Squeeze())
        size = (size[0] * 4, size[1] // 2, size[2] // 2)
        for _ in range(block_size):
            layers.append(FastFlowUnit(size[0], size[0], (3, 3)))
            if actnorm:
                layers.append(ActNorm(size[0]))
            layers.append(Conv1x1(size[0]))
            layers.append(Coupling(size, width=coupling_width))
        if split_prior and level < num_blocks - 1:
            layers.append(SplitPrior(size, GaussianPrior, width=coupling_width))
            size = (size[0] // 2, size[1], size[2])
    return FlowSequential(GaussianPrior(size), *layers)
