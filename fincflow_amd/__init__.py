"""fincflow_amd -- MI355X-native hot path of FInC Flow (invertible k x k convolution).

Host side is Python on PyTorch-ROCm (device memory, streams, torch.distributed);
the arithmetic is hand-written CDNA4 HIP behind the C ABI in include/finc.h
(fincflow_amd/libfinc_hip.so).  There is NO CPU fallback: every op raises if the
library is missing or a tensor is not on a ROCm device.
"""
from . import ops  # noqa: F401
from .layers import (CINCFlowUnit, FastFlowUnit, FlowLayer, FlowSequential, PaddedConv2d,  # noqa: F401
                     load_reference_checkpoint)
from .ops import finc_forward, finc_inverse, inverse  # noqa: F401

__all__ = ["FastFlowUnit", "CINCFlowUnit", "load_reference_checkpoint", "PaddedConv2d", "FlowLayer", "FlowSequential", "finc_forward", "finc_inverse", "inverse",
           "ops"]
