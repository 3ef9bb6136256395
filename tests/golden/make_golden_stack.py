"""Golden trace of a config-4 style stack through the REFERENCE's own layers (build container only):

    python oracle/build_ref.py && python tests/golden/make_golden_stack.py

Imports from the reference: layers/squeeze.py, actnorm.py, conv1x1.py, coupling.py and fastflow.py::FastFlowUnit
(CPU paths: forward via F.conv2d, reverse via reverse_level1 = the Cython solver).  The stack topology and the
deterministic parameter fill live in tests/helpers.py (STACK_SPEC, fill_stack_parameters); only the input, the
FastFlowUnit weights (they carry the unit-triangular invariant) and the reference's outputs are stored.
"""
import importlib.util
import os
import sys
import tempfile

import numpy as np

REPO = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
OUT = os.path.dirname(os.path.abspath(__file__))
REF = "/root/reference/fastflow"
os.environ.setdefault("MPLBACKEND", "Agg")
sys.path.insert(0, REPO)
sys.path.insert(0, os.path.join(REPO, "tests"))
from oracle import build_ref  # noqa: E402
from helpers import STACK_INPUT, STACK_SPEC, fill_stack_parameters  # noqa: E402

import torch  # noqa: E402
import torch.utils.cpp_extension  # noqa: E402

so = build_ref.build()
spec = importlib.util.spec_from_file_location("utils.fastflow_inverse.solve_parallel_mc", so)
ref_cy = importlib.util.module_from_spec(spec)
spec.loader.exec_module(ref_cy)
sys.modules["utils.fastflow_inverse.solve_parallel_mc"] = ref_cy
torch.utils.cpp_extension.load = lambda *a, **k: None
sys.path.insert(0, REF)
_cwd = os.getcwd()
os.chdir(tempfile.mkdtemp())
from layers.squeeze import Squeeze  # noqa: E402
from layers.actnorm import ActNorm  # noqa: E402
from layers.conv1x1 import Conv1x1  # noqa: E402
from layers.coupling import Coupling  # noqa: E402
from fastflow import FastFlowUnit  # noqa: E402
os.chdir(_cwd)
torch.set_num_threads(1)


def build():
    layers = []
    for s in STACK_SPEC:
        if s[0] == "squeeze":
            layers.append(Squeeze())
        elif s[0] == "ffu":
            layers.append(FastFlowUnit(s[1], s[1], (s[2], s[2])))
        elif s[0] == "actnorm":
            layers.append(ActNorm(s[1]))
        elif s[0] == "conv1x1":
            layers.append(Conv1x1(s[1]))
        elif s[0] == "coupling":
            layers.append(Coupling(s[1], width=s[2]))
    return layers


if __name__ == "__main__":
    torch.manual_seed(4242)
    np.random.seed(4242)
    layers = build()
    fill_stack_parameters(layers)
    arrays = {}
    for idx, (s, m) in enumerate(zip(STACK_SPEC, layers)):
        if s[0] == "ffu":
            for o in ("tl", "tr", "bl", "br"):
                arrays[f"L{idx}.{o}"] = getattr(m, f"conv_{o}").conv.weight.detach().numpy().copy()
    x = torch.randn(*STACK_INPUT)
    with torch.no_grad():
        h, logdet = x, 0
        for m in layers:                                   # FlowSequential.forward, layers/flowsequential.py:21-44
            h, ld = m(h, None)
            logdet = logdet + ld
        z_in = torch.randn_like(h)                         # FlowSequential.sample walk, :89-100
        r = z_in
        for m in reversed(layers):
            r = m.reverse_level1(r) if isinstance(m, FastFlowUnit) else m.reverse(r, None)
            r = r[0] if isinstance(r, tuple) else r
    arrays.update(x=x.numpy(), z=h.numpy(), logdet=np.asarray(logdet, dtype=np.float32) * np.ones(x.shape[0], np.float32),
                  z_in=z_in.numpy(), x_rev=r.numpy())
    path = os.path.join(OUT, "stack_c4_small.npz")
    np.savez_compressed(path, **arrays)
    print("stack_c4_small:", os.path.getsize(path) // 1024, "KiB", "logdet", arrays["logdet"])
