"""Generate tests/golden/*.npz by running the REFERENCE's own CPU path.

Run in the build container only (needs /root/reference and oracle/_ref):

    python oracle/build_ref.py && python tests/golden/make_golden.py

What is imported from the reference (nothing is copied; only inputs and
outputs are saved):
  * fastflow/utils/solve_mc.py            solve, solve_parallel   (fp32, python loops)
  * fastflow/utils/fastflow_inverse/solve_parallel_mc.pyx  (built into oracle/_ref)
  * fastflow/layers/conv.py               PaddedConv2d.forward / reverse / reverse_python
  * fastflow/fastflow.py                  FastFlowUnit.forward / reverse_level1
    (the nvcc JIT `load` at fastflow.py:9-10 is replaced by a no-op: there is
    no CUDA here, and reverse_level2 is not called)

Every fixture stores: the layer's STORED weights (state-dict form,
layers/conv.py:72-79), the input x, the reference forward z, the reference
inverse of z.  The literal known-answer cases of
cuda/cinc_cuda/test_cuda_kernel.py:3-56 and fastflow/test_examples.py:6-25,54-73
are data, re-entered here, and solved with the reference's `solve`.
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
from oracle import build_ref  # noqa: E402

import torch  # noqa: E402
import torch.utils.cpp_extension  # noqa: E402

so = build_ref.build()
assert so, "oracle/_ref could not be built"
spec = importlib.util.spec_from_file_location("utils.fastflow_inverse.solve_parallel_mc", so)
ref_cy = importlib.util.module_from_spec(spec)
spec.loader.exec_module(ref_cy)
sys.modules["utils.fastflow_inverse.solve_parallel_mc"] = ref_cy

torch.utils.cpp_extension.load = lambda *a, **k: None  # no nvcc here (fastflow.py:9-10)
sys.path.insert(0, REF)
_cwd = os.getcwd()
os.chdir(tempfile.mkdtemp())  # anything the reference writes at import lands here
from utils import solve_mc as ref_solve_mc  # noqa: E402
from layers.conv import PaddedConv2d  # noqa: E402
from fastflow import FastFlowUnit  # noqa: E402
os.chdir(_cwd)

torch.set_num_threads(1)


def save(name, **arrays):
    path = os.path.join(OUT, name + ".npz")
    np.savez_compressed(path, **arrays)
    print(f"{name}: {os.path.getsize(path) / 1024:.1f} KiB")


def unit_case(name, B, C, H, W, ks, seed, python_fp32=False, heavy=1.0):
    torch.manual_seed(seed)
    unit = FastFlowUnit(C, C, ks).eval()
    if heavy != 1.0:  # trained-like weights: scale the free taps, keep the invariant
        for m in (unit.conv_tl, unit.conv_tr, unit.conv_bl, unit.conv_br):
            m.conv.weight.data = m.conv.weight.data * (1 + (heavy - 1) * m.get_mask())
    x = torch.randn(B, C, H, W)
    with torch.no_grad():
        z, ld = unit.forward(x)
        xr = unit.reverse_level1(z)  # Cython fp64 path per group (fastflow.py:57-76)
        arrays = dict(
            x=x.numpy(), z=z.numpy(), x_rev_cython=xr.numpy(), logdet=np.float32(ld),
            w_tl=unit.conv_tl.conv.weight.numpy(), w_tr=unit.conv_tr.conv.weight.numpy(),
            w_bl=unit.conv_bl.conv.weight.numpy(), w_br=unit.conv_br.conv.weight.numpy(),
            kernel_size=np.array(unit.conv_tl.kernel_size), seed=np.int64(seed))
        if python_fp32:  # fp32 python-loop path, PaddedConv2d.reverse_python (layers/conv.py:165-189)
            chunks = torch.chunk(z, 4, dim=1)
            outs = [m.reverse_python(c)[0] for m, c in
                    zip((unit.conv_tl, unit.conv_tr, unit.conv_bl, unit.conv_br), chunks)]
            arrays["x_rev_python_fp32"] = torch.cat(outs, dim=1).numpy()
    save(name, **arrays)


def padded_case(name, B, C, H, W, ks, order, seed, python_fp32=True, z_direct=False):
    torch.manual_seed(seed)
    layer = PaddedConv2d(C, C, ks, order=order).eval()
    x = torch.randn(B, C, H, W)
    with torch.no_grad():
        z, ld = layer.forward(x)
        if z_direct:  # sampling direction: the input of reverse is N(0,1), not a forward output
            z = torch.randn(B, C, H, W)
        xr, ld2 = layer.reverse(z)
        arrays = dict(x=x.numpy(), z=z.numpy(), x_rev_cython=xr.numpy(), w=layer.conv.weight.numpy(),
                      mask=layer.mask.numpy(), pad=np.array(layer.pad), order=np.array(order),
                      logdet=np.float32(ld), logdet_rev=np.float32(ld2), seed=np.int64(seed))
        if python_fp32:
            arrays["x_rev_python_fp32"] = layer.reverse_python(z)[0].numpy()
            # diagonal-order twin, canonical orientation only (utils/solve_mc.py:8-50)
            if order == "TL":
                arrays["x_rev_python_fp32_diag"] = ref_solve_mc.solve_parallel(z, layer.conv.weight.data, ks).numpy()
    save(name, **arrays)


def literal_case(name, inp, kernel, order="TL", reverse_first=True):
    """Single-channel known-answer cases.  `kernel` is given as the test wrote it
    (stored form for `order`)."""
    x = torch.tensor(inp, dtype=torch.float32)
    if x.dim() == 2:
        x = x[None]
    x = x[:, None]  # [m,1,n,n]
    k = torch.tensor(kernel, dtype=torch.float32)
    layer = PaddedConv2d(1, 1, tuple(k.shape), order=order).eval()
    layer.conv.weight.data = k.reshape(1, 1, *k.shape).clone()
    with torch.no_grad():
        if reverse_first:
            y = layer.reverse_python(x)[0]
            back = layer.forward(y)[0]
        else:
            y = layer.forward(x)[0]
            back = layer.reverse_python(y)[0]
    save(name, inp=x.numpy(), w=layer.conv.weight.detach().numpy(), order=np.array(order), out=y.numpy(), back=back.numpy(),
         reverse_first=np.bool_(reverse_first))


if __name__ == "__main__":
    # --- FastFlowUnit (4 groups TL/TR/BL/BR) --------------------------------
    unit_case("unit_c1_B2_C4_8x8_k3", 2, 4, 8, 8, 3, seed=11, python_fp32=True)      # BASELINE configs[0]
    unit_case("unit_B2_C8_6x9_k3", 2, 8, 6, 9, 3, seed=12, python_fp32=True)          # H<W
    unit_case("unit_B1_C12_16x16_k3", 1, 12, 16, 16, 3, seed=13)                       # config-4 level-1 shape (Cq=3)
    unit_case("unit_B1_C24_8x8_k3", 1, 24, 8, 8, 3, seed=14)                           # config-4 level-2 shape (Cq=6)
    unit_case("unit_B2_C48_32x32_k3", 2, 48, 32, 32, 3, seed=15)                       # configs[1] shape, 2 images
    unit_case("unit_B1_C96_16x16_k3", 1, 96, 16, 16, 3, seed=16)                       # configs[2] channels, small map
    unit_case("unit_B1_C16_12x12_k2", 1, 16, 12, 12, 2, seed=17, python_fp32=True)     # 2x2
    unit_case("unit_B1_C16_12x20_k5", 1, 16, 12, 20, 5, seed=18)                       # 5x5, H<W
    unit_case("unit_B1_C8_10x14_k3x5", 1, 8, 10, 14, (3, 5), seed=19, python_fp32=True)  # KH != KW
    unit_case("unit_B1_C64_24x24_k3_heavy", 1, 64, 24, 24, 3, seed=20, heavy=2.0)      # Cq=16, larger free taps
    # --- PaddedConv2d, every order -------------------------------------------
    for i, order in enumerate(("TL", "TR", "BL", "BR")):
        padded_case(f"padded_{order}_B2_C3_7x7_k3", 2, 3, 7, 7, (3, 3), order, seed=30 + i)
    padded_case("padded_TL_B1_C24_64x64_k3", 1, 24, 64, 64, (3, 3), "TL", seed=40, python_fp32=False)  # one configs[2] group
    padded_case("padded_BR_B1_C5_9x9_k2x3_zdirect", 1, 5, 9, 9, (2, 3), "BR", seed=41, z_direct=True)
    # --- literal known-answer cases -----------------------------------------
    sq4 = [[1, 2, 3, 4], [5, 6, 7, 8], [9, 10, 11, 12], [13, 14, 15, 16]]
    literal_case("literal_cinc_id2", sq4, [[0, 0], [0, 1]])                 # cuda/cinc_cuda/test_cuda_kernel.py:3-13
    literal_case("literal_cinc_eye2", sq4, [[1, 0], [0, 1]])                # :16-26
    literal_case("literal_cinc_batch2", [[[1, 2, 3], [5, 6, 7], [9, 10, 11]],
                                         [[12, 13, 14], [15, 16, 17], [18, 19, 20]]], [[0, 0], [0, 1]])  # :29-42
    literal_case("literal_cinc_eye3", sq4, [[1, 0, 0], [0, 1, 0], [0, 0, 1]])  # :45-56
    literal_case("literal_examples_BL", [[1, 2, 3], [4, 5, 6], [7, 8, 9]], [[0, 1], [1, 0]], order="BL",
                 reverse_first=False)                                        # fastflow/test_examples.py:6-25
    literal_case("literal_examples_TR", [[1, 2, 3], [8, 11, 6], [17, 20, 9]], [[0, 2], [1, 0]], order="TR",
                 reverse_first=True)                                         # fastflow/test_examples.py:54-73
