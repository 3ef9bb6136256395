"""GPU helper for rocprofv3: 20 launches of the big-bank inverse at CINC C = 96, 64x64, B = 256."""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from fincflow_amd import ops, _lib
from oracle import oracle
dev = torch.device("cuda:0")
B, G, Cq, H, W = 256, 1, 96, 64, 64
ws = oracle.make_stored_weights(G, Cq, 3, 3, orient=0, seed=1, std=0.025)
wc = ops.canonicalize(torch.from_numpy(ws).to(dev), G, 0)
z = torch.randn(B, G * Cq, H, W, device=dev)
L = _lib.lib()
packed = torch.empty(L.finc_workspace_bytes(G, Cq, 3, 3), dtype=torch.uint8, device=dev)
_lib.check(L.finc_pack_inverse_weights_f32(wc.data_ptr(), packed.data_ptr(), G, Cq, 3, 3, None), "pack")
out = torch.empty_like(z)
for _ in range(20):
    _lib.check(L.finc_inverse_packed_f32(z.data_ptr(), packed.data_ptr(), out.data_ptr(), B, G, Cq, H, W, 3, 3, 0, None), "inv")
torch.cuda.synchronize()
