"""GPU helper for SURVEY 8 f3 (second half): what the 1x1-conv inverse next to the unit costs as the separate launch it is
today (Conv1x1.reverse = F.conv2d with the cached inverse, layers/conv1x1.py:35-43), at the c3 shape and at the unit
shapes of the c4 stack, next to the unit inverse itself.  Prints one JSON object; keep it under profiles/."""
import json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ.setdefault("MIOPEN_FIND_MODE", "2")
import torch
from fincflow_amd import FastFlowUnit, glow

dev = torch.device("cuda:0")
torch.manual_seed(0)


def timeit(fn, n=100):
    t_end = time.perf_counter() + 0.2
    while time.perf_counter() < t_end:
        for _ in range(5):
            fn()
        torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(n):
        fn()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / n * 1e3


out = {}
# (name, B, C, H, W): c3, then the three levels of the c4 stack (split prior: C = 12 / 24 / 48 at 16x16 / 8x8 / 4x4)
for name, B, C, H, W in (("c3", 256, 96, 64, 64), ("c2", 64, 48, 32, 32), ("c4_L0", 128, 12, 16, 16),
                         ("c4_L1", 128, 24, 8, 8), ("c4_L2", 128, 48, 4, 4)):
    unit = FastFlowUnit(C, C, 3).to(dev)
    c11 = glow.Conv1x1(C).to(dev)
    an = glow.ActNorm(C).to(dev)
    an.initialized.fill_(1)
    y = torch.randn(B, C, H, W, device=dev)
    with torch.no_grad():
        c11.reverse(y); unit.reverse(y); an.reverse(y)
        t11 = timeit(lambda: c11.reverse(y))                       # the HIP mixing kernel (ops.finc_mix)
        w4 = c11._inverse_matrix().view(C, C, 1, 1)
        tmi = timeit(lambda: torch.nn.functional.conv2d(y, w4))   # what it replaces: F.conv2d through MIOpen
        tfa = timeit(lambda: c11.reverse_then_affine(y, an.log_scale, an.translation))
        tan = timeit(lambda: an.reverse(y))
        tun = timeit(lambda: unit.reverse(y))
        tfu = timeit(lambda: unit.reverse_affine(y, an.log_scale, an.translation))
        pre = {}
        ws = unit._weights()
        if unit._cache.premultiplied_supported((B, C, H, W), ws, 4, 0xE4):
            lead = unit._cache.lead_inverse(ws, 4, 0xE4)
            zp = c11.reverse_premultiplied(y, lead, an.log_scale, an.translation)
            a = unit._cache.inverse_premultiplied(zp, ws, 4, 0xE4)
            b = unit.reverse_affine(c11.reverse(y), an.log_scale, an.translation)
            pre = {"mix_with_lead_and_actnorm_us": timeit(lambda: c11.reverse_premultiplied(y, lead, an.log_scale, an.translation)),
                   "unit_reverse_premultiplied_us": timeit(lambda: unit._cache.inverse_premultiplied(zp, ws, 4, 0xE4)),
                   "reverse_step_plain_us": timeit(lambda: unit.reverse_affine(c11.reverse(y), an.log_scale, an.translation)),
                   "reverse_step_lead_folded_us": timeit(lambda: unit.reverse_after_mix(y, c11, (an.log_scale, an.translation))),
                   "rel_err_between_the_two": float((a - b).abs().max() / b.abs().max())}
    E = B * C * H * W
    out[name] = {**pre, "shape": [B, C, H, W], "conv1x1_reverse_us": t11, "conv1x1_reverse_miopen_us": tmi,
                 "conv1x1_reverse_then_actnorm_one_launch_us": tfa, "actnorm_reverse_us": tan, "unit_reverse_us": tun,
                 "unit_reverse_with_actnorm_folded_us": tfu, "conv1x1_GBps": 8 * E / t11 / 1e3,
                 "bytes_moved_by_conv1x1": 8 * E}
print(json.dumps(out, indent=1))
