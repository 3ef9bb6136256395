"""ctypes binding of the C ABI declared in include/finc.h (libfinc_hip.so).

Fails loudly: a missing library is an ImportError-like RuntimeError at first use,
never a silent fallback.
"""
import ctypes
import os

HERE = os.path.dirname(os.path.abspath(__file__))
#: FINCFLOW_LIB selects another build of the library (scripts/: A/B timing of experiment builds).  It is never silent: the
#: override is announced on stderr with the library's build flags, `library_info()` reports it, and bench.py refuses to print a
#: judged line from an overridden library unless its build flags are 0 and the line says which file ran.
LIB_OVERRIDE = os.environ.get("FINCFLOW_LIB") or None
LIB_PATH = LIB_OVERRIDE or os.path.join(HERE, "libfinc_hip.so")

OK = 0
ALGO = {"auto": 0, "strict": 1, "mfma": 2}

# every symbol include/finc.h declares (tests/test_abi.py checks the list against the header)
SYMBOLS = [
    "finc_version", "finc_status_string", "finc_last_hip_error", "finc_canonicalize_weights_f32",
    "finc_check_invariant_f32", "finc_workspace_bytes", "finc_inverse_algo_for", "finc_forward_algo_for",
    "finc_inverse_f32", "finc_forward_f32", "finc_pack_inverse_weights_f32", "finc_pack_forward_weights_f32",
    "finc_inverse_packed_f32", "finc_forward_packed_f32", "finc_backward_f32", "finc_backward_workspace_bytes",
    "finc_inverse_workspace_bytes", "finc_pack_inverse_weights_affine_f32",
    "finc_canonicalize_weights_f64", "finc_inverse_f64", "finc_forward_f64",
    "finc_f64_workspace_bytes", "finc_inverse_f64_algo", "finc_forward_f64_algo",
    "finc_inverse_kernel_variant", "finc_debug_attr_table_insert", "finc_debug_inverse_table_row",
    "finc_mix_supported_f32", "finc_mix_f32", "finc_pack_forward_weights_affine_f32", "finc_debug_hlp_timeouts",
    "finc_build_flags", "finc_inverse_packed_premultiplied_f32", "finc_inverse_premultiplied_supported", "finc_clear_fault", "finc_debug_backward_variant", "finc_debug_set_forward_form",
    "finc_debug_row_chunks", "finc_debug_inverse_remainder_images",
    "finc_inverse_affine_supported", "finc_fault_pending", "finc_runtime_switches",
    "finc_debug_clock_probe_begin", "finc_debug_clock_probe_end",
]

_lib = None


class FincError(RuntimeError):
    pass


def lib():
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise FincError(
            f"{LIB_PATH} not found: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
            "(or `make -C fincflow_amd/csrc`).  fincflow_amd has no CPU / PyTorch fallback.")
    L = ctypes.CDLL(LIB_PATH)
    vp, i, u, sz = ctypes.c_void_p, ctypes.c_int, ctypes.c_uint, ctypes.c_size_t
    L.finc_version.restype = i
    L.finc_build_flags.restype = u
    L.finc_clear_fault.restype = i
    L.finc_status_string.restype = ctypes.c_char_p
    L.finc_status_string.argtypes = [i]
    L.finc_last_hip_error.restype = ctypes.c_char_p
    L.finc_canonicalize_weights_f32.argtypes = [vp, vp, i, i, i, i, u, vp]
    L.finc_check_invariant_f32.argtypes = [vp, i, i, i, i, vp]
    L.finc_workspace_bytes.restype = sz
    L.finc_workspace_bytes.argtypes = [i, i, i, i]
    L.finc_inverse_algo_for.argtypes = [i, i, i, i, i]
    L.finc_forward_algo_for.argtypes = [i, i, i, i, i]
    run = [vp, vp, vp, i, i, i, i, i, i, i, u, i, vp, sz, vp]
    L.finc_inverse_f32.argtypes = run
    L.finc_forward_f32.argtypes = run
    L.finc_pack_inverse_weights_f32.argtypes = [vp, vp, i, i, i, i, vp]
    L.finc_pack_inverse_weights_affine_f32.argtypes = [vp, vp, vp, vp, i, i, i, i, vp]
    L.finc_pack_forward_weights_f32.argtypes = [vp, vp, i, i, i, i, vp]
    L.finc_pack_forward_weights_affine_f32.argtypes = [vp, vp, vp, vp, i, i, i, i, vp]
    runp = [vp, vp, vp, i, i, i, i, i, i, i, u, vp]
    L.finc_inverse_packed_f32.argtypes = runp
    L.finc_inverse_packed_premultiplied_f32.argtypes = runp
    L.finc_inverse_premultiplied_supported.argtypes = [i, i, i, i, i, i, i]
    L.finc_inverse_affine_supported.argtypes = [i, i, i, i, i, i, i]
    L.finc_fault_pending.restype = i
    L.finc_runtime_switches.argtypes = [ctypes.c_char_p, sz]
    L.finc_debug_clock_probe_begin.argtypes = [i, i]
    L.finc_debug_clock_probe_end.argtypes = [ctypes.POINTER(ctypes.c_double)]
    L.finc_forward_packed_f32.argtypes = runp
    L.finc_backward_workspace_bytes.restype = sz
    L.finc_inverse_workspace_bytes.restype = sz
    L.finc_inverse_workspace_bytes.argtypes = [i, i, i, i, i, i, i]
    L.finc_backward_workspace_bytes.argtypes = [i, i, i, i, i, i, i]
    L.finc_backward_f32.argtypes = [vp, vp, vp, vp, vp, i, i, i, i, i, i, i, u, vp, sz, vp]
    L.finc_canonicalize_weights_f64.argtypes = [vp, vp, i, i, i, i, u, vp]
    run64 = [vp, vp, vp, i, i, i, i, i, i, i, u, vp]
    L.finc_inverse_f64.argtypes = run64
    L.finc_forward_f64.argtypes = run64
    L.finc_f64_workspace_bytes.restype = sz
    L.finc_f64_workspace_bytes.argtypes = [i, i, i, i]
    L.finc_inverse_f64_algo.argtypes = run
    L.finc_forward_f64_algo.argtypes = run
    L.finc_inverse_kernel_variant.argtypes = [i, i, i, i, i, i, i, ctypes.POINTER(ctypes.c_int)]
    L.finc_debug_attr_table_insert.argtypes = [i, sz]
    L.finc_debug_inverse_table_row.argtypes = [i, ctypes.POINTER(ctypes.c_int)]
    L.finc_debug_hlp_timeouts.argtypes = [ctypes.POINTER(ctypes.c_uint)]
    L.finc_debug_backward_variant.argtypes = [i, i, i, i, i, i, i, ctypes.POINTER(ctypes.c_int)]
    L.finc_debug_inverse_remainder_images.argtypes = [i, i, i, i, i, i, i]
    L.finc_debug_row_chunks.argtypes = [ctypes.c_longlong, ctypes.c_longlong, i, i, i, i]
    L.finc_debug_set_forward_form.argtypes = [i]
    L.finc_mix_supported_f32.argtypes = [i]
    L.finc_mix_f32.argtypes = [vp, vp, vp, vp, i, i, i, vp]
    for name in SYMBOLS:
        getattr(L, name)  # AttributeError here = header and library out of sync
    _lib = L
    if LIB_OVERRIDE:
        import sys
        print(f"fincflow_amd: FINCFLOW_LIB override -> {LIB_PATH} (finc_build_flags = {int(L.finc_build_flags()):#x})", file=sys.stderr)
    return L


def library_info():
    """Which library file is loaded, whether the environment chose it, and the measurement knobs it was built with."""
    return {"path": os.path.relpath(LIB_PATH, os.path.dirname(HERE)) if not LIB_OVERRIDE else LIB_PATH,
            "env_override": bool(LIB_OVERRIDE), "build_flags": int(lib().finc_build_flags()), "version": int(lib().finc_version())}


def inverse_variant(B, G, Cq, H, W, KH, KW):
    """The MFMA inverse kernel variant the library launches for this problem, or None (strict kernel)."""
    info = (ctypes.c_int * 8)()
    st = lib().finc_inverse_kernel_variant(B, G, Cq, H, W, KH, KW, info)
    if st == 3:
        return None
    check(st, "finc_inverse_kernel_variant")
    keys = ("cqp", "nw", "npw", "sec", "lds_bytes", "workgroups", "row", "rows")
    return dict(zip(keys, list(info)))


def inverse_remainder_images(B, G, Cq, H, W, KH, KW):
    """Images of an inverse call that go to a second launch on the remainder's own kernel (0: one launch)."""
    return int(lib().finc_debug_inverse_remainder_images(B, G, Cq, H, W, KH, KW))


def backward_variant(B, G, Cq, H, W, KH, KW):
    """Kernels finc_backward_f32 runs for this shape: grad-weight form (0 direct, 1 dword MFMA, 2 staged, 3 tiled), grad-input
    waves per strip (0 = direct kernel, > 1 = K-split), whether grad-input takes the staged form, and `conv_form`: the kernel
    the forward and grad-input run ("scalar", "strip", "strip16" = 16-byte pieces, "winograd" = F(2,3) along W, "msplit" = the
    big banks' M-split over eight waves, "winograd4" = F(4,3) along W, "winograd25" = F(2,5) along W for the 5x5 banks, "winograd4m" = F(4,3) M-split over a workgroup's waves
    for the 3x3 banks of 28 .. 64 channels, "stream" = the streaming-bank kernel of the banks beyond every table, finc_stream.hip)."""
    info = (ctypes.c_int * 3)()
    check(lib().finc_debug_backward_variant(B, G, Cq, H, W, KH, KW, info), "finc_debug_backward_variant")
    form = "scalar" if info[1] == 0 else ("strip", "strip16", "winograd", "msplit", "winograd4", "winograd25", "winograd4m", "stream")[info[2]]
    return {"gradw": ("direct", "dword", "staged", "tiled", "winograd", "winograd_tiled")[info[0]], "gradx_waves": info[1], "gradx_staged": bool(info[2]),
            "conv_form": form}


def set_forward_form(form):
    """Pin the 3x3 forward / grad-input kernel family for this process: 0 library's choice, 1 strip kernel, 2 Winograd F(2,3),
    4 Winograd F(4,3) (tests, A/B timing)."""
    check(lib().finc_debug_set_forward_form(int(form)), "finc_debug_set_forward_form")


def build_flags():
    """Measurement knobs the library was built with (0 = product build)."""
    return int(lib().finc_build_flags())


def runtime_switches():
    """FINC_* environment switches the library found set, plus a pinned forward form: [] = the library's own dispatch."""
    buf = ctypes.create_string_buffer(1024)
    n = lib().finc_runtime_switches(buf, len(buf))
    return [t for t in buf.value.decode().split(",") if t] if n else []


def clock_probe_begin(period_us=250, max_ms=3000):
    """Start the one-wave shader-clock probe on the current device (include/finc.h).  No device-wide synchronisation until
    clock_probe_end()."""
    check(lib().finc_debug_clock_probe_begin(int(period_us), int(max_ms)), "finc_debug_clock_probe_begin")


def clock_probe_end():
    st = (ctypes.c_double * 6)()
    check(lib().finc_debug_clock_probe_end(st), "finc_debug_clock_probe_end")
    return {"mean_mhz": st[0], "min_mhz": st[1], "max_mhz": st[2], "samples": int(st[3]), "seconds": st[4], "median_mhz": st[5]}


def fault_pending():
    """Has a helper-wave wait of an earlier launch on the current device given up (its output is garbage)?  Host-side read."""
    return bool(lib().finc_fault_pending())


def raise_if_faulted(where):
    if fault_pending():
        raise FincError(f"{where}: a helper-wave wait of an earlier launch on this device gave up -- its output is not valid "
                        "(fincflow_amd._lib.clear_fault() resets)")


def clear_fault():
    check(lib().finc_clear_fault(), "finc_clear_fault")


def hlp_timeouts():
    """Waits of the helper-wave inverse that gave up (must be 0)."""
    n = ctypes.c_uint(0)
    check(lib().finc_debug_hlp_timeouts(ctypes.byref(n)), "finc_debug_hlp_timeouts")
    return int(n.value)


def inverse_table():
    """Every row of the MFMA inverse instantiation table (host-only call)."""
    rows, r = [], 0
    info = (ctypes.c_int * 6)()
    while lib().finc_debug_inverse_table_row(r, info) == OK:
        rows.append(dict(zip(("cqp", "kh", "kw", "nw", "npw", "max_problems"), list(info))))
        r += 1
    return rows


def check(status, what):
    if status != OK:
        L = lib()
        msg = L.finc_status_string(status).decode()
        if status == 5:
            msg += ": " + L.finc_last_hip_error().decode()
        if status in (1, 2, 7):
            raise ValueError(f"{what}: {msg}")
        raise FincError(f"{what}: {msg}")
