"""Compile the reference's own Cython CPU solver into oracle/_ref/.

TEST INFRASTRUCTURE ONLY (see finc_oracle.c).  The source is read where it
lies under /root/reference (fastflow/utils/fastflow_inverse/solve_parallel_mc.pyx);
nothing is copied into the repository and every output (generated C, .so) goes
to oracle/_ref/, which is git-ignored.  The checked-in generated C next to the
.pyx (Cython 0.29.28) does not compile against this image's numpy 2.2 headers
(`PyArray_Descr` has no member `subarray`), so the .pyx is re-cythonized with
the image's Cython -- the same step the reference's setup.py performs
(fastflow/utils/fastflow_inverse/setup.py:1-5), plus the numpy include dir the
reference omits.

When /root/reference is absent (the GPU box) this is a no-op, and load()
returns None there: oracle/_ref/ is excluded from the gpurun snapshot
(.gpurunignore) -- the compiled reference runs in the build container only.
"""
import os
import subprocess
import sys
import sysconfig

HERE = os.path.dirname(os.path.abspath(__file__))
REF_PYX = "/root/reference/fastflow/utils/fastflow_inverse/solve_parallel_mc.pyx"
OUT_DIR = os.path.join(HERE, "_ref")


def ref_so_path():
    return os.path.join(OUT_DIR, "solve_parallel_mc" + sysconfig.get_config_var("EXT_SUFFIX"))


def build(verbose=False):
    so = ref_so_path()
    if not os.path.exists(REF_PYX):
        return so if os.path.exists(so) else None
    if os.path.exists(so) and os.path.getmtime(so) >= os.path.getmtime(REF_PYX):
        return so
    try:
        import numpy
        from Cython.Compiler.Main import compile as cython_compile, CompilationOptions, default_options
    except Exception as e:  # no Cython -> reference unbuildable here
        if verbose:
            print("build_ref: Cython/numpy unavailable:", e)
        return None
    os.makedirs(OUT_DIR, exist_ok=True)
    c_file = os.path.join(OUT_DIR, "solve_parallel_mc.c")
    opts = CompilationOptions(default_options, output_file=c_file, language_level=3)
    res = cython_compile(REF_PYX, opts)
    if res.num_errors:
        raise RuntimeError("cythonizing the reference solver failed")
    cmd = ["gcc", "-O2", "-shared", "-fPIC", "-w",
           "-I" + sysconfig.get_paths()["include"], "-I" + numpy.get_include(),
           c_file, "-o", so]
    subprocess.check_call(cmd)
    return so


def load():
    """Return the reference's `solve_parallel` (fp64, in place) or None."""
    so = ref_so_path()
    if not os.path.exists(so):
        return None
    import importlib.util
    spec = importlib.util.spec_from_file_location("solve_parallel_mc", so)
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod.solve_parallel


if __name__ == "__main__":
    p = build(verbose=True)
    print("oracle/_ref:", p)
    sys.exit(0 if p else 1)
