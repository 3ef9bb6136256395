"""BUILD CONTAINER ONLY: time the reference's own CPU inverse (its rebuilt Cython solver, oracle/_ref) on the bench
workloads and write profiles/reference_cpu_container.json.  The compiled reference does not travel to the GPU box
(.gpurunignore lists oracle/_ref/), so bench.py carries these numbers as constants with their provenance, next to the
pinned port (oracle/finc_oracle.c) it times live on the box.

    python scripts/measure_reference_cpu.py

The call pattern is FastFlowUnit.reverse_level1's (fastflow.py:57-76 -> layers/conv.py:113-163): one
solve_parallel(fp64, in place) per group and image, single thread (setup.py:1-5 builds without OpenMP).
"""
import json
import os
import platform
import sys
import time

import numpy as np

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
from oracle import build_ref, oracle  # noqa: E402

CASES = {"c3": (2, 96, 64, 64, 3, 0.05), "c2": (8, 48, 32, 32, 3, 0.05), "c5": (1, 192, 128, 128, 5, 0.02)}


def main():
    build_ref.build()
    solve = build_ref.load()
    assert solve is not None, "oracle/_ref not built (needs /root/reference + Cython)"
    out = {"host": platform.processor() or platform.machine(), "cpu_count": os.cpu_count(),
           "source": "scripts/measure_reference_cpu.py in the build container: solve_parallel_mc.pyx rebuilt by "
                     "oracle/build_ref.py, one fp64 solve per (image, group), single thread as the reference runs it",
           "workloads": {}}
    for name, (n, C, H, W, K, std) in CASES.items():
        Cq = C // 4
        ws = oracle.make_stored_weights(4, Cq, K, K, std=std)
        wc = oracle.canonicalize(ws, 4, oracle.ORIENT_FASTFLOW)
        x = np.random.default_rng(0).standard_normal((n, C, H, W)).astype(np.float32)
        z = oracle.forward_f32(x, wc, nthreads=oracle.max_threads())
        t0 = time.perf_counter()
        sol = np.empty_like(z, dtype=np.float64)
        for b in range(n):
            for g in range(4):
                zz = np.ascontiguousarray(z[b:b + 1, g * Cq:(g + 1) * Cq], dtype=np.float64)
                solve(zz, np.ascontiguousarray(wc[g * Cq:(g + 1) * Cq], dtype=np.float64), (K, K))
                sol[b:b + 1, g * Cq:(g + 1) * Cq] = zz
        dt = time.perf_counter() - t0
        # the port, same images, same single thread: the two must agree bit for bit and run at a comparable rate
        t0 = time.perf_counter()
        port = oracle.inverse_via_f64(z, wc, nthreads=1)
        dtp = time.perf_counter() - t0
        # the reference solves the canonical system of every group; un-flip to compare with the port's output
        ref = sol.astype(np.float32)
        out["workloads"][name] = {"images": n, "reference_images_per_s": n / dt, "port_1thread_images_per_s": n / dtp,
                                  "bit_equal_on_TL_group": bool(np.array_equal(ref[:, :Cq], port[:, :Cq]))}
        print(name, out["workloads"][name], flush=True)
    with open(os.path.join(REPO, "profiles", "reference_cpu_container.json"), "w") as f:
        json.dump(out, f, indent=1)


if __name__ == "__main__":
    main()
