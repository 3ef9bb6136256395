"""bench.py's N>1 entry point on the CPU: `python bench.py --gpus 2` must start 2 ranks by itself (the
driver calls exactly this), rendezvous on 127.0.0.1, and have rank 0 print ONE JSON line with n_gpus = 2.
FINC_BENCH_STUB=1 swaps the GPU step for a host-only stand-in so spawn / barrier / gather run here over gloo."""
import json
import os
import subprocess
import sys

import pytest

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def run_bench(extra, env_extra=None, timeout=300):
    env = dict(os.environ, FINC_BENCH_STUB="1", FINC_BENCH_BACKEND="gloo")
    env.pop("WORLD_SIZE", None)
    env.pop("RANK", None)
    env.update(env_extra or {})
    return subprocess.run([sys.executable, os.path.join(REPO, "bench.py"), "--steps", "5", "--warmup", "2"] + extra,
                          env=env, capture_output=True, text=True, timeout=timeout, cwd=REPO)


def json_lines(stdout):
    return [json.loads(l) for l in stdout.splitlines() if l.startswith("{")]


def test_single_rank_line():
    r = run_bench(["--gpus", "1"])
    assert r.returncode == 0, r.stderr
    lines = json_lines(r.stdout)
    assert len(lines) == 1 and lines[0]["n_gpus"] == 1 and lines[0]["steps"] == 5 and lines[0]["warmup"] == 2
    assert lines[0]["config"]["world_size_seen"] == 1


def test_gpus_2_spawns_two_ranks():
    r = run_bench(["--gpus", "2"])
    assert r.returncode == 0, r.stderr[-2000:]
    lines = json_lines(r.stdout)
    assert len(lines) == 1, r.stdout
    line = lines[0]
    assert line["n_gpus"] == 2 and line["config"]["world_size_seen"] == 2 and line["config"]["backend"] == "gloo"
    assert len(line["config"]["per_rank_ms"]) == 2
    # value is the whole-job aggregate over max-over-ranks time
    assert line["ms_per_step"] * 5 >= max(line["config"]["per_rank_ms"]) * 0.999
    for k in ("median_ms", "p10_ms", "p90_ms", "mean_ms"):
        assert k in line["launch"]


def test_strong_scaling_keeps_the_global_batch():
    """--scaling strong: the workload's batch is split over the ranks (SURVEY 8e's strong split); weak keeps it per GPU.
    Every line carries the keys the driver indexes, null where they do not apply (cpu_baseline at N > 1)."""
    weak = json_lines(run_bench(["--gpus", "2"]).stdout)[0]
    r = run_bench(["--gpus", "2", "--scaling", "strong"])
    assert r.returncode == 0, r.stderr[-2000:]
    strong = json_lines(r.stdout)[0]
    assert weak["scaling"] == "weak" and strong["scaling"] == "strong"
    assert weak["config"]["per_gpu_batch"] == 4 and weak["config"]["global_batch"] == 8
    assert strong["config"]["per_gpu_batch"] == 2 and strong["config"]["global_batch"] == 4
    for line in (weak, strong):
        assert "cpu_baseline" in line and line["cpu_baseline"] is None and "roofline" in line


def test_world_size_mismatch_fails_loudly():
    r = run_bench(["--gpus", "2"], env_extra={"WORLD_SIZE": "1", "RANK": "0"})
    assert r.returncode != 0 and "WORLD_SIZE" in (r.stderr + r.stdout)


def test_parent_does_not_import_torch_before_spawning():
    """The parent must not touch the GPU before its children exist: the launch decision sits above every torch import."""
    src = open(os.path.join(REPO, "bench.py")).read()
    head = src[:src.index("def launch_ranks")]
    assert "import torch" not in head
    main_body = src[src.index("def main("):]
    assert main_body.index("launch_ranks(args, argv)") < main_body.index("bench_unit(args)")
