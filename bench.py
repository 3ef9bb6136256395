#!/usr/bin/env python3
"""Headline benchmark of the FInC Flow hot path on MI355X.

    python bench.py --gpus N --steps K --warmup W
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...

`--gpus N` with N > 1 and no WORLD_SIZE in the environment starts the N ranks itself (fresh child processes through
torch.distributed.run, BEFORE this process imports torch or touches the GPU) and exits with their return code; under
torchrun (WORLD_SIZE set) it is one rank of the job.  Rank 0 prints ONE JSON line.

Workload (BASELINE.json: "inverse images/sec + fwd+logdet ms/img, 3x3 conv @64x64x96"): one FastFlowUnit,
3x3, C=96 (Cq=24), 64x64, batch 256 PER GPU (configs[2]); synthetic fp32 data, weights by the reference's
init rule (layers/conv.py:63-79).  A step = one pass of the hot path over one batch resident in HBM:
`unit.reverse(z)`, the module's public call (allocation of the result included; one launch of the MFMA
wavefront kernel).  The timed region is exactly K such steps between barrier + torch.cuda.synchronize();
value = images solved by all ranks / max-over-ranks time.  The forward (+logdet, identically 0) is timed the
same way right after and reported next to it, and so is the inverse on z ~ N(0,1) (the sampling distribution,
train/losses.py:42-45).

Images are independent, so ranks shard the batch with no data-path collective; the only RCCL traffic is one broadcast of
the layer's weights before the timed region.  `--scaling weak` (default: what the driver's 1/2/4/8 run measures) keeps the
per-GPU batch fixed; `--scaling strong` keeps the GLOBAL batch of the workload fixed and gives every rank B/N images -- the
split SURVEY 8(e) worries about (c3: 32 images = 128 problems per GPU at N = 8, fewer problems than compute units).  The
N = 1 line also carries `strong_share`: the time of ONE GPU on the share B/P it would get at P = 2, 4, 8 (a single-GPU
measurement of the per-GPU work of a strong split, NOT a multi-GPU measurement).

Extra objects on the JSON line:
  roofline      dominant kernel (inverse): algorithmic bytes per launch (8*E + 4*C*Cq*KH*KW)
                / mean launch duration from HIP events on the launch stream, vs 8 TB/s HBM peak; `traffic`
                = HBM bytes per launch from rocprofv3 PMC passes (profiles/traffic_*.json; a profiler pass, not
                measured in this run) or null.
  cpu_baseline  the pinned CPU port of the reference's Cython solver (oracle/finc_oracle.c, bit-equal to it on the golden
                vectors) timed on this box's host cores, one thread as the reference runs its solver, on a bounded sample of
                the same workload (kind "port"); beside it the same solve on all cores, and -- as constants measured in the
                build container, because the compiled reference never travels -- the reference's own rate and the
                port / reference ratio on one core there.

Self-diagnosis (round 4): every leg carries the board power read from sysfs while it ran (`power_w`), and the `clock` object
re-runs the two inverse legs -- z = forward(x) and z ~ N(0,1) -- in both orders with a one-wave clock probe beside them
(include/finc.h: finc_debug_clock_probe_*): the shader clock each leg actually ran at.  `config.library` / `config.runtime_switches`
say which library file ran and that no A/B switch was in effect (a judged line is refused otherwise).

`--workload ref_timing` reproduces the one protocol the reference publishes numbers for (fastflow/timing_comparision.py:10-14,
fastflow/test_layers.py:1624-1693): `sample(100)` of the FInC stack (num_blocks=2, block_size=16, actnorm, split prior) at seven
image sizes, 1 warm-up + mean of 10, wall clock, and prints them beside the published seconds (unstated NVIDIA GPU, CUDA 10.2).

FINC_BENCH_STUB=1 replaces the step by a host-only stand-in (no GPU, no HIP library): it exists so that the
launcher / rendezvous / gather logic is covered by a CPU test (tests/test_bench_launcher.py), never for numbers.
"""
import argparse
import json
import os
import socket
import subprocess
import sys
import time

REPO = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, REPO)

HBM_PEAK_GBS = 8000.0      # MI355X_MICROARCH.md: 8.0 TB/s spec
FP32_PEAK_TFLOPS = 157.3   # same table: fp32 vector == fp32-input MFMA

WORKLOADS = {
    # name: (per-GPU batch, C, H, W, K, weight std)
    "c3": (256, 96, 64, 64, 3, 0.05),   # BASELINE configs[2] -- the metric's shape
    "c2": (64, 48, 32, 32, 3, 0.05),    # BASELINE configs[1]
    # BASELINE configs[4]: batch 512 sharded over 8 GPUs = 64 per GPU.  std 0.02: at 5x5/Cq=48 the init std 0.05 of
    # layers/conv.py:64 makes the inverse itself unstable; 0.02 matches the operator norm of c3.
    "c5": (64, 192, 128, 128, 5, 0.02),
    # the per-GPU work of an 8-way STRONG split of configs[2] (256 images over 8 GPUs): 128 problems, fewer than compute units --
    # the role-split kernel's regime (profiles, A/B runs; `--scaling strong --gpus 8` runs exactly this on every rank)
    "c3_share8": (32, 96, 64, 64, 3, 0.05),
    # NOT a BASELINE config: a bank outside every register-resident table (4 groups x 128 channels, 3x3), i.e. the streaming-bank
    # kernel of finc_stream.hip through the unchanged FastFlowUnit path, so that this kernel has a line in the judged
    # format too -- roofline and CPU baseline beside it.  std = 0.05 * sqrt(24 / 128): the operator norm of c3's bank.
    "w512": (256, 512, 32, 32, 3, 0.02165),
    # NOT a BASELINE config either: configs[2]'s layer at a batch that is no whole number of the one-wave kernel's rounds (1,280 problems =
    # 1,024 + 256) -- the inverse is then two launches, the whole round on the wavefront kernel and the remainder's images on the
    # role-split kernel (finc_mfma.hip remainder_images) -- so that this launch rule has a line in the judged format too
    "c3_b320": (320, 96, 64, 64, 3, 0.05),
}
CONFIG_INDEX = {"c3": 2, "c2": 1, "c5": 4, "c3_share8": 2, "w512": None, "c3_b320": None}
NOT_BASELINE = {"w512": "not a BASELINE config (a bank outside the register-resident tables: the streaming-bank kernel)",
                "c3_b320": "not a BASELINE config (configs[2]'s layer at 320 images: one round of the wavefront kernel + a remainder launch)"}
# single-thread CPU sample sizes (images): about 10-20 s of host work per workload
CPU_SAMPLE_1T = {"c3": 160, "c2": 64, "c5": 3, "c3_share8": 32, "w512": 8, "c3_b320": 160}


def parse(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--warmup", type=int, default=20)
    ap.add_argument("--workload", default="c3", choices=sorted(WORKLOADS) + ["c4", "ref_timing"])
    ap.add_argument("--ref-block-size", type=int, default=16, help="ref_timing: layers per block (16: the 12.86 M series; 48: the 39.47 M one)")
    ap.add_argument("--allow-switches", action="store_true",
                    help="A/B runs only: report a line although FINC_* switches / a FINCFLOW_LIB override are in effect (never for a judged run)")
    ap.add_argument("--no-clock", action="store_true", help="skip the clock-diagnosis legs")
    ap.add_argument("--scaling", default="weak", choices=["weak", "strong"],
                    help="weak: the workload's batch per GPU; strong: the workload's batch in total, B/N per GPU")
    ap.add_argument("--cpu-sample", type=int, default=None, help="images in the single-thread CPU baseline sample")
    ap.add_argument("--no-cpu", action="store_true", help="skip the CPU baseline leg")
    ap.add_argument("--no-share", action="store_true",
                    help="skip the strong_share legs (profile runs: they launch other variants of the inverse kernel)")
    ap.add_argument("--master-port", type=int, default=None, help="rendezvous port when this process starts the ranks")
    return ap.parse_args(argv)


# ----------------------------------------------------------------------------------------------------------------
# launcher: `python bench.py --gpus N` -> N ranks
# ----------------------------------------------------------------------------------------------------------------
def free_port():
    with socket.socket(socket.AF_INET, socket.SOCK_STREAM) as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def launch_ranks(args, argv):
    """Start args.gpus ranks as fresh children of a process that has not touched the GPU (no torch import so far).
    stdout/stderr pass through, so rank 0's JSON line is this command's JSON line."""
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    env.setdefault("MASTER_ADDR", "127.0.0.1")
    port = args.master_port or free_port()
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={args.gpus}",
           "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + list(argv)
    return subprocess.run(cmd, env=env).returncode


# ----------------------------------------------------------------------------------------------------------------
# the rank harness (shared by the GPU step and the host-only stub)
# ----------------------------------------------------------------------------------------------------------------
class Sensors:
    """Board power and the driver's shader-clock reading of ONE card, from sysfs (hwmon), sampled by a thread of this process
    every few milliseconds while a leg runs.  Reading sysfs touches neither HIP nor the card's queues.  Everything here is best
    effort: a box that does not expose the files yields nulls, never a failure.  (MI355X_MICROARCH.md: board power and
    pp_dpm_sclk are NOT the in-kernel clock -- that is what the clock probe measures; power is what tells a power-capped leg from
    an idle one.)"""

    def __init__(self, torch, dev):
        self.power_path = self.freq_path = None
        self.note = None
        try:
            import glob
            props = torch.cuda.get_device_properties(dev)
            bus = f"{getattr(props, 'pci_domain_id', 0):04x}:{props.pci_bus_id:02x}:{props.pci_device_id:02x}.0"
            base = f"/sys/bus/pci/devices/{bus}"
            if not os.path.isdir(base):
                cards = sorted(glob.glob("/sys/class/drm/card[0-9]*/device"))
                amd = [c for c in cards if open(os.path.join(c, "vendor")).read().strip() == "0x1002"] if cards else []
                base = amd[getattr(dev, "index", 0) or 0] if amd else None
            if base:
                for hw in sorted(glob.glob(os.path.join(base, "hwmon", "hwmon*"))):
                    for name in ("power1_average", "power1_input"):
                        f = os.path.join(hw, name)
                        if self.power_path is None and os.path.exists(f):
                            self.power_path = f
                    f = os.path.join(hw, "freq1_input")
                    if self.freq_path is None and os.path.exists(f):
                        self.freq_path = f
            self.note = f"sysfs {base}" if base else "no sysfs node for the device"
        except Exception as e:                              # (permissions, containers without sysfs ...)
            self.note = f"sysfs unavailable: {type(e).__name__}"
        self._stop = None
        self._thread = None
        self._p, self._f = [], []

    @staticmethod
    def _read(path):
        try:
            with open(path) as f:
                return float(f.read().strip())
        except (OSError, ValueError):
            return None

    def start(self):
        import threading
        if self.power_path is None and self.freq_path is None:
            return
        self._p, self._f = [], []
        self._stop = threading.Event()
        self._quiet = threading.Event()

        def run():
            while not self._stop.is_set():
                if self._quiet.is_set():                    # (a timed region: no driver file is touched while it runs)
                    self._stop.wait(0.001)
                    continue
                ts = time.perf_counter()
                if self.power_path:
                    v = self._read(self.power_path)
                    if v is not None:
                        self._p.append((ts, v * 1e-6))      # microwatts
                if self.freq_path:
                    v = self._read(self.freq_path)
                    if v is not None:
                        self._f.append((ts, v * 1e-6))      # hertz
                self._stop.wait(0.004)

        self._thread = threading.Thread(target=run, daemon=True)
        self._thread.start()

    def quiet(self, on):
        """No sampling inside a timed region: a read of the hwmon files goes through the driver to the SMU, and the one launch on record
        that took 38 ms (profiles/r05/notes/bench_c3_share8_outlier_run.json) fell together with a sampler stalled for as long.  The
        samples that describe a leg are taken in the 12 ms behind it, while the same launches keep the card busy."""
        if self._thread is None:
            return
        if on:
            self._quiet.set()
        else:
            self._quiet.clear()

    def stop(self, t0=None, t1=None):
        """Stops the sampler; the statistics cover the samples taken in [t0, t1] (the window right behind the timed region -- the
        sampler itself is started before the leg's untimed spin-up, so that the first, slow, read of the driver's files is long past)."""
        if self._thread is None:
            return {"power_w": None, "smi_sclk_mhz": None, "samples": 0, "source": self.note}
        self._stop.set()
        self._thread.join()
        self._thread = None
        inside = lambda v: [x for ts, x in v if (t0 is None or ts >= t0) and (t1 is None or ts <= t1)] or [x for _, x in v[-1:]]
        p, f = inside(self._p), inside(self._f)
        mean = lambda v: (sum(v) / len(v)) if v else None
        return {"power_w": mean(p), "power_w_max": max(p) if p else None, "smi_sclk_mhz": mean(f),
                "samples": max(len(p), len(f)), "source": self.note}


class Harness:
    def __init__(self, args, stub):
        import torch
        import torch.distributed as dist
        self.torch, self.dist, self.stub = torch, dist, stub
        self.world = int(os.environ.get("WORLD_SIZE", "1"))
        self.rank = int(os.environ.get("RANK", "0"))
        self.local_rank = int(os.environ.get("LOCAL_RANK", "0"))
        if self.world != args.gpus:
            raise SystemExit(f"bench.py: --gpus {args.gpus} but WORLD_SIZE={self.world}: launch N ranks with --gpus N")
        self.backend = None
        if stub:
            self.dev = torch.device("cpu")
        else:
            ndev = max(torch.cuda.device_count(), 1)
            self.dev = torch.device("cuda", self.local_rank % ndev if self.world > 1 else 0)
            torch.cuda.set_device(self.dev)
        if self.world > 1:
            os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
            os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
            # one rank per GPU over RCCL ("nccl" IS RCCL on ROCm).  FINC_BENCH_BACKEND=gloo lets the N>1 code path be
            # rehearsed on the CPU (stub) or with several ranks on ONE GPU (RCCL refuses two ranks per device); never
            # used for numbers.
            self.backend = os.environ.get("FINC_BENCH_BACKEND", "gloo" if stub else "nccl")
            if self.backend == "nccl":
                dist.init_process_group("nccl", device_id=self.dev)
            else:
                dist.init_process_group(self.backend)
            # multi-GPU readiness, checked rather than assumed: the job has as many ranks as --gpus says, and (RCCL) every
            # rank drives its own card -- two ranks on one device would halve each rank's throughput and still "scale"
            assert dist.get_world_size() == args.gpus == self.world, (dist.get_world_size(), args.gpus, self.world)
        self.devices = self.gather_devices()
        self.sensors = None if stub else Sensors(torch, self.dev)
        self.spin_up_s = 0.0 if stub else 0.25

    def gather_devices(self):
        """[(rank, device index, PCI bus id)] of every rank.  With RCCL the cards must be distinct."""
        torch, dist = self.torch, self.dist
        if self.stub:
            mine = [self.rank, -1, -1]
        else:
            pr = torch.cuda.get_device_properties(self.dev)
            mine = [self.rank, torch.cuda.current_device(), int(getattr(pr, "pci_bus_id", -1))]
        if self.world == 1:
            return [mine]
        tt = torch.tensor(mine, dtype=torch.int64, device=self.dev if self.backend == "nccl" else "cpu")
        got = [torch.zeros_like(tt) for _ in range(self.world)]
        dist.all_gather(got, tt)
        devs = [[int(v) for v in g.tolist()] for g in got]
        if self.backend == "nccl":
            cards = {(d[1], d[2]) for d in devs}
            assert len(cards) == self.world, f"ranks share a GPU: {devs}"
        return devs

    def sync(self):
        if not self.stub:
            self.torch.cuda.synchronize()

    def barrier(self):
        if self.world > 1:
            self.dist.barrier()
        self.sync()

    def spin_up(self, fn):
        """Untimed preamble, before the W warmup steps: the card needs ~50 launches (tens of ms) after idling
        before its clocks settle; a 20-step run measured cold reads 15-20 % low.  Not a step, not timed;
        its length is reported as `spin_up_s`."""
        t_end = time.perf_counter() + self.spin_up_s
        while time.perf_counter() < t_end:
            for _ in range(10):
                fn()
            self.sync()

    def timed(self, fn, steps, warmup):
        """W untimed steps, then exactly K steps between barrier+synchronize; per-launch durations from HIP events on
        the launch stream (torch's current stream IS the stream the ABI call launches on).  Returns the max-over-ranks
        wall time, every rank's wall time and the sorted per-launch milliseconds of this rank."""
        torch = self.torch
        if self.sensors is not None and not self.stub:
            self.sensors.start()
        self.spin_up(fn)
        for _ in range(warmup):
            fn()
        if self.stub:
            stamps = []
            self.barrier()
            t0 = time.perf_counter()
            for _ in range(steps):
                a = time.perf_counter()
                fn()
                stamps.append((time.perf_counter() - a) * 1e3)
            self.barrier()
            dt = time.perf_counter() - t0
            per = sorted(stamps)
        else:
            evs = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(steps)]
            if self.sensors is not None:
                self.sensors.quiet(True)
            self.barrier()
            t0 = time.perf_counter()
            for a, b in evs:
                a.record()
                fn()
                b.record()
            self.barrier()
            dt = time.perf_counter() - t0
            if self.sensors is not None:
                # (keep the card busy while the sensors are read: the readings then describe the leg, not the idle card behind it)
                t1 = time.perf_counter()
                self.sensors.quiet(False)
                while time.perf_counter() - t1 < 0.012:
                    fn()
                    self.sync()
            self.last_sensors = self.sensors.stop(t0 + dt, None) if self.sensors is not None else None
            per = sorted(a.elapsed_time(b) for a, b in evs)
        ranks = [dt]
        if self.world > 1:
            tt = torch.tensor([dt], device=self.dev if self.backend == "nccl" else "cpu", dtype=torch.float64)
            gathered = [torch.zeros_like(tt) for _ in range(self.world)]
            self.dist.all_gather(gathered, tt)
            ranks = [float(g.item()) for g in gathered]
        return max(ranks), ranks, per

    def probed(self, fn, steps, warmup):
        """A DIAGNOSTIC leg (never `value`): the same K launches with the one-wave clock probe running beside them on its own
        stream.  The probe must not meet a device-wide synchronisation, so this leg ends on a synchronisation of the launch
        stream; per-launch times are HIP events as everywhere.  Returns (launch stats, probe stats, sensor stats)."""
        torch = self.torch
        from fincflow_amd import _lib
        self.sensors.start()
        self.spin_up(fn)
        for _ in range(warmup):
            fn()
        evs = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(steps)]
        self.barrier()
        stream = torch.cuda.current_stream(self.dev)
        _lib.clock_probe_begin(period_us=100, max_ms=3000)
        t0 = time.perf_counter()
        for a, b in evs:
            a.record()
            fn()
            b.record()
        stream.synchronize()
        sens = self.sensors.stop(t0, time.perf_counter())
        probe = _lib.clock_probe_end()
        self.barrier()
        return launch_stats(sorted(a.elapsed_time(b) for a, b in evs)), probe, sens

    def finish(self):
        if self.world > 1:
            self.dist.barrier()
            self.dist.destroy_process_group()


def launch_stats(per):
    n = len(per)
    pick = lambda f: per[min(n - 1, max(0, int(round(f * (n - 1)))))]
    return {"mean_ms": sum(per) / n, "median_ms": pick(0.5), "p10_ms": pick(0.1), "p90_ms": pick(0.9)}


# ----------------------------------------------------------------------------------------------------------------
# CPU baseline
# ----------------------------------------------------------------------------------------------------------------
def cpu_baseline(workload, B, C, H, W, K, std, sample_1t):
    """`value`: the pinned CPU port (oracle/finc_oracle.c: a restatement of solve_parallel_mc.pyx:77-126, bit-equal to the
    reference's Cython solver on the golden vectors and, in the build container, on fresh data), timed HERE on this box's host
    cores on ONE thread, as the reference runs its solver (setup.py:1-5 builds without OpenMP) -- kind "port".  `all_cores`: the
    same solve with OpenMP over (image, group).  The compiled reference itself never travels to the GPU box (BASELINE.md 2,
    SURVEY 8c): its rate is carried as a constant measured in the build container (`reference_in_build_container`,
    scripts/measure_reference_cpu.py) with the port's rate on the same core beside it, so `port_over_reference` lets a
    reader scale `value` to the reference's own solver."""
    import numpy as np
    from oracle import oracle
    ws = oracle.make_stored_weights(4, C // 4, K, K, std=std)
    wc = oracle.canonicalize(ws, 4, oracle.ORIENT_FASTFLOW)
    rng = np.random.default_rng(0)
    ncores = os.cpu_count() or 1
    nthreads = min(oracle.max_threads(), ncores)
    n_all = min(B, max(sample_1t, 2 * nthreads))
    x = rng.standard_normal((n_all, C, H, W)).astype(np.float32)
    z = oracle.forward_f32(x, wc, nthreads=nthreads)
    n1 = min(sample_1t, n_all)
    t0 = time.perf_counter()
    x1 = oracle.inverse_via_f64(z[:n1], wc, nthreads=1)
    t_1 = time.perf_counter() - t0
    assert np.abs(x1 - x[:n1]).max() / np.abs(x[:n1]).max() < 1e-4
    t0 = time.perf_counter()
    oracle.inverse_via_f64(z, wc, nthreads=nthreads)
    t_all = time.perf_counter() - t0
    port = {"value": n1 / t_1, "unit": "images/s", "cores": 1, "kind": "port",
            "sample": f"{n1} images of the bench workload, fp64 solve (oracle/finc_oracle.c), single thread as the "
                      f"reference runs its solver"}
    out = dict(port)
    out["all_cores"] = {"value": n_all / t_all, "unit": "images/s", "cores": nthreads, "kind": "port",
                        "sample": f"{n_all} images, same solve, OpenMP over image x group"}
    path = os.path.join(REPO, "profiles", "reference_cpu_container.json")
    if os.path.exists(path):
        with open(path) as f:
            ref = json.load(f)
        if workload in ref.get("workloads", {}):
            r = ref["workloads"][workload]
            out["reference_in_build_container"] = {
                "value": r["reference_images_per_s"], "unit": "images/s", "cores": 1, "kind": "reference",
                "port_same_host_1thread": r["port_1thread_images_per_s"],
                "provenance": ref.get("source"), "host_cpu_count": ref.get("cpu_count"),
                "note": "constant, not measured in this run: the compiled reference does not travel to the GPU box"}
            out["port_over_reference"] = r["port_1thread_images_per_s"] / r["reference_images_per_s"]
    return out


def load_traffic(workload):
    path = os.path.join(REPO, "profiles", f"traffic_{workload}.json")
    if os.path.exists(path):
        with open(path) as f:
            return json.load(f)
    return None


# ----------------------------------------------------------------------------------------------------------------
# --workload c4: the Glow stack
# ----------------------------------------------------------------------------------------------------------------
class TimedLibrary:
    """Stands in for the ctypes library while ONE eager pass runs: every launching entry point of the hot path is bracketed by a
    HIP-event pair on the launch stream, so the pass itself says how much of it the finc_* kernels are (`hot_path_share`)."""
    LAUNCHERS = ("finc_inverse_f32", "finc_inverse_packed_f32", "finc_inverse_packed_premultiplied_f32", "finc_forward_f32",
                 "finc_forward_packed_f32", "finc_mix_f32")

    def __init__(self, real, torch):
        self._real, self._torch, self.events = real, torch, []

    def __getattr__(self, name):
        f = getattr(self._real, name)
        if name not in self.LAUNCHERS:
            return f

        def wrapped(*a):
            e0, e1 = self._torch.cuda.Event(enable_timing=True), self._torch.cuda.Event(enable_timing=True)
            e0.record()
            r = f(*a)
            e1.record()
            self.events.append((name, e0, e1))
            return r
        return wrapped


def hot_path_share(torch, run_pass, graph_pass_ms):
    """One eager pass with the library's launches bracketed by events.  `finc_kernel_ms` = the sum of those brackets (a
    bracket holds one kernel; an event pair adds a few microseconds, so the share is an upper bound)."""
    from fincflow_amd import _lib
    real = _lib.lib()
    tl = TimedLibrary(real, torch)
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    _lib._lib = tl
    try:
        torch.cuda.synchronize()
        a.record()
        run_pass()
        b.record()
        torch.cuda.synchronize()
    finally:
        _lib._lib = real
    by = {}
    for name, e0, e1 in tl.events:
        by[name] = by.get(name, 0.0) + e0.elapsed_time(e1)
    finc_ms = sum(by.values())
    eager_ms = a.elapsed_time(b)
    return {"finc_kernel_ms": finc_ms, "finc_launches": len(tl.events), "by_entry_point_ms": by, "eager_pass_ms": eager_ms,
            "graph_pass_ms": graph_pass_ms,
            "finc_kernels_of_pass_time": finc_ms / (graph_pass_ms or eager_ms),
            "source": "measured in this run: HIP-event brackets around every finc_* launch of one eager pass, against the "
                      "replayed pass" if graph_pass_ms else "measured in this run (eager pass)",
            "note": "the coupling nets (MIOpen 3x3, rocBLAS 1x1, elementwise) are the rest: outside the hot path of SURVEY 8"}


def bench_stack(args):
    """--workload c4: BASELINE configs[3], sampling 128 images through the CIFAR Glow stack of
    fastflow_cifar.py:35-63 (num_blocks=3, block_size=32, actnorm, split prior: 96 FastFlowUnits at 16x16 / 8x8 /
    4x4 + ActNorm + 1x1 + coupling nets), "batch-sharded 1->8 GPUs": with N ranks every rank samples 128/N images from
    its own replica of the model (weights broadcast once; sampling needs no collective).  A step = model.sample(n),
    replayed from one HIP graph when capture succeeds.  The reference's published whole-stack numbers
    (timing_comparision.py:10-14) are for a different stack size and batch 100, so vs_baseline stays null.  The step is
    mostly NOT the hot path: `hot_path_share` says what fraction of the kernel time of one pass the finc_* kernels are
    (profiles/r02/c4_kernel_stats.csv)."""
    os.environ.setdefault("MIOPEN_FIND_MODE", "2")   # coupling-net convs: heuristics, not an exhaustive find per shape
    h = Harness(args, stub=False)
    import numpy as np
    torch = h.torch
    from fincflow_amd import FastFlowUnit, glow
    dev, world, rank = h.dev, h.world, h.rank
    torch.manual_seed(0)
    np.random.seed(0)
    total = 128
    if total % world:
        raise SystemExit(f"--workload c4 samples {total} images: --gpus must divide it")
    n = total // world if args.scaling == "strong" or world > 1 else total
    model = glow.create_model(num_blocks=3, block_size=32, actnorm=True, split_prior=True).to(dev).eval()
    if world > 1:
        from fincflow_amd.dist import broadcast_weights
        broadcast_weights(model, src=0)
    with torch.no_grad():
        for m in model:                                    # ActNorm is data-initialised by the first forward only
            if isinstance(m, glow.ActNorm):
                m.initialized.fill_(1)
        for _ in range(3):
            s, _ = model.sample(n)
        torch.cuda.synchronize()
        graph, mode = None, "eager"
        try:
            g = torch.cuda.CUDAGraph()
            with torch.cuda.graph(g):
                s, _ = model.sample(n)
            g.replay()
            torch.cuda.synchronize()
            graph, mode = g, "hip-graph"
        except Exception as e:                             # capture is an optimisation, not a requirement
            sys.stderr.write(f"graph capture failed, running eager: {e}\n")
            torch.cuda.synchronize()
        fn = (lambda: graph.replay()) if graph is not None else (lambda: model.sample(n))
        dt, ranks, per = h.timed(fn, args.steps, args.warmup)
        finite = bool(torch.isfinite(s).all())
        share = hot_path_share(torch, lambda: model.sample(n), launch_stats(per)["mean_ms"] if graph is not None else None)
    units = sum(isinstance(m, FastFlowUnit) for m in model)
    from fincflow_amd import _lib as _lq
    switches, libinfo = _lq.runtime_switches(), _lq.library_info()
    refuse_switches(args, switches, libinfo)
    if rank == 0:
        print(json.dumps({
            "metric": "sampled images/sec, CIFAR Glow stack (fastflow_cifar.py create_model, 96 FastFlowUnits)",
            "value": world * n * args.steps / dt, "unit": "images/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": dt / args.steps * 1e3, "higher_is_better": True,
            "scaling": "strong" if world > 1 else args.scaling, "vs_baseline": None,
            "dtype": "f32", "data": "synthetic",
            "config": {"workload": f"BASELINE configs[3]: model.sample({n}) per GPU ({world * n} images in all), num_blocks=3, "
                                   f"block_size=32, actnorm, split_prior; {units} FastFlowUnits (HIP) + ActNorm/Conv1x1/Coupling "
                                   f"(PyTorch-ROCm); random init",
                       "mode": mode, "finite": finite, "per_gpu_samples": n, "global_samples": world * n,
                       "parallelism": f"batch-sharded x{world}, no collective on the data path",
                       "world_size_seen": world, "backend": h.backend, "per_rank_ms": [r * 1e3 for r in ranks],
                       "rank_devices": h.devices, "library": libinfo, "runtime_switches": switches,
                       "hot_path_share": share},
            "launch": launch_stats(per), "power": h.last_sensors, "roofline": None, "cpu_baseline": None}), flush=True)
    h.finish()


def refuse_switches(args, switches, libinfo):
    """A judged line comes from the in-tree product library under its own dispatch, or not at all."""
    bad = []
    if switches:
        bad.append(f"run-time switches in effect: {switches}")
    if libinfo["build_flags"] != 0:
        bad.append(f"library built with measurement knobs ({libinfo['build_flags']:#x})")
    if libinfo["env_override"]:
        bad.append(f"FINCFLOW_LIB override: {libinfo['path']}")
    if bad and not args.allow_switches:
        raise SystemExit("bench.py: refusing to print a benchmark line -- " + "; ".join(bad) +
                         " (A/B runs: --allow-switches; the line then says so in config)")


# ----------------------------------------------------------------------------------------------------------------
# --workload ref_timing: the one protocol the reference publishes numbers for
# ----------------------------------------------------------------------------------------------------------------
REF_TIMING = {
    # fastflow/timing_comparision.py:10-14 (seconds per sample(100)); image sizes: the axis labels at :45
    "sizes": [(8, 8), (8, 16), (16, 16), (16, 32), (32, 32), (32, 64), (64, 64)],
    16: [0.05727847329999989, 0.06612396699999952, 0.08215905969999984, 0.11489676929999995, 0.17220211640000027,
         0.29212771209999955, 0.5724227276999997],
    48: [0.1746840266999996, 0.20273688460000017, 0.2525629038000005, 0.35165285309999916, 0.5279688547000007,
         0.8938244006000033, 1.7693690774000033],
    "hardware": "unstated NVIDIA GPU, CUDA 10.2 (env.sh:1-2)",
    "protocol": "create_model_fastflow(num_blocks=2, block_size=B, actnorm=True, split_prior=True, current_size=(3,s,s')); "
                "sample(n_samples=100); 1 warm-up + mean of 10 (fastflow/test_layers.py:1624-1693; the reference reads "
                "time.process_time(), its inverse synchronises the device at every launch -- here: wall clock, synchronised)",
}


def bench_ref_timing(args):
    """Seconds per `model.sample(100)` for the reference's timing sweep.  `value` is the (64, 64) point -- the size BASELINE.json's
    metric is quoted at -- and `vs_baseline` = value / published seconds (time-like: below 1 is faster).  Context for the
    north star's "reported wall-clock", NOT a same-hardware comparison: the published numbers are from an unnamed NVIDIA GPU.
    Eager, exactly as the protocol runs it (one Python call per layer); the same pass replayed from one HIP graph is reported
    beside it."""
    os.environ.setdefault("MIOPEN_FIND_MODE", "2")
    h = Harness(args, stub=False)
    if h.world != 1:
        raise SystemExit("--workload ref_timing is a single-GPU protocol")
    torch = h.torch
    from fincflow_amd import FastFlowUnit, _lib, glow
    bs = args.ref_block_size
    published = REF_TIMING.get(bs)
    rows = []
    n_samples, n_loops = 100, 10
    for idx, (hh, ww) in enumerate(REF_TIMING["sizes"]):
        torch.manual_seed(0)
        model = glow.create_model(num_blocks=2, block_size=bs, actnorm=True, split_prior=True, image_size=(3, hh, ww)).to(h.dev).eval()
        with torch.no_grad():
            for m in model:
                if isinstance(m, glow.ActNorm):
                    m.initialized.fill_(1)
            times = []
            for i in range(n_loops + 1):                       # the protocol: 1 warm-up + mean of 10
                torch.cuda.synchronize()
                t0 = time.perf_counter()
                s, _ = model.sample(n_samples)
                torch.cuda.synchronize()
                if i:
                    times.append(time.perf_counter() - t0)
            eager = sum(times) / len(times)
            graph_s = None
            try:
                g = torch.cuda.CUDAGraph()
                with torch.cuda.graph(g):
                    s, _ = model.sample(n_samples)
                g.replay()
                torch.cuda.synchronize()
                gt = []
                for i in range(n_loops + 1):
                    torch.cuda.synchronize()
                    t0 = time.perf_counter()
                    g.replay()
                    torch.cuda.synchronize()
                    if i:
                        gt.append(time.perf_counter() - t0)
                graph_s = sum(gt) / len(gt)
            except Exception as e:
                sys.stderr.write(f"graph capture failed at {(hh, ww)}: {e}\n")
                torch.cuda.synchronize()
            share = hot_path_share(torch, lambda: model.sample(n_samples), graph_s * 1e3 if graph_s else None)
            finite = bool(torch.isfinite(s).all())
        units = sum(isinstance(m, FastFlowUnit) for m in model)
        params = sum(p.numel() for p in model.parameters())
        pub = published[idx] if published else None
        rows.append({"image_size": [3, hh, ww], "seconds_eager": eager, "seconds_graph": graph_s, "published_seconds": pub,
                     "ratio_eager_over_published": eager / pub if pub else None,
                     "ratio_graph_over_published": graph_s / pub if (pub and graph_s) else None,
                     "finite": finite, "units": units, "parameters": params, "hot_path_share": share})
        del model, s
        torch.cuda.empty_cache()
    assert _lib.hlp_timeouts() == 0
    switches, libinfo = _lib.runtime_switches(), _lib.library_info()
    refuse_switches(args, switches, libinfo)
    last = rows[-1]
    print(json.dumps({
        "metric": "seconds per sample(100), FInC stack (num_blocks=2, block_size=%d) at 3x64x64 -- the reference's timing protocol" % bs,
        "value": last["seconds_eager"], "unit": "s", "n_gpus": 1, "steps": n_loops, "warmup": 1,
        "ms_per_step": last["seconds_eager"] * 1e3, "higher_is_better": False, "scaling": "weak",
        "vs_baseline": last["ratio_eager_over_published"], "dtype": "f32", "data": "synthetic",
        "config": {"workload": f"fastflow/timing_comparision.py:10-14 sweep: sample({n_samples}) of create_model(num_blocks=2, "
                               f"block_size={bs}, actnorm, split_prior) at 7 image sizes; random init; eager (one Python call per "
                               f"layer, as the protocol runs it); FastFlowUnits on the HIP kernels, the rest PyTorch-ROCm",
                   "protocol": REF_TIMING["protocol"], "published_hardware": REF_TIMING["hardware"],
                   "comparison": "context for the bar, NOT a same-node comparison: the published seconds are from other, unnamed "
                                 "hardware; vs_baseline = value / published (below 1 = faster)",
                   "library": libinfo, "runtime_switches": switches},
        "sweep": rows, "roofline": None, "cpu_baseline": None}), flush=True)
    h.finish()


# ----------------------------------------------------------------------------------------------------------------
# host-only stub (launcher / gather coverage on the CPU)
# ----------------------------------------------------------------------------------------------------------------
def bench_stub(args):
    import numpy as np
    h = Harness(args, stub=True)
    B = 4 // h.world if args.scaling == "strong" else 4      # (strong: the global batch is fixed, every rank owns B / N)
    a = np.random.default_rng(h.rank).standard_normal((B, 64, 64))

    def fn():
        np.linalg.norm(a @ a.transpose(0, 2, 1))

    dt, ranks, per = h.timed(fn, args.steps, args.warmup)
    if h.rank == 0:
        print(json.dumps({"metric": "stub steps/s (host-only stand-in: launcher test, not a measurement)",
                          "value": h.world * B * args.steps / dt, "unit": "images/s", "n_gpus": h.world,
                          "steps": args.steps, "warmup": args.warmup, "ms_per_step": dt / args.steps * 1e3,
                          "higher_is_better": True, "scaling": args.scaling, "vs_baseline": None, "dtype": "f64",
                          "data": "stub", "config": {"workload": "host-only stub", "backend": h.backend, "per_gpu_batch": B,
                                                     "global_batch": B * h.world,
                                                     "world_size_seen": h.world, "per_rank_ms": [r * 1e3 for r in ranks],
                                                     "rank_devices": h.devices},
                          "launch": launch_stats(per), "roofline": None, "cpu_baseline": None}), flush=True)
    h.finish()


# ----------------------------------------------------------------------------------------------------------------
# the hot path
# ----------------------------------------------------------------------------------------------------------------
def inverse_kernel_name(B, Cq, H, W, K):
    """The inverse kernel this process launches for the workload, as the library itself reports it (not a constant)."""
    from fincflow_amd import _lib
    v = _lib.inverse_variant(B, 4, Cq, H, W, K, K)
    if v is None:
        return "inverse_strict_kernel<float> (inverse, scalar)"
    if v["sec"] == 7:
        return (f"finc_stream_kernel<MT={v['cqp'] // (16 * v['nw'])},NW={v['nw']}> (inverse, streaming bank: the bank read from the L2 once per "
                f"step, the pixels of the last steps in an LDS ring; {v['workgroups']} workgroups of {v['nw']} waves, {v['lds_bytes']} B LDS)")
    if v["sec"] == 4:
        per = v["workgroups"] // (B * 4)
        bands = "" if per == 1 else (f"; the bands of a problem dealt out to {per} workgroups on different compute units, the rows "
                                     f"above a band handed over through memory")
        return (f"finc_split_kernel<CQP={v['cqp']},{K},{K}> (inverse, role-split: 1 wave carries the recurrence, {v['nw'] - 1} prepare "
                f"the rest one step ahead{bands}; {v['workgroups']} workgroups of {v['nw']} waves, {v['lds_bytes']} B LDS)")
    io = {0: "32-byte I/O", 1: "32-byte I/O, lane pairs", 2: "64-byte sector pairing",
          3: "64-byte sector pairing, helper waves do the I/O (512-thread workgroups of 4 problems)"}.get(v["sec"], str(v["sec"]))
    tail = _lib.inverse_remainder_images(B, 4, Cq, H, W, K, K)
    if tail:                                   # whole rounds on this kernel, the last `tail` images on the kernel picked for them alone
        head = _lib.inverse_variant(B - tail, 4, Cq, H, W, K, K)
        return (f"two launches: finc_wave_kernel<CQP={head['cqp']},{K},{K},NW={head['nw']},NPW={head['npw']}> (inverse; {io}; {head['workgroups']} workgroups, "
                f"{head['lds_bytes']} B LDS) on the first {B - tail} images (whole rounds), then the last {tail} images on "
                + inverse_kernel_name(tail, Cq, H, W, K))
    return (f"finc_wave_kernel<CQP={v['cqp']},{K},{K},NW={v['nw']},NPW={v['npw']}> (inverse; {io}; "
            f"{v['workgroups']} workgroups, {v['lds_bytes']} B LDS)")


def bench_unit(args):
    h = Harness(args, stub=False)
    torch = h.torch
    from fincflow_amd import FastFlowUnit
    dev, world, rank = h.dev, h.world, h.rank

    Bw, C, H, W, K, std = WORKLOADS[args.workload]
    if args.scaling == "strong":
        if Bw % world:
            raise SystemExit(f"--scaling strong: the workload's batch {Bw} is not divisible by --gpus {world}")
        B = Bw // world                      # the global batch is the workload's; every rank owns B/N images
    else:
        B = Bw
    Cq = C // 4
    torch.manual_seed(1234)
    unit = FastFlowUnit(C, C, K)
    if std != 0.05:                                      # rescale the free taps, keep the unit-triangular corner
        with torch.no_grad():
            for cv in (unit.conv_tl, unit.conv_tr, unit.conv_bl, unit.conv_br):
                cv.conv.weight.mul_(1 - (1 - std / 0.05) * cv.mask)
    unit = unit.to(dev)
    if world > 1:  # the one collective of the path: replicate the layer (<= 83 KB) from rank 0
        from fincflow_amd.dist import broadcast_weights
        broadcast_weights(unit, src=0)
    gen = torch.Generator(device=dev).manual_seed(1000 + rank)  # each rank owns different images
    x = torch.randn(B, C, H, W, device=dev, generator=gen)
    zn = torch.randn(B, C, H, W, device=dev, generator=gen)     # the sampling distribution
    with torch.no_grad():
        z, logdet = unit(x)
        xr = unit.reverse(z)                      # also builds the packed-fragment cache
    torch.cuda.synchronize()
    err = float((xr - x).abs().max() / x.abs().max())
    assert logdet == 0.0 and err <= 1e-5, f"round trip broken before timing: {err}"
    del xr
    keep = {}

    def step_inverse():
        keep["x"] = unit.reverse(z)

    def step_sampling():
        keep["xs"] = unit.reverse(zn)

    with torch.no_grad():
        inv_dt, inv_ranks, inv_per = h.timed(step_inverse, args.steps, args.warmup)
        inv_sens = h.last_sensors
        fwd_dt, _, fwd_per = h.timed(lambda: unit(x), args.steps, args.warmup)
        fwd_sens = h.last_sensors
        smp_dt, _, smp_per = h.timed(step_sampling, args.steps, args.warmup)
        smp_sens = h.last_sensors
        # Clock diagnosis: the two inverse legs launch the SAME kernel; on one driver box they differed by
        # 27 %.  Both legs again, in both orders, each with the shader clock it really ran at (one-wave probe beside it) and
        # the board power: a swing that follows the clock is DVFS (data-dependent power), one that does not is stalls.
        clock = None
        if world == 1 and not args.no_clock:
            legs = []
            k = max(args.steps, 50)                        # (>= 20 ms per leg: the probe samples every 0.1 ms)
            for order, name, fn in ((1, "round_trip", step_inverse), (1, "sampling", step_sampling),
                                    (2, "sampling", step_sampling), (2, "round_trip", step_inverse)):
                ls, probe, sens = h.probed(fn, k, 5)
                legs.append({"leg": name, "order": order, "launch": ls, "sclk_mhz": probe["mean_mhz"], "sclk": probe,
                             "power_w": sens["power_w"], "sensors": sens,
                             "cycles_per_launch": ls["mean_ms"] * 1e-3 * probe["mean_mhz"] * 1e6 if probe["mean_mhz"] else None})
            by = lambda nm, key: [l[key] for l in legs if l["leg"] == nm and l[key]]
            mean = lambda v: sum(v) / len(v) if v else None
            rt_ms, sm_ms = mean([l["launch"]["mean_ms"] for l in legs if l["leg"] == "round_trip"]), mean([l["launch"]["mean_ms"] for l in legs if l["leg"] == "sampling"])
            rt_clk, sm_clk = mean(by("round_trip", "sclk_mhz")), mean(by("sampling", "sclk_mhz"))
            rt_cyc, sm_cyc = mean(by("round_trip", "cycles_per_launch")), mean(by("sampling", "cycles_per_launch"))
            clock = {"what": "the two inverse legs (same kernel) re-run in both orders, each with a one-wave clock probe beside it "
                             "(d s_memtime / d s_memrealtime x 100 MHz, sampled every 0.1 ms) and the board power from sysfs; "
                             "diagnostic legs, not `value`",
                     "legs": legs,
                     "sampling_over_round_trip": {"time": sm_ms / rt_ms if rt_ms else None,
                                                  "clock": sm_clk / rt_clk if (rt_clk and sm_clk) else None,
                                                  "cycles": sm_cyc / rt_cyc if (rt_cyc and sm_cyc) else None},
                     "reading": "time ratio ~ 1/clock ratio with equal cycles: the swing is the clock (DVFS on data-dependent power); "
                                "cycles ratio > 1: the kernel itself stalls on that data"}
        err_after = float((keep["x"] - x).abs().max() / x.abs().max())
        resid = float((unit(keep["xs"])[0] - zn).abs().max() / zn.abs().max())
    assert err_after <= 1e-5, err_after
    assert resid <= 1e-5, resid
    # SURVEY 8 f1, reported beside the metric (not part of `value`): the layer's training step -- forward, grad-input and
    # grad-weight (+ the fused corner-tap mask) through autograd, gradients reset every step
    xg = x.detach().clone().requires_grad_(True)
    gz = torch.randn_like(z)

    def step_train():
        xg.grad = None
        for p_ in unit.parameters():
            p_.grad = None
        zz, _ = unit(xg)
        zz.backward(gz)

    tr_steps = min(args.steps, 20)
    tr_dt, _, tr_per = h.timed(step_train, tr_steps, 3)
    del xg, gz

    # the per-GPU work of a STRONG split, measured on this one GPU: the share B/P of the workload's batch for P = 2, 4, 8
    strong_share = None
    if world == 1 and args.scaling == "weak" and not args.no_share:
        from fincflow_amd import _lib as _l
        strong_share = {"what": "one GPU on the batch share Bw/P it would own in a P-way strong split of the workload's batch "
                                "(single-GPU measurement of the per-GPU work, NOT a multi-GPU measurement)",
                        "global_batch": Bw, "shares": []}
        with torch.no_grad():
            for P_ in (2, 4, 8):
                if Bw % P_:
                    continue
                zs = z[:Bw // P_].contiguous()
                sh_dt, _, sh_per = h.timed(lambda: unit.reverse(zs), min(args.steps, 50), 5)
                v = _l.inverse_variant(Bw // P_, 4, Cq, H, W, K, K)
                st_ = launch_stats(sh_per)
                strong_share["shares"].append({
                    "P": P_, "per_gpu_batch": Bw // P_, "launch_ms": st_["mean_ms"], "median_ms": st_["median_ms"],
                    "images_per_s_one_gpu": (Bw // P_) / (st_["mean_ms"] * 1e-3),
                    "implied_images_per_s_at_P": Bw / (st_["mean_ms"] * 1e-3),
                    "speedup_over_full_batch_implied": None,
                    "kernel": inverse_kernel_name(Bw // P_, Cq, H, W, K), "form": v["sec"] if v else None})
    from fincflow_amd import _lib as _lt
    hlp_timeouts = _lt.hlp_timeouts()                      # helper-wave protocol: waits that gave up in this process (must be 0)
    assert hlp_timeouts == 0 and not _lt.fault_pending(), f"helper-wave protocol timed out {hlp_timeouts} times: results are not trustworthy"
    switches, libinfo = _lt.runtime_switches(), _lt.library_info()
    refuse_switches(args, switches, libinfo)

    if rank == 0:
        E = B * C * H * W
        alg_bytes = 8 * E + 4 * C * Cq * K * K
        alg_flops = 2 * E * K * K * Cq
        bvar = _lt.backward_variant(B, 4, Cq, H, W, K, K)
        conv_form = bvar["conv_form"]
        # grad-weight: the transposed F(4,3) executes 6 products per tile of 4 columns where the direct sum has 12
        # (5x5 on tile pairs: the transposed F(2,5), 6 products per tile of 2 columns where the direct sum has 10)
        gw_flops = (alg_flops // 2 if bvar["gradw"] == "winograd" or (bvar["gradw"] == "winograd_tiled" and K == 3)
                    else alg_flops * 3 // 5 if bvar["gradw"] == "winograd_tiled" else alg_flops)
        # multiplies the forward kernel executes: Winograd F(2,3) = 4 frequencies x 3 row taps per 2 outputs (2/3 of the direct
        # sum's), F(4,3) = 6 x 3 per 4 outputs (1/2)
        # (5x5: F(2,5) = 6 frequencies x 5 row taps per 2 outputs: 3/5)
        fwd_flops = (alg_flops * 2 // 3 if conv_form == "winograd" else alg_flops // 2 if conv_form in ("winograd4", "winograd4m")
                     else alg_flops * 3 // 5 if conv_form == "winograd25" else alg_flops)
        inv, fwd, smp = launch_stats(inv_per), launch_stats(fwd_per), launch_stats(smp_per)
        inv_launch_ms, fwd_launch_ms = inv["mean_ms"], fwd["mean_ms"]
        inv_gbs = alg_bytes / (inv_launch_ms * 1e-3) / 1e9
        traffic = load_traffic(args.workload)
        P = min(16, W)
        chain = ((H + P - 1) // P) * W + P - 1           # dependent steps of one problem (bands of P rows, chained)
        hbm_ceiling = (alg_bytes / HBM_PEAK_GBS / 1e9) / (alg_flops / FP32_PEAK_TFLOPS / 1e12)
        line = {
            "metric": "inverse images/sec (+ fwd+logdet ms/img in `forward`), 3x3 conv @64x64x96" if args.workload == "c3"
                      else f"inverse images/sec (FastFlowUnit, workload {args.workload})",
            "value": world * B * args.steps / inv_dt,
            "unit": "images/s",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": inv_dt / args.steps * 1e3,
            "higher_is_better": True,
            "scaling": args.scaling,
            "vs_baseline": None,
            "dtype": "f32",
            "data": "synthetic",
            "config": {"workload": (f"BASELINE configs[{CONFIG_INDEX[args.workload]}]" if CONFIG_INDEX[args.workload] is not None
                                    else NOT_BASELINE[args.workload])
                                   + f": FastFlowUnit {K}x{K}, C={C} "
                                   f"(4 groups x Cq={Cq}), {H}x{W}, batch {B} per GPU"
                                   + (f" (strong split of the workload's {Bw})" if args.scaling == "strong" else "")
                                   + f"; step = unit.reverse(z), "
                                   f"z = unit.forward(x), x ~ N(0,1); weights N(0,{std}^2) + reference init rule",
                       "per_gpu_batch": B, "global_batch": B * world, "parallelism": f"batch-sharded x{world}",
                       "world_size_seen": world, "backend": h.backend, "per_rank_ms": [r * 1e3 for r in inv_ranks],
                       "rank_devices": h.devices, "library": libinfo, "runtime_switches": switches,
                       "round_trip_rel_err": err_after, "spin_up_s": h.spin_up_s, "hlp_timeouts": hlp_timeouts},
            "launch": inv,
            "power": inv_sens,
            "sampling": {"what": "the same step on z ~ N(0,1) (train/losses.py:42-45)",
                         "images_per_s": world * B * args.steps / smp_dt, "launch": smp, "power": smp_sens,
                         "forward_residual_rel_err": resid,
                         "note": "the conservative figure: the sampling distribution is the real use of the inverse, and on some "
                                 "boxes this leg runs at a lower clock than the round-trip leg (see `clock`)"},
            "forward": {"ms_per_img": fwd_dt / args.steps / B * 1e3, "images_per_s": world * B * args.steps / fwd_dt,
                        "logdet": 0.0, "launch_ms": fwd_launch_ms, "launch": fwd, "power": fwd_sens,
                        "kernel": conv_form,
                        "frac_hbm_peak": alg_bytes / (fwd_launch_ms * 1e-3) / 1e9 / HBM_PEAK_GBS,
                        "executed_flops_per_launch": fwd_flops,
                        "frac_fp32_peak": fwd_flops / (fwd_launch_ms * 1e-3) / 1e12 / FP32_PEAK_TFLOPS,
                        "direct_equivalent_frac_fp32_peak": alg_flops / (fwd_launch_ms * 1e-3) / 1e12 / FP32_PEAK_TFLOPS,
                        "hbm_frac_ceiling_at_fp32": hbm_ceiling * alg_flops / fwd_flops,
                        "note": f"fp32-compute-bound shape: at 100 % of the fp32 peak the direct sum reaches {hbm_ceiling:.0%} of "
                                f"HBM peak" + (f"; the Winograd F(2,3) kernel executes 2/3 of the direct multiplies, which lifts "
                                               f"that ceiling to {hbm_ceiling * 1.5:.0%}" if conv_form == "winograd" else
                                               f"; the Winograd F(4,3) kernel executes 1/2 of the direct multiplies, which lifts "
                                               f"that ceiling to {hbm_ceiling * 2:.0%}" if conv_form == "winograd4" else
                                               f"; the Winograd F(2,5) kernel executes 3/5 of the direct multiplies, which lifts "
                                               f"that ceiling to {hbm_ceiling / 0.6:.0%}" if conv_form == "winograd25" else "")},
            "training_step": {"what": "z = unit(x); z.backward(gz): forward + grad-input + grad-weight with the corner-tap mask "
                                      "(SURVEY 8 f1), HIP kernels under autograd", "ms_per_step": tr_dt / tr_steps * 1e3,
                              "steps": tr_steps, "launch": launch_stats(tr_per),
                              "grad_weight_kernel": bvar["gradw"],
                              "frac_fp32_peak": (gw_flops + 2 * fwd_flops) / (tr_dt / tr_steps) / 1e12 / FP32_PEAK_TFLOPS,
                              "direct_equivalent_frac_fp32_peak": 3 * alg_flops / (tr_dt / tr_steps) / 1e12 / FP32_PEAK_TFLOPS},
            "roofline": {"kernel": inverse_kernel_name(B, Cq, H, W, K), "bound": "hbm", "achieved": inv_gbs,
                         "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": inv_gbs / HBM_PEAK_GBS,
                         "traffic": (traffic or {}).get("inverse_hbm_bytes_per_launch"),
                         "traffic_source": (traffic or {}).get("source"),
                         "traffic_measured_in_this_run": False,
                         "algorithmic_bytes_per_launch": alg_bytes, "launch_ms": inv_launch_ms,
                         "frac_fp32_peak": alg_flops / (inv_launch_ms * 1e-3) / 1e12 / FP32_PEAK_TFLOPS,
                         # which of the two roofs is the lower one for this workload (time at the roof: bytes / HBM peak against
                         # flops / fp32 MFMA peak); `frac` above stays the HBM figure so that rounds compare, `compute.frac` is
                         # the fraction of the roof that binds when this says "mfma"
                         "binding_roof": "mfma" if alg_flops / (FP32_PEAK_TFLOPS * 1e12) > alg_bytes / (HBM_PEAK_GBS * 1e9) else "hbm",
                         # the shape is compute-bound (K^2*Cq/4 flop/B against a ridge of ~20): the same launch against
                         # the ceiling that actually limits it, the dense fp32 MFMA peak
                         "compute": {"bound": "mfma", "achieved": alg_flops / (inv_launch_ms * 1e-3) / 1e12,
                                     "peak": FP32_PEAK_TFLOPS, "unit": "TFLOP/s",
                                     "frac": alg_flops / (inv_launch_ms * 1e-3) / 1e12 / FP32_PEAK_TFLOPS,
                                     "algorithmic_flops_per_launch": alg_flops},
                         # the recurrence is a chain: one problem needs `dependent_steps` steps one after the other
                         "latency": {"dependent_steps": chain, "us_per_step": inv_launch_ms * 1e3 / chain,
                                     "note": "shapes whose problems do not fill the chip (c2, the c4 units) are bound by "
                                             "dependent_steps x us_per_step, not by bandwidth"}},
        }
        if strong_share is not None:
            for sh in strong_share["shares"]:
                sh["speedup_over_full_batch_implied"] = sh["implied_images_per_s_at_P"] / (B / (inv_launch_ms * 1e-3))
            line["strong_share"] = strong_share
        if clock is not None:
            line["clock"] = clock
        # the CPU baseline is an N = 1 measurement (rank 0 only); at N > 1 the object keeps its shape and says so
        line["cpu_baseline"] = {"value": None, "unit": "images/s", "cores": None, "kind": "port",
                                "sample": "not measured in this run (the CPU baseline is taken at N = 1 only)" if world > 1
                                else "skipped (--no-cpu)"}
        if not args.no_cpu and world == 1:
            line["cpu_baseline"] = cpu_baseline(args.workload, B, C, H, W, K, std,
                                                args.cpu_sample or CPU_SAMPLE_1T[args.workload])
        print(json.dumps(line), flush=True)
    h.finish()


def main(argv=None):
    argv = list(sys.argv[1:] if argv is None else argv)
    args = parse(argv)
    if args.gpus < 1:
        raise SystemExit("--gpus must be >= 1")
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        return launch_ranks(args, argv)             # nothing above this line touches torch or the GPU
    if os.environ.get("FINC_BENCH_STUB") == "1":
        return bench_stub(args)
    if args.workload == "c4":
        return bench_stack(args)
    if args.workload == "ref_timing":
        return bench_ref_timing(args)
    return bench_unit(args)


if __name__ == "__main__":
    sys.exit(main() or 0)
