"""The N>1 host logic on CPU: world_size 2 over gloo (rendezvous on 127.0.0.1).  No kernels are launched here;
what is covered is exactly what bench.py adds for N>1 -- batch slices, the one weight broadcast, the
max-over-ranks timing rule, and the optional gather."""
import os

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from fincflow_amd import FastFlowUnit
from fincflow_amd import dist as fdist


def test_shard_bounds_cover_the_batch():
    for n in (0, 1, 7, 256, 257):
        for world in (1, 2, 3, 8):
            parts = [fdist.shard_bounds(n, r, world) for r in range(world)]
            assert parts[0][0] == 0 and parts[-1][1] == n
            assert all(a[1] == b[0] for a, b in zip(parts, parts[1:]))
            sizes = [hi - lo for lo, hi in parts]
            assert max(sizes) - min(sizes) <= 1
    with pytest.raises(ValueError):
        fdist.shard_bounds(4, 2, 2)


def _worker(rank, world, port, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        torch.manual_seed(100 + rank)                      # ranks start with DIFFERENT weights
        unit = FastFlowUnit(8, 8, 3)
        before = [w.detach().clone() for w in unit._weights()]
        unit._cache._bank(torch.device("cpu")).key = ("stale",)   # pretend the sampling cache was built before the broadcast
        fdist.broadcast_weights(unit, src=0)
        after = [w.detach().clone() for w in unit._weights()]
        bumped = not unit._cache._banks                    # ... and check the broadcast dropped it
        full = torch.arange(7 * 3, dtype=torch.float32).reshape(7, 3)
        mine = fdist.shard_batch(full)
        gathered = fdist.gather_shards(mine * 2, 7)
        tmax = fdist.max_over_ranks(1.0 + rank)
        q.put((rank, [a.numpy() for a in after], [b.numpy() for b in before], tuple(mine.shape), gathered.numpy(),
               tmax, bumped))
    finally:
        dist.destroy_process_group()


def test_two_ranks_gloo():
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 29500 + os.getpid() % 2000
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = sorted([q.get(timeout=120) for _ in procs], key=lambda r: r[0])
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    (r0, a0, b0, s0, g0, t0, v0), (r1, a1, b1, s1, g1, t1, v1) = res
    import numpy as np
    assert all(np.array_equal(x, y) for x, y in zip(a0, a1)), "weights differ after broadcast"
    assert all(np.array_equal(x, y) for x, y in zip(a0, b0)), "rank 0 must keep its weights"
    assert not all(np.array_equal(x, y) for x, y in zip(b0, b1)), "test needs different initial weights"
    assert s0 == (4, 3) and s1 == (3, 3)
    expect = np.arange(21, dtype=np.float32).reshape(7, 3) * 2
    assert np.array_equal(g0, expect) and np.array_equal(g1, expect)
    assert t0 == 2.0 and t1 == 2.0
    assert v0 and v1
