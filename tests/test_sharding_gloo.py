"""N > 1 path on CPU: world_size 2 over gloo.  Each rank evaluates its shard (here with the CPU
oracle standing in for the GPU kernels -- the point is the sharding + all-gather plumbing that
bench.py uses with RCCL), the records are all-gathered and must equal the single-process result."""
import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _worker(rank, world, port, n, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import torch
    import torch.distributed as dist
    import oracle_binding as ob
    pkg = ob._pkg()
    abi, wl, sh = pkg.abi, pkg.workloads, pkg.sharding
    dist.init_process_group("gloo", rank=rank, world_size=world)
    b = wl.cfg5_mixed(n=n, seed=11)
    req = wl.make_request(b, abi)
    lo, hi = sh.shard_range(n, rank, world)
    part = ob.collide_batch(b.shapes, b.verts, b.s1[lo:hi], b.s2[lo:hi], b.tf1[lo:hi], b.tf2[lo:hi], req)
    words = torch.from_numpy(sh.records_to_words(part).copy())
    full = sh.all_gather_records(words, n, abi.RESULT_DTYPE.itemsize // 4, dist)
    rec = full.numpy().view(abi.RESULT_DTYPE)
    if rank == 0:
        q.put(rec.tobytes())
    dist.barrier()
    dist.destroy_process_group()


def test_two_rank_shard_and_all_gather(pkg, oracle):
    import torch.multiprocessing as mp
    abi, wl = pkg.abi, pkg.workloads
    n = 2501  # odd: ragged last shard
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 29500 + (os.getpid() % 2000)
    procs = [ctx.Process(target=_worker, args=(r, 2, port, n, q)) for r in range(2)]
    for p in procs:
        p.start()
    data = q.get(timeout=180)
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    got = np.frombuffer(data, dtype=abi.RESULT_DTYPE)
    b = wl.cfg5_mixed(n=n, seed=11)
    ref = oracle.collide_batch(b.shapes, b.verts, b.s1, b.s2, b.tf1, b.tf2, wl.make_request(b, abi))
    assert len(got) == n
    assert np.array_equal(got["status"], ref["status"])
    assert np.array_equal(np.nan_to_num(got["distance"]), np.nan_to_num(ref["distance"]))


def test_shard_ranges_cover_everything(pkg):
    sh = pkg.sharding
    for n in (0, 1, 7, 8, 1000003):
        for world in (1, 2, 3, 8):
            spans = [sh.shard_range(n, r, world) for r in range(world)]
            assert spans[0][0] == 0 and spans[-1][1] == n
            for (a, b), (c, d) in zip(spans, spans[1:]):
                assert b == c and a <= b and c <= d
