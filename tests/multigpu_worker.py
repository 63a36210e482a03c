"""Rank program of tests/test_multigpu.py::test_spawned_ranks_exchange_cpu: started by multigpu.spawn_ranks (i.e. by
torch.distributed.run), exchanges synthetic ragged shards of records through multigpu.RecordExchange on CPU tensors over
gloo and writes what rank 0 gathered."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from __graft_entry__ import load_pkg  # noqa: E402


def main():
    out_path, n_total = sys.argv[1], int(sys.argv[2])
    import torch
    import torch.distributed as dist
    pkg = load_pkg()
    mg, sh, abi = pkg.multigpu, pkg.sharding, pkg.abi
    rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
    dist.init_process_group("gloo", rank=rank, world_size=world)
    rng = np.random.default_rng(5)
    full = rng.integers(-2**31, 2**31 - 1, size=(n_total, mg.FULL_WORDS["f64"]), dtype=np.int64).astype(np.int32)
    lo, hi = sh.shard_range(n_total, rank, world)
    per = sh.padded_shard_len(n_total, world)
    x = mg.RecordExchange(None, torch.device("cpu"), per, "f64", "full", dist=dist)
    ok = True
    for step in range(3):  # double buffering: buffers 0, 1, 0
        buf = step & 1
        mine = torch.zeros(per * x.words, dtype=torch.int32)
        mine[:(hi - lo) * x.words] = torch.from_numpy((full[lo:hi] + step).reshape(-1))
        x.before_launch(buf)
        sent = x.after_launch(buf, mine, hi - lo, None)
        x.drain()
        ok = ok and x.verify(buf, sent, rank)
        got = x.gathered[buf][:n_total * x.words].numpy().reshape(n_total, -1)
        ok = ok and np.array_equal(got, full + step)
    pr = x.probe(sent, reps=2)  # what bench.py --gpus N reports about the exchange: the ranks reached and the gather alone
    ok = ok and pr["ranks_seen"] == world and pr["ms"] > 0 and pr["bytes_received_per_rank"] == x.bytes_per_rank_per_step()[1]
    if rank == 0:
        np.save(out_path, np.array([int(ok), world, x.bytes_per_rank_per_step()[1]]))
    dist.barrier()
    dist.destroy_process_group()
    sys.exit(0 if ok else 3)


if __name__ == "__main__":
    main()
