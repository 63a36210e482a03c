"""One process per GPU: launching the ranks and exchanging per-shard result records.

Pairs are independent, so the only inter-GPU traffic of the path is the exchange of *results*: north_star asks for an
all-gather of the per-shard CollisionResult buffers over RCCL/xGMI.  `RecordExchange` does that on a side stream, double
buffered, so that the records of step i travel while the kernels of step i+1 run.  Three forms (`mode`):
  full     the 96-B (fp32 path: 44-B) records as the kernels wrote them
  compact  hfcl_result_compact{,_f32} (24 / 8 B: distance, b1, b2, status, num_contacts), packed on the device by
           hfcl_compact_results_device -- what a caller that folds isCollision() / min_distance needs
  none     every rank keeps its shard (results are read where they were computed)
Backends: "nccl" (= RCCL; device buffers go straight into the collective) and "gloo" (device buffers are staged through
host memory; used to run two ranks on ONE GPU in the tests -- RCCL refuses two ranks on the same device).

Plumbing only: torch supplies device memory, streams and the process group; the records come from the C ABI."""
import os
import socket
import subprocess
import sys

import numpy as np

from . import sharding

GATHER_MODES = ("full", "compact", "none")
FULL_WORDS = {"f64": 24, "f32": 11}      # int32 words of a full record (hfcl_result / hfcl_result_f32)
COMPACT_WORDS = {"f64": 6, "f32": 2}     # ... of hfcl_result_compact / hfcl_result_compact_f32


def record_words(dtype, mode):
    return COMPACT_WORDS[dtype] if mode == "compact" else FULL_WORDS[dtype]


def free_port():
    s = socket.socket(socket.AF_INET, socket.SOCK_STREAM)
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    return port


def rank_command(script, argv, n_ranks, port=None, python=None):
    """The command that starts `n_ranks` processes of `script` on this node, one per GPU: torch.distributed.run with a
    loopback rendezvous (the container hostname may not resolve)."""
    return [python or sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(int(n_ranks)),
            "--master-addr", "127.0.0.1", "--master-port", str(port or free_port()), script] + list(argv)


def spawn_ranks(script, argv, n_ranks, env=None, timeout=None):
    """Run `script argv` as n_ranks ranks and pass their stdout / stderr through; returns the exit code.  Used by
    `bench.py --gpus N` when it was started as a plain process (no WORLD_SIZE in the environment)."""
    e = dict(os.environ if env is None else env)
    e.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")  # dmabuf IPC: the host driver supports nothing else (RCCL needs it)
    e.setdefault("OMP_NUM_THREADS", str(max(1, (os.cpu_count() or 1) // max(1, int(n_ranks)))))
    return subprocess.call(rank_command(script, argv, n_ranks), env=e, timeout=timeout)


def device_of_rank(local_rank, device_map=None):
    """Device index of a local rank.  `device_map` ("0,0" or [0, 0]) lets several ranks share a device (tests)."""
    if not device_map:
        return int(local_rank)
    m = [int(x) for x in device_map.split(",")] if isinstance(device_map, str) else [int(x) for x in device_map]
    return m[int(local_rank) % len(m)]


def words_checksum(t):
    """Order-independent fingerprint of an int32 word tensor (sum of the words as int64)."""
    return int(t.to(dtype=__import__("torch").int64).sum().item())


class _Done:
    def wait(self):
        return True


class RecordExchange:
    """Double-buffered exchange of one rank's result records with all ranks.

    lib: engine.Library of this rank; dev: torch device; per: records every rank contributes per step (shards padded to
    equal length); dtype: "f64" / "f32"; mode: full / compact / none; group: torch.distributed module (initialised) or
    None for a single process."""

    def __init__(self, lib, dev, per, dtype, mode, dist=None, staged=False):
        import torch
        assert mode in GATHER_MODES
        self.torch, self.lib, self.dev, self.per, self.dtype, self.mode = torch, lib, dev, int(per), dtype, mode
        self.dist = dist if (dist is not None and mode != "none") else None
        self.world = dist.get_world_size() if dist is not None else 1
        self.staged = staged
        self.words = record_words(dtype, mode)
        z = lambda n: torch.zeros(n, dtype=torch.int32, device=dev)  # noqa: E731
        self.packed = [z(self.per * self.words) for _ in range(2)] if mode == "compact" else None
        self.gathered = [torch.empty(self.world * self.per * self.words, dtype=torch.int32, device=dev) for _ in range(2)] \
            if self.dist is not None else None
        self.on_cpu = torch.device(dev).type == "cpu"  # CPU tensors (tests of the exchange logic without a GPU)
        self.comm_stream = torch.cuda.Stream(device=dev) if (self.dist is not None and not self.on_cpu) else None
        self.inflight = {}

    def bytes_per_rank_per_step(self):
        """(sent, received) bytes of one rank in one step."""
        if self.dist is None:
            return 0, 0
        b = self.per * self.words * 4
        return b, b * (self.world - 1)

    def before_launch(self, buf, stream=None):
        """The kernels of this step overwrite the buffer the exchange of two steps ago read: order them after it.  `stream`:
        the torch stream the kernels of this step are launched on (None: torch's current stream).  Work.wait() of an RCCL
        collective makes the CURRENT stream wait for it (the host does not block), so the wait is issued with the launch
        stream current -- a caller that launches on another stream than torch's current one is ordered as well."""
        h = self.inflight.pop(buf, None)
        if h is None:
            return
        if stream is not None and not self.on_cpu and not self.staged:
            with self.torch.cuda.stream(stream):
                h.wait()
        else:
            h.wait()

    def after_launch(self, buf, records, n_valid, stream):
        """records: this rank's full records of the step (int32 words, `per` records, the first n_valid computed);
        packs them if asked and starts the exchange.  Returns the tensor that travels."""
        torch = self.torch
        send = records
        if self.mode == "compact":
            send = self.packed[buf]
            if n_valid:
                self.lib.compact_results_device(records, n_valid, send, f32=self.dtype == "f32", stream=stream.cuda_stream)
            if n_valid < self.per:  # rows past this rank's shard travel as zeros, never as the rows of an earlier, longer step
                if self.on_cpu:
                    send[n_valid * self.words:].zero_()
                else:
                    with torch.cuda.stream(stream):
                        send[n_valid * self.words:].zero_()
        if self.dist is None:
            return send
        if self.on_cpu:
            self.dist.all_gather_into_tensor(self.gathered[buf], send)
            self.inflight[buf] = _Done()
            return send
        ev = torch.cuda.Event()
        ev.record(stream)
        if self.staged:  # gloo: through host memory, synchronous (test path)
            ev.synchronize()
            host = send.cpu()
            out = torch.empty(self.world * host.numel(), dtype=torch.int32)
            self.dist.all_gather_into_tensor(out, host)
            self.gathered[buf].copy_(out)
            self.inflight[buf] = _Done()
        else:
            with torch.cuda.stream(self.comm_stream):
                self.comm_stream.wait_event(ev)
                self.inflight[buf] = self.dist.all_gather_into_tensor(self.gathered[buf], send, async_op=True)
        return send

    def probe(self, send, reps=5):
        """What the exchange does on its own, OUTSIDE the timed region of a run: the ranks the backend actually reaches (an all-reduce of
        ones: `ranks_seen` must equal the world size or some rank sits in another communicator) and the duration of the all-gather of
        one step's records alone (device events on the communication stream; host clock for the staged / CPU forms), best of `reps`.
        Returns {"ranks_seen", "ms", "bus_GBps", "bytes_sent_per_rank", "bytes_received_per_rank"}; bus_GBps = bytes one rank RECEIVES
        per second (what its incoming xGMI links carry: DESIGN.md section 5 holds it against 7 links x 153 GB/s)."""
        if self.dist is None:
            return None
        import time
        torch = self.torch
        host = self.staged or self.on_cpu
        one = torch.ones(1, dtype=torch.int32, device="cpu" if host else self.dev)
        self.dist.all_reduce(one)
        sent, received = self.bytes_per_rank_per_step()
        best = None
        for _ in range(reps):
            if host:
                t0 = time.perf_counter()
                src = send.cpu()
                out = torch.empty(self.world * src.numel(), dtype=torch.int32)
                self.dist.all_gather_into_tensor(out, src)
                ms = 1e3 * (time.perf_counter() - t0)
            else:
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                torch.cuda.synchronize()
                self.dist.barrier()
                with torch.cuda.stream(self.comm_stream):
                    e0.record(self.comm_stream)
                    self.dist.all_gather_into_tensor(self.gathered[0], send)
                    e1.record(self.comm_stream)
                e1.synchronize()
                ms = e0.elapsed_time(e1)
            best = ms if best is None else min(best, ms)
        t = torch.tensor([best], dtype=torch.float64, device="cpu" if host else self.dev)
        self.dist.all_reduce(t, op=self.dist.ReduceOp.MAX)  # the slowest rank's best
        ms = float(t.item())
        return {"ranks_seen": int(one.item()), "ms": ms, "bus_GBps": (received / (ms * 1e-3) / 1e9) if ms > 0 else None,
                "bytes_sent_per_rank": sent, "bytes_received_per_rank": received}

    def drain(self):
        for h in self.inflight.values():
            h.wait()
        self.inflight = {}

    def verify(self, buf, send, rank):
        """Checksum of checksums: every rank fingerprints what it sent, the fingerprints are all-gathered, and each
        rank checks every block of what it received against them.  Returns True / False (None without exchange)."""
        if self.dist is None:
            return None
        torch = self.torch
        mine = torch.tensor([words_checksum(send)], dtype=torch.int64, device="cpu" if (self.staged or self.on_cpu) else self.dev)
        allc = torch.empty(self.world, dtype=torch.int64, device=mine.device)
        self.dist.all_gather_into_tensor(allc, mine)
        g = self.gathered[buf].view(self.world, -1)
        got = g.to(dtype=torch.int64).sum(dim=1).cpu()
        ok = bool(torch.equal(got, allc.cpu()))
        own = bool(torch.equal(g[rank], send))
        return ok and own


def expected_exchange(records_full, dtype, mode):
    """What the gathered buffer of a strong-scaling step must hold, from the single-process records of the whole list
    (numpy structured array): int32 words."""
    from . import abi
    r = abi.compact_records(records_full) if mode == "compact" else records_full
    return sharding.records_to_words(r)
