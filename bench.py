#!/usr/bin/env python
"""bench.py -- queries/second of the batched narrow phase on N MI355X (one process per GPU).

A "step" is one pass of the hot path over one batch of synthetic pairs that is already resident
in HBM: classify -> GJK kernels -> EPA kernel, through the C ABI's device-resident entry point.
Default workload = BASELINE.json configs[2] (the configuration north_star's target is quoted on):
1M Convex-Convex (32-vertex hulls, shared 4096-hull library) distance() queries, signed
distance (GJK+EPA), Nesterov acceleration, fp32.  `--workload cfg2` runs configs[1]
(1M Box-Capsule collide(), fp64) instead.

N > 1 (launched by torch.distributed.run): every rank owns its own 1M-pair shard (weak scaling,
different seed per rank) and the per-shard result records are all-gathered over RCCL/xGMI
(north_star), overlapped with the next step's kernels on a separate stream.

Prints ONE JSON line on rank 0.
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
from __graft_entry__ import load_pkg  # noqa: E402

HBM_PEAK_GBS = 8000.0  # MI355X HBM3E peak, /opt/skills/guides/MI355X_MICROARCH.md

# ALGORITHMIC bytes per query (DESIGN.md "Measurement"): compulsory traffic only
BYTES_PER_QUERY = {
    "cfg3": 8 + 2 * 28 + 44,   # 2 shape ids + 2 (quat+T) fp32 poses + 44-B fp32 record = 108 B
    "cfg2": 8 + 2 * 96 + 96,   # 2 shape ids + 2 Transform3f images (fp64) + 96-B fp64 record = 296 B
    # cfg4: ids + poses + record, plus the BV nodes / triangles the *reference DFS* visits
    # (SURVEY.md 8d): 2 x 128 B (fp64 device node) per BV test, 2 x (3 x 24 B vertices + 12 B indices)
    # per leaf test; N_bv, N_leaf are measured with the oracle on a sample and reported.
    "cfg4": 8 + 2 * 96 + 96,
    "cfg5": 8 + 2 * 96 + 96,   # mixed primitive+convex collide, fp64: as cfg2
    # cfg1's shape pair (Sphere-Sphere distance(), closed form) at a batch that is not launch bound, through
    # the general boundary: ids + full Transform3f images + 96-B record (SURVEY's 144 B assumes a
    # sphere-only entry point that reads centres and radii alone; the drop-in ABI does not have one)
    "cfg1": 8 + 2 * 96 + 96,
    "cfg3u": 8 + 2 * 28 + 44 + 2 * 32 * 12,  # cfg3 with one hull pair per query: + 2 x 32 fp32 vertices = 876 B
}
CFG4_BYTES_PER_BV_TEST = 2 * 128
CFG4_BYTES_PER_LEAF_TEST = 2 * (3 * 24 + 12)


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--workload", default="cfg3", choices=["cfg3", "cfg2", "cfg4", "cfg5", "cfg1", "cfg3u"])
    ap.add_argument("--pairs", type=int, default=0, help="pairs per GPU per step (default 1M; cfg4: 100k; cfg5: 1.25M = 10M / 8)")
    ap.add_argument("--no-gather", action="store_true", help="skip the RCCL all-gather of result records (N>1)")
    ap.add_argument("--split", type=int, default=0, help="0: the library decides (two half-batches on two streams for "
                    "mixed libraries); 1: one stream; 2: always split")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-sample", type=int, default=1_000_000)
    return ap.parse_args()


# VALU issue peak: 256 CUs x 4 SIMDs, one wave64 VALU instruction per 4 cycles (16 lanes per SIMD) at the 2.4 GHz
# peak clock (MI355X_MICROARCH.md) = 614 G wave-instructions/s.  Packed / dual-issue forms are not assumed.
VALU_ISSUE_PEAK = 256 * 4 * 2.4e9 / 4


def load_traffic(workload, dominant, n, want="traffic"):
    """HBM bytes per launch of the dominant kernel from the committed PMC pass
    (tools/measure_traffic.py -> profiles/traffic_<workload>.json; FETCH_SIZE / WRITE_SIZE in KB).
    gfx950 correction (MI355X_MICROARCH.md, HBM section): FETCH_SIZE tallies 128-B requests at
    64 B, exact x2 for wide streaming reads and uncalibrated otherwise -> we report
    2 x FETCH + WRITE (upper bound) and keep the raw figures in the note."""
    import re
    path = os.path.join(ROOT, "profiles", "traffic_%s.json" % workload)
    if not os.path.exists(path):
        return None, "no PMC pass committed for this workload"
    t = json.load(open(path))
    default_n = {"cfg4": 100_000, "cfg5": 1_250_000, "cfg1": 4_000_000}.get(workload, 1_000_000)
    if (t.get("pairs") or default_n) != n:
        return None, "PMC pass was taken at a different batch size"
    m = re.match(r"(k_\w+)(?:<(\w+)>)?", dominant)
    base, tag = m.group(1), m.group(2)
    sel = {"fast": ", 1>(", "full": ", 2>(", "cc": ", 0>(", "pc": ", 1>(", "cp": ", 2>("}.get(tag, "")
    for name, v in t["kernels"].items():
        hit = base + "<" in name and sel in name
        if tag == "fast" and "k_epa_stream<" in name:  # the fp32 fast tier is the streaming form of the same kernel
            hit = True
        if base == "k_closed" and "k_closed_staged(" in name:  # the fp64 closed-form kernel (LDS-staged I/O)
            hit = True
        if hit and want == "valu":
            if "SQ_INSTS_VALU_per_dispatch" in v:
                return v["SQ_INSTS_VALU_per_dispatch"], {k: v[k] for k in v if k.startswith("SQ_")}
            continue
        if hit and "FETCH_SIZE_KB_per_dispatch" in v and "WRITE_SIZE_KB_per_dispatch" in v:
            f, w = v["FETCH_SIZE_KB_per_dispatch"] * 1024, v["WRITE_SIZE_KB_per_dispatch"] * 1024
            return 2 * f + w, "PMC per launch: FETCH_SIZE raw %.3g B (x2 gfx950 correction applied), WRITE_SIZE %.3g B; %s" % (f, w, os.path.relpath(path, ROOT))
    return None, "kernel not found in " + os.path.relpath(path, ROOT)


def main():
    args = parse()
    import torch
    import torch.distributed as dist

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if args.gpus != world:
        if world == 1 and args.gpus > 1:
            raise SystemExit("launch with: python -m torch.distributed.run --nproc-per-node %d bench.py --gpus %d" %
                             (args.gpus, args.gpus))
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU (the engine has no CPU fallback)")
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    # HFCL_BENCH_FORCE_DIST=1: run the N>1 code path (RCCL group, comm stream, async all-gather, barriers, max over
    # ranks) with a single rank -- a dry run of the multi-GPU plumbing on a 1-GPU box
    dist_on = world > 1 or os.environ.get("HFCL_BENCH_FORCE_DIST") == "1"
    if dist_on:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29533")
        dist.init_process_group("nccl", device_id=dev, rank=rank, world_size=world)

    pkg = load_pkg()
    abi, wl = pkg.abi, pkg.workloads
    n = args.pairs or {"cfg4": 100_000, "cfg5": 1_250_000, "cfg1": 4_000_000}.get(args.workload, 1_000_000)
    if args.workload == "cfg3":
        batch = wl.cfg3_convex_convex(n=n, seed=1 + rank)
        dtype = "f32"
    elif args.workload == "cfg3u":
        batch = wl.cfg3_unique_hulls(n=n, seed=1 + rank)
        dtype = "f32"
    elif args.workload == "cfg1":
        batch = wl.cfg1_sphere_sphere(n=n, seed=1 + rank)
        dtype = "f64"
    elif args.workload == "cfg2":
        batch = wl.cfg2_box_capsule(n=n, seed=1 + rank)
        dtype = "f64"
    elif args.workload == "cfg5":
        # pair list = host broadphase over a scene of n/10 posed objects (not in the timed region);
        # every rank owns its own scene shard; the list is cut to exactly n pairs
        over = 1.15
        while True:
            t_bp = time.perf_counter()
            batch = wl.cfg5_broadphase_scene(n_objects=max(n // 10, 100), target_pairs=int(over * n), seed=1 + rank)
            t_bp = time.perf_counter() - t_bp
            if len(batch) >= n:
                break
            over *= 1.3
        scene = {"objects": batch.scene["n_objects"], "broadphase_pairs": len(batch), "host_broadphase_s": t_bp}
        batch = batch.slice(0, n)
        batch.scene = scene
        dtype = "f64"
    else:
        batch = wl.cfg4_mesh_mesh(n=n, seed=1 + rank)
        dtype = "f64"
    req = wl.make_request(batch, abi)
    lib = wl.make_library(pkg, batch, device=local_rank)
    if args.split:
        lib.set_split(args.split)

    d_s1 = torch.from_numpy(batch.s1.astype(np.int32)).to(dev)
    d_s2 = torch.from_numpy(batch.s2.astype(np.int32)).to(dev)
    if dtype == "f32":
        d_p1 = torch.from_numpy(batch.pose1_f32).to(dev)
        d_p2 = torch.from_numpy(batch.pose2_f32).to(dev)
        rec_words = 11
        launch = lib.distance_device_f32 if batch.kind == "distance" else lib.collide_device_f32
    else:
        d_p1 = torch.from_numpy(batch.tf1).to(dev)
        d_p2 = torch.from_numpy(batch.tf2).to(dev)
        rec_words = 24
        launch = lib.distance_device if batch.kind == "distance" else lib.collide_device
    outs = [torch.zeros(n * rec_words, dtype=torch.int32, device=dev) for _ in range(2)]
    gather = dist_on and not args.no_gather
    gathered = [torch.empty(world * n * rec_words, dtype=torch.int32, device=dev) for _ in range(2)] if gather else None
    stream = torch.cuda.current_stream()

    kernel_ms = {}

    def one_step(i, record_times):
        buf = i & 1
        launch(d_s1, d_s2, d_p1, d_p2, n, req, outs[buf], stream=stream.cuda_stream)
        work = None
        if gather:
            # results of this step travel over xGMI while the next step's kernels run
            ev = torch.cuda.Event()
            ev.record(stream)
            work = (buf, ev)
        if record_times:
            for name, ms in lib.last_kernel_breakdown():  # HIP events on the launch stream
                kernel_ms.setdefault(name, []).append(ms)
        return work

    comm_stream = torch.cuda.Stream(device=dev) if gather else None

    def flush_gather(work):
        buf, ev = work
        with torch.cuda.stream(comm_stream):
            comm_stream.wait_event(ev)
            h = dist.all_gather_into_tensor(gathered[buf], outs[buf], async_op=True)
        return h

    def sync_all():
        torch.cuda.synchronize()
        if dist_on:
            dist.barrier()
        torch.cuda.synchronize()

    lib.set_kernel_timing(False)  # the timed region runs without the per-kernel event markers
    for i in range(args.warmup):
        w = one_step(i, False)
        if w:
            flush_gather(w).wait()
    sync_all()
    inflight = {}  # result buffer -> all-gather still reading it
    t0 = time.perf_counter()
    for i in range(args.steps):
        # double buffering: the kernels of step i overwrite the buffer the all-gather of step i-2 read;
        # make the launch stream wait for that collective first (stream-side wait, the host does not block)
        h = inflight.pop(i & 1, None)
        if h is not None:
            h.wait()
        w = one_step(i, False)
        if w:
            inflight[i & 1] = flush_gather(w)
    for h in inflight.values():
        h.wait()
    sync_all()
    elapsed = time.perf_counter() - t0
    if dist_on:
        t = torch.tensor([elapsed], dtype=torch.float64, device=dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())

    # per-kernel durations (HIP events inside the library, on the launch stream), separate pass so
    # the event reads do not serialise the timed region
    lib.set_kernel_timing(True)
    for i in range(min(args.steps, 10)):
        one_step(i, True)
    torch.cuda.synchronize()

    if rank == 0:
        res = outs[(args.steps - 1) & 1].cpu().numpy()
        status = res.view(abi.RESULT_F32_DTYPE if dtype == "f32" else abi.RESULT_DTYPE)["status"]
        contact_frac = float(abi.status_contact(status).mean())
        buckets = lib.last_bucket_counts()
        total_q = args.steps * n * world
        qps = total_q / elapsed
        ms_per_step = 1e3 * elapsed / args.steps
        avg = {k: float(np.mean(v)) for k, v in kernel_ms.items() if np.mean(v) > 0}
        dominant = max(avg, key=avg.get) if avg else ""
        bpq = BYTES_PER_QUERY[args.workload]
        extra_cfg = {}
        if args.workload == "cfg4":
            sys.path.insert(0, os.path.join(ROOT, "tests"))
            import oracle_binding as ob0  # measures N_bv / N_leaf of the reference DFS on a sample
            ns0 = min(2000, n)
            ML = pkg.bvh_builder.MeshLibrary(batch.meshes)
            _, st0 = ob0.bvh_collide_batch(ML, batch.s1[:ns0], batch.s2[:ns0], batch.tf1[:ns0], batch.tf2[:ns0], req,
                                           n_threads=os.cpu_count() or 1, want_stats=True)
            nbv, nleaf = float(st0[:, 0].mean()), float(st0[:, 1].mean())
            bpq = bpq + CFG4_BYTES_PER_BV_TEST * nbv + CFG4_BYTES_PER_LEAF_TEST * nleaf
            extra_cfg = {"mean_bv_tests": nbv, "mean_leaf_tests": nleaf, "stats_sample": ns0}
        if args.workload == "cfg5":
            extra_cfg = dict(batch.scene)
        # units the dominant kernel processes in one launch
        if dominant.startswith("k_epa<fast"):
            units = buckets["epa_queue"]
        elif dominant.startswith("k_epa<full"):
            units = buckets["epa_overflow"]
        else:
            units = n
        split = lib.last_split_parts()
        units = units / split  # a launch covers one half of a split batch
        dom_ms = avg.get(dominant, float("nan"))
        achieved = (units * bpq) / (dom_ms * 1e-3) / 1e9 if dom_ms == dom_ms and dom_ms > 0 else None
        pipeline_ms = float(sum(avg.values()))
        traffic, traffic_note = load_traffic(args.workload, dominant, n)
        valu_insts, sq = load_traffic(args.workload, dominant, n, want="valu")
        valu = None
        if valu_insts and dom_ms == dom_ms and dom_ms > 0:
            # the bound that actually applies to the iterative kernels: wave-level VALU instructions issued per
            # launch (SQ_INSTS_VALU, committed PMC pass) against the issue peak over the live kernel duration
            valu = {"insts_per_launch": valu_insts, "issue_peak_per_s": VALU_ISSUE_PEAK,
                    "frac": valu_insts / (dom_ms * 1e-3) / VALU_ISSUE_PEAK, "sq_counters_per_launch": sq,
                    "note": "wave-level VALU instructions / (kernel time x 256 CUs x 4 SIMDs x 2.4 GHz / 4); "
                            "fp64 arithmetic issues at half that rate, so an fp64 kernel saturates near 0.5"}
        roofline = {
            "bound": "hbm", "kernel": dominant, "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
            "frac": (achieved / HBM_PEAK_GBS) if achieved else None, "traffic": traffic, "traffic_note": traffic_note,
            "bytes_per_query": bpq, "units_per_launch": units, "kernel_ms": dom_ms,
            "pipeline_ms": pipeline_ms, "pipeline_achieved": (n * bpq) / (ms_per_step * 1e-3) / 1e9,
            "kernels_ms": avg, "valu_issue": valu,
        }
        cpu = None
        if not args.no_cpu_baseline and world == 1:  # reported on rank 0 at N=1 only
            sys.path.insert(0, os.path.join(ROOT, "tests"))
            import oracle_binding as ob  # checker/baseline only -- never on the product path
            ns = min(args.cpu_sample, n)
            sb = batch.slice(0, ns)
            tf1, tf2 = sb.tf1, sb.tf2
            if args.workload == "cfg4":
                MLc = pkg.bvh_builder.MeshLibrary(batch.meshes)

                def run_cpu(lo, hi, threads):
                    ob.bvh_collide_batch(MLc, sb.s1[lo:hi], sb.s2[lo:hi], tf1[lo:hi], tf2[lo:hi], req, n_threads=threads)
            else:
                fn = ob.distance_batch if sb.kind == "distance" else ob.collide_batch

                def run_cpu(lo, hi, threads):
                    fn(sb.shapes, sb.verts, sb.s1[lo:hi], sb.s2[lo:hi], tf1[lo:hi], tf2[lo:hi], req, n_threads=threads)
            run_cpu(0, min(1000, ns), 1)  # warm-up
            reps, t_cpu = 0, 0.0
            while t_cpu < 10.0 and reps < 5:
                t1 = time.perf_counter()
                run_cpu(0, ns, 1)
                t_cpu += time.perf_counter() - t1
                reps += 1
            cores_all = os.cpu_count() or 1
            t1 = time.perf_counter()
            run_cpu(0, ns, cores_all)
            t_all = time.perf_counter() - t1
            cpu = {"value": reps * ns / t_cpu, "unit": "queries/s", "cores": 1, "kind": "port",
                   "sample": "%d pairs of the same workload x %d repeats, fp64 CPU oracle (oracle/), 1 thread" % (ns, reps),
                   "all_cores": {"value": ns / t_all, "cores": cores_all}}
        line = {
            "metric": "narrow-phase queries/s (collision+distance)", "value": qps, "unit": "queries/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": ms_per_step,
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": dtype, "data": "synthetic",
            "config": {"workload": batch.name, **extra_cfg,
                       "baseline_config": {"cfg3": "configs[2]", "cfg2": "configs[1]", "cfg4": "configs[3]", "cfg5": "configs[4]",
                                           "cfg1": "configs[0] (shape pair; GPU batch size)",
                                           "cfg3u": "configs[2], one hull pair per query"}[args.workload],
                       "pairs_per_gpu_per_step": n, "contact_fraction": contact_frac, "buckets": buckets,
                       "request": batch.kind, "all_gather_results": bool(gather), "split_parts": lib.last_split_parts(),
                       "lane_group_width": os.environ.get("HFCL_CVX_W", "auto (2; fp64 convex-convex 4)")},
            "roofline": roofline, "cpu_baseline": cpu,
        }
    else:
        line = None
    if dist_on:
        dist.barrier()
        dist.destroy_process_group()
    lib.close()
    if line is not None:
        # the JSON line goes out last: RCCL writes a version banner to the C stdout, which would otherwise be
        # flushed after Python's buffer when stdout is a file or pipe
        import ctypes
        try:
            ctypes.CDLL(None).fflush(None)
        except Exception:
            pass
        sys.stdout.flush()
        print(json.dumps(line), flush=True)


if __name__ == "__main__":
    main()
