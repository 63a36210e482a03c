#!/usr/bin/env python
"""bench.py -- queries/second of the batched narrow phase on N MI355X (one process per GPU).

A "step" is one pass of the hot path over one batch of synthetic pairs that is already resident
in HBM: classify -> GJK kernels -> EPA kernels (or the BVH traversal), through the C ABI's device-resident entry point.

Headline (the `value` of the JSON line) = BASELINE.json configs[2], the configuration north_star's target is quoted on:
1M Convex-Convex (32-vertex hulls, shared 4096-hull library) distance() queries, signed distance (GJK+EPA), Nesterov
acceleration, fp32.  Without `--workload` the line also carries `secondary`: the other BASELINE configurations, each with
its own timed region, roofline and (trimmed) CPU baseline --
  cfg2  1M Box-Capsule collide(), fp64 (+ the same batch through the host-buffer boundary, `host_buffers`)
  cfg4  100k BVHModel<OBBRSS> mesh-mesh collide(), fp64
  cfg5  1.25M (= 10M / 8) mixed pairs from the host broadphase, fp64, per GPU (weak)
  cfgmix  200k pairs of a scene of meshes AND solids in one batch (solid x solid, mesh x solid, mesh x mesh), fp64
  cfg5_strong  ONE 10M-pair list from the host broadphase, sharded over the ranks with sharding.shard_range and its
        real records all-gathered (north_star's configs[4] as written; `--scaling strong --workload cfg5 --pairs 10000000`
        runs it as the headline)
`--workload X` runs X alone as the headline (used by the profiling tools).

N > 1: one process per GPU.  `python bench.py --gpus N` started as a plain process spawns its N ranks itself
(torch.distributed.run, loopback rendezvous); started by torch.distributed.run (WORLD_SIZE in the environment) it is one
of the ranks.  Weak scaling -- every rank owns its own shard (different seed per rank) -- and the per-shard result records
are all-gathered over RCCL/xGMI (north_star) on a separate stream, overlapped with the next step's kernels
(`--gather full|compact|none`: 96/44-B records, 24/8-B hfcl_result_compact records, no exchange; DESIGN.md section 5 has the
byte budget).  `--scaling strong`: one list, rank r owns sharding.shard_range(n, r, world).
`--backend gloo --device-map 0,0` runs two ranks on one GPU (tests; RCCL refuses duplicate devices).

Rank 0 prints ONE compact JSON line (<= 4 KB: the contract's fields, the headline's `roofline` and `cpu_baseline`, the
secondary workloads as short rows) as the LAST line of stdout; the full record goes to bench_full.json.
"""
import argparse
import hashlib
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
from __graft_entry__ import load_pkg  # noqa: E402

HBM_PEAK_GBS = 8000.0  # MI355X HBM3E peak, /opt/skills/guides/MI355X_MICROARCH.md
XGMI_IN_GBPS = 7 * 76.8  # what a rank can receive in an 8-GPU all-gather: 7 point-to-point xGMI links x 153.6 GB/s both directions together = 76.8 GB/s inbound each
# Link rates of the GPU box between PAGEABLE host arrays and the device, measured with tools/link_probe.hip
# (profiles/r03_c): one direction at a time / both at once from two host threads (46.5 - 50 GB/s per direction)
LINK_H2D_GBS, LINK_D2H_GBS, LINK_BOTH_GBS = 57.4, 57.0, 2 * 47.0

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
    "cfg2f": 8 + 2 * 28 + 44,  # cfg2's pairs through the fp32 device path (7-float poses, 44-B records)
    # cfg4's distance() variant and mesh x solid collide(): ids + poses + record, plus what the reference's walk
    # visits -- num_bv_tests / num_leaf_tests of its traversal node (traversal_node_bvhs.h:126-128,380-381,423,434), measured
    # with the oracle on a sample as for cfg4
    "cfg4d": 8 + 2 * 96 + 96,
    "cfg4s": 8 + 2 * 96 + 96,
    # a scene of meshes AND solids in one batch (40 % solid x solid, 40 % mesh x solid, 20 % mesh x mesh): ids + poses + record, plus the walk
    # bytes of the mesh pairs averaged over all pairs (mixed_collide_batch's statistics count the mesh x solid walks)
    "cfgmix": 8 + 2 * 96 + 96,
}
# per BV test / per leaf test of the reference's walk: cfg4 two 128-B OBB node records / two triangles (3 x 24 B vertices + 12 B
# of indices each); cfg4d two 128-B RSS node records (DNodeD) / two triangles; cfg4s ONE mesh node record (the solid's box is
# computed from its 56-B shape record, counted once in the 296 B) / one triangle
CFG4_BYTES_PER_BV_TEST = 2 * 128
CFG4_BYTES_PER_LEAF_TEST = 2 * (3 * 24 + 12)
WALK_BYTES = {"cfg4": (CFG4_BYTES_PER_BV_TEST, CFG4_BYTES_PER_LEAF_TEST), "cfg4d": (2 * 128, 2 * (3 * 24 + 12)), "cfg4s": (128, 3 * 24 + 12), "cfgmix": (128, 3 * 24 + 12)}
DEFAULT_PAIRS = {"cfg4": 100_000, "cfg4d": 100_000, "cfg4s": 100_000, "cfg5": 1_250_000, "cfg1": 4_000_000, "cfgmix": 200_000}
BASELINE_CONFIG = {"cfg3": "configs[2]", "cfg2": "configs[1]", "cfg4": "configs[3]", "cfg5": "configs[4]",
                   "cfg1": "configs[0] (shape pair; GPU batch size)", "cfg3u": "configs[2], one hull pair per query",
                   "cfg2f": "configs[1] through the fp32 device path", "cfg4d": "configs[3], distance() instead of collide()",
                   "cfg4s": "configs[3]'s models against convex solids (SURVEY.md 8 f3: BVH x primitive traversal), six solid kinds mixed",
                   "cfgmix": "a scene of configs[3]'s models and configs[4]'s solid kinds in ONE batch: solid x solid, mesh x solid, mesh x mesh"}

# VALU issue peak, MEASURED on the box (tools/valu_peak.hip, profiles/r02_a_valu_issue_peak.txt): the select / compare /
# fma mix these kernels are made of tops out at 1.00e12 wave64 instructions per second chip-wide with 8 waves per SIMD
# (0.96e12 with 2; pure v_fma_f32: 0.88e12, the clock drops to ~2.0 GHz; fp64 FMA: 0.56e12).  The guide's nominal figure
# (a wave64 VALU op every 2 clocks at 2.4 GHz) is 1.229e12.  Round 1 priced against 0.614e12 (4 clocks): wrong.
VALU_ISSUE_PEAK = 1.0e12


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--workload", default=None, choices=["cfg3", "cfg2", "cfg4", "cfg5", "cfg1", "cfg3u", "cfg2f", "cfg4d", "cfg4s", "cfgmix"],
                    help="run this workload alone as the headline (default: cfg3 + the secondary list)")
    ap.add_argument("--pairs", type=int, default=0, help="pairs per GPU per step (default 1M; cfg4: 100k; cfg5: 1.25M = 10M / 8); "
                    "with --scaling strong: pairs of the whole job")
    ap.add_argument("--scaling", default="weak", choices=["weak", "strong"])
    ap.add_argument("--gather", default="full", choices=["full", "compact", "none"],
                    help="N>1: what the ranks exchange after every step -- full result records (north_star: CollisionResult "
                    "buffers), hfcl_result_compact records (24 / 8 B), or nothing")
    ap.add_argument("--no-gather", action="store_true", help="same as --gather none")
    ap.add_argument("--backend", default="nccl", choices=["nccl", "gloo"], help="process-group backend (nccl = RCCL; gloo "
                    "stages the records through host memory and allows several ranks per device)")
    ap.add_argument("--device-map", default="", help="comma list: device of each local rank (default: rank r -> device r)")
    ap.add_argument("--no-verify-gather", dest="verify_gather", action="store_false", default=None, help="skip the check below")
    ap.add_argument("--verify-gather", action="store_true", default=None, help="(default when WORLD_SIZE > 1) after the timed region check what the exchange "
                    "delivered: block checksums of every rank; --scaling strong: bytes equal to the whole list run on rank 0")
    ap.add_argument("--split", type=int, default=0, help="0: the library decides (two half-batches on two streams for "
                    "mixed libraries); 1: one stream; 2: always split")
    ap.add_argument("--two-streams", action="store_true", help="with --workload: even / odd steps on two streams (two batches in "
                    "flight); a throughput figure beside the one-stream headline, never instead of it")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-secondary", action="store_true")
    ap.add_argument("--host-buffers", action="store_true", help="with --workload cfg2/cfg5/...: also time the batch through the host-buffer boundary")
    ap.add_argument("--cpu-sample", type=int, default=1_000_000)
    return ap.parse_args()


def kernel_source_sha():
    """Fingerprint of the device code: a committed PMC pass is only quoted for the code it was taken on."""
    h = hashlib.sha256()
    d = os.path.join(ROOT, "hpp-fcl_amd", "csrc")
    for f in sorted(os.listdir(d)):
        if f.endswith((".hip", ".hpp")) or f == "Makefile":  # (the per-unit compiler flags are part of the code)
            h.update(f.encode())
            h.update(open(os.path.join(d, f), "rb").read())
    return h.hexdigest()[:16]


def load_traffic(workload, dominant, n, want="traffic"):
    """HBM bytes per launch of the dominant kernel from the committed PMC pass
    (tools/measure_traffic.py -> profiles/traffic_<workload>.json; FETCH_SIZE / WRITE_SIZE in KB).
    gfx950 correction (MI355X_MICROARCH.md, HBM section): FETCH_SIZE tallies 128-B requests at
    64 B, exact x2 for wide streaming reads and uncalibrated otherwise -> we report
    2 x FETCH + WRITE (upper bound) and keep the raw figures in the note.
    The pass is only used when it was taken on the present device code (`source_sha`) and batch size: a kernel change that
    was not re-profiled yields traffic = null, never stale counters."""
    import re
    path = os.path.join(ROOT, "profiles", "traffic_%s.json" % workload)
    if not os.path.exists(path):
        return None, "no PMC pass committed for this workload"
    t = json.load(open(path))
    if (t.get("pairs") or DEFAULT_PAIRS.get(workload, 1_000_000)) != n:
        return None, "PMC pass was taken at a different batch size"
    if t.get("source_sha") != kernel_source_sha():
        return None, "committed PMC pass (%s) was taken on other device code (source_sha %s, now %s): re-run tools/measure_traffic.py" % (
            os.path.relpath(path, ROOT), t.get("source_sha"), kernel_source_sha())
    # a timer label can stand for several kernels launched back to back (the fp64 fast EPA tier is one kernel per class of pairs; the
    # mesh timers stand for a pass: the walk, its leaves, the continuation of the suspended queries ...): their counters add up
    tot_f = tot_w = tot_v = 0.0
    sq = {}
    found = False
    for name, v in t["kernels"].items():
        hit = _label_matches(dominant, name)
        if not hit:
            continue
        if want == "valu":
            if "SQ_INSTS_VALU_per_dispatch" in v:
                found = True
                tot_v += v["SQ_INSTS_VALU_per_dispatch"]
                for k in v:
                    if k.startswith("SQ_"):
                        sq[k] = sq.get(k, 0.0) + v[k]
        elif "FETCH_SIZE_KB_per_dispatch" in v and "WRITE_SIZE_KB_per_dispatch" in v:
            found = True
            tot_f += v["FETCH_SIZE_KB_per_dispatch"] * 1024
            tot_w += v["WRITE_SIZE_KB_per_dispatch"] * 1024
    if found and want == "valu":
        return tot_v, sq
    if found:
        return 2 * tot_f + tot_w, "PMC per launch: FETCH_SIZE raw %.3g B (x2 gfx950 correction applied), WRITE_SIZE %.3g B; %s" % (
            tot_f, tot_w, os.path.relpath(path, ROOT))
    return None, "kernel not found in " + os.path.relpath(path, ROOT)


def _label_matches(label, name):
    """Does the rocprofv3 kernel name `name` run under the library's timer label `label`?  (the rules of load_traffic, names only)"""
    import re
    m = re.match(r"(k_\w+)(?:<(\w+)>)?", label)
    if not m:
        return False
    base, tag = m.group(1), m.group(2)
    sel = {"fast": r"k_epa<\w+, \d+, \d+, 1[,>]", "full": r"k_epa<\w+, \d+, \d+, 2[,>]", "cc": r"k_gjk_cvx(?:64)?<\d+, 0, ",
           "pc": r"k_gjk_cvx(?:64)?<\d+, 1, ", "cp": r"k_gjk_cvx(?:64)?<\d+, 2, "}.get(tag, "")
    hit = any(base + suffix in name for suffix in ("<", "64<", "(")) and re.search(sel, name) is not None
    if tag == "fast" and ("k_epa_stream<" in name or "k_epa_loop<" in name):
        hit = True
    if base == "k_closed" and "k_closed_staged(" in name:
        hit = True
    solid_walk = re.search(r"k_bvh_collide<\w+, false, false, true>", name) is not None
    if base == "k_bvh_collide":
        hit = (hit and not solid_walk) or any(x in name for x in ("k_bvh_coop<", "k_bvh_walk<", "k_tri_leaves<", "k_bvh_resolve<", "k_bvh_combine<"))
    if base == "k_bvh_shape":
        hit = hit or solid_walk or any(x in name for x in ("k_shape_obb", "k_bvh_shape_coop<", "k_bvh_shape_finish<"))
    if base == "k_bvh_distance":
        hit = hit or any(x in name for x in ("k_bvh_distance_pool<", "k_bvh_distance_coop<"))
    return hit


def rocprof_names(workload, label):
    """The kernels rocprofv3 lists for a timer label of the library (the label of a pass of several kernels: all of them), longest-running
    first as far as the committed kernel trace of the workload tells (profiles/r06_z_<workload>_kernel_trace_stats.txt, else the PMC
    pass).  Names only -- they do not depend on the device code's fingerprint; [] when no committed file knows the label."""
    import glob
    import re
    names = []
    for path in sorted(glob.glob(os.path.join(ROOT, "profiles", "r0*_z_%s_kernel_trace_stats.txt" % workload)), reverse=True):
        for ln in open(path):
            m = re.match(r"((?:void )?k_\w+.*?\S)\s+\d+\s+[\d.]+\s+[\d.]+\s+[\d.]+\s*$", ln)
            if m and _label_matches(label, m.group(1)) and m.group(1) not in names:
                names.append(m.group(1))  # (the file is sorted by total time)
        if names:
            return names
    path = os.path.join(ROOT, "profiles", "traffic_%s.json" % workload)
    if os.path.exists(path):
        names = [k for k in json.load(open(path)).get("kernels", {}) if _label_matches(label, k)]
    return names


def load_step_traffic(workload, n):
    """HBM bytes of ONE STEP -- every kernel of the pipeline, 2 x FETCH_SIZE + WRITE_SIZE as in load_traffic -- from the committed PMC pass
    of this device code and batch size; None otherwise."""
    path = os.path.join(ROOT, "profiles", "traffic_%s.json" % workload)
    if not os.path.exists(path):
        return None
    t = json.load(open(path))
    if (t.get("pairs") or DEFAULT_PAIRS.get(workload, 1_000_000)) != n or t.get("source_sha") != kernel_source_sha():
        return None
    tot, seen = 0.0, False
    for v in t["kernels"].values():
        if "FETCH_SIZE_KB_per_dispatch" in v and "WRITE_SIZE_KB_per_dispatch" in v:
            tot += 2 * v["FETCH_SIZE_KB_per_dispatch"] * 1024 + v["WRITE_SIZE_KB_per_dispatch"] * 1024
            seen = True
    return tot if seen else None


# "useful" floating-point work of a hull x hull query, for `roofline.useful_flop_frac`: a MODEL of what the reference's algorithm needs
# per iteration (SURVEY.md 8d quotes 5-10 kflop per query for configs[2]) times the iteration counts the records of this run carry.
#   GJK iteration (gjk.cpp:188-370): two support scans of 32 vertices (a dot product = 5 flop + the compare) = 2 x 32 x 6, the direction
#     into both frames and the support point back (2 x 15 + 2 x 18), the simplex projection (line 20 / triangle 90 / tetrahedron 250:
#     150 on average over a run), the convergence check and momentum update (40)              -> ~640 flop
#   EPA iteration (gjk.cpp:1311-1466): the same two scans and transforms (450), the silhouette over ~40 faces (a dot + compare each:
#     6 x 40), ~4 new faces (normal = cross 9 + normalise 8 + distance 5 + the degenerate test 6 each), the closest-face scan (2 x 40)
#     -> ~880 flop;  set-up of a penetrating pair (encloseOrigin of the GJK simplex + the first tetrahedron's four faces) ~300 flop
# The lanes execute 4-8 times as many instructions (both lanes of a pair run the serial simplex code, eight lanes share a
# polytope): this figure prices that redundancy, `valu_frac` prices issue efficiency.
FLOP_GJK_ITER, FLOP_EPA_ITER, FLOP_EPA_SETUP = 640.0, 880.0, 300.0
FP32_VECTOR_PEAK_TFLOPS = 157.3  # MI355X, /opt/skills/guides/MI355X_MICROARCH.md (non-matrix fp32)


class Ctx:
    pass


def _r(x, sig=5):
    """Numbers of the compact line carry `sig` significant digits: the line has a byte budget."""
    if isinstance(x, bool) or x is None or isinstance(x, (int, str)):
        return x
    if isinstance(x, float):
        if x != x or x in (float("inf"), float("-inf")):
            return None
        return float("%.*g" % (sig, x))
    return x


def _pick(d, keys):
    return {k: _r(d[k]) for k in keys if d is not None and k in d}


COMPACT_LIMIT = 4096  # bytes; the driver keeps the last 8 KB of stdout and parses its last line


def compact_line(full):
    """The LAST stdout line of bench.py: the contract's fields, `roofline` and `cpu_baseline` of the headline, and the
    secondary workloads as 6-field rows -- at most COMPACT_LIMIT bytes whatever the full record grows to.  The full
    record (per-kernel durations, SQ counters, notes, bucket counts, the exchange check ...) goes to bench_full.json; nothing in the compact line is computed here, every number is taken from the full record."""
    line = _pick(full, ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling",
                        "vs_baseline", "dtype", "data"))
    cfg = full.get("config") or {}
    line["config"] = _pick(cfg, ("workload", "baseline_config", "pairs_per_gpu_per_step", "pairs_per_step_all_gpus",
                                 "contact_fraction", "request", "gather", "backend", "mean_bv_tests", "mean_leaf_tests"))
    if cfg.get("exchange"):
        line["config"]["exchange"] = _pick(cfg["exchange"], ("ranks_seen", "ms", "bus_GBps", "frac_of_link_budget", "bytes_received_per_rank"))
    if cfg.get("per_rank_ms_no_exchange"):
        line["config"]["per_rank_ms_no_exchange"] = _pick(cfg["per_rank_ms_no_exchange"], ("max", "min"))
    gc = cfg.get("gather_check")
    if gc:
        line["config"]["gather_check"] = {k: (all(v) if isinstance(v, (list, tuple)) else v) for k, v in gc.items()}
    rf = full.get("roofline") or {}
    line["roofline"] = _pick(rf, ("bound", "kernel", "timer_label", "achieved", "peak", "unit", "frac", "traffic", "traffic_ratio", "bytes_per_query",
                                  "units_per_launch", "kernel_ms", "pipeline_ms", "pipeline_achieved", "step_traffic", "step_traffic_ratio",
                                  "useful_flop_per_query", "useful_flop_frac", "hbm_side", "l2_side"))
    if rf.get("traffic_source"):
        line["roofline"]["traffic_source"] = rf["traffic_source"]
    if rf.get("valu_issue"):
        line["roofline"]["valu_frac"] = _r(rf["valu_issue"]["frac"])
    cb = full.get("cpu_baseline")
    if cb:
        line["cpu_baseline"] = _pick(cb, ("value", "unit", "cores", "kind", "sample"))
        if cb.get("all_cores"):
            line["cpu_baseline"]["all_cores"] = _pick(cb["all_cores"], ("value", "cores"))
        if cb.get("native"):
            line["cpu_baseline"]["native"] = _pick(cb["native"], ("value", "cores", "build"))
    else:
        line["cpu_baseline"] = None
    hb = full.get("host_buffers")
    if hb:
        line["host_buffers"] = _pick(hb, ("pairs", "value", "ms_per_call", "frac_of_link_bound"))
    rows = []
    for s in full.get("secondary") or []:
        if "error" in s:
            rows.append({"workload": s.get("workload"), "error": str(s["error"])[:80]})
            continue
        srf, scb = s.get("roofline") or {}, s.get("cpu_baseline") or {}
        # two fractions of the HBM peak per row, each with ONE meaning on every row: l2_frac = what the lanes consume (the algorithmic
        # bytes of the row: visited node / triangle records for the mesh rows, which live in L2 / MALL) / kernel time; hbm_frac = what the
        # memory controllers moved (PMC bytes of the committed pass of this device code) / kernel time, null without such a pass
        kms = srf.get("kernel_ms")
        hbm_frac = (srf["traffic"] / (kms * 1e-3) / 1e9 / HBM_PEAK_GBS) if (srf.get("traffic") and kms and kms == kms and kms > 0) else None
        row = {"workload": s.get("tag") or str(s.get("workload"))[:72], "value": _r(s.get("value"), 4), "ms_per_step": _r(s.get("ms_per_step"), 4),
               "dtype": s.get("dtype"), "l2_frac": _r(srf.get("frac"), 3), "hbm_frac": _r(hbm_frac, 3), "cpu_1thread": _r(scb.get("value"), 4)}
        if (s.get("config") or {}).get("pairs_per_step_all_gpus"):
            row["pairs"] = s["config"]["pairs_per_step_all_gpus"]
        shb = s.get("host_buffers")
        if shb:
            row["host_buffers_qps"] = _r(shb.get("value"), 4)
            row["host_link_frac"] = _r(shb.get("frac_of_link_bound"), 3)
        if s.get("batches_in_flight", 1) > 1:
            row["batches_in_flight"] = s["batches_in_flight"]
        rows.append(row)
    if rows:
        line["secondary"] = rows
    line["full_record"] = "bench_full.json"
    out = json.dumps(line, separators=(",", ":"))
    # the budget holds by construction for today's eleven secondary rows (~2.6 KB); if rows are added until it does not,
    # the rows go first, never the headline
    while len(out) > COMPACT_LIMIT and line.get("secondary"):
        line["secondary"] = line["secondary"][:-1]
        line["secondary_truncated"] = True
        out = json.dumps(line, separators=(",", ":"))
    for optional in ("host_buffers", "cpu_baseline", "roofline"):  # (cannot happen with today's fields; a line is printed whatever happens)
        if len(out) <= COMPACT_LIMIT:
            break
        line[optional] = None
        line["truncated"] = True
        out = json.dumps(line, separators=(",", ":"))
    return out


def make_batch(ctx, workload, n, strong):
    """The synthetic batch of this rank: (batch, dtype, extra config).  weak: own seed per rank; strong: one list
    (same seed everywhere), rank r keeps sharding.shard_range(n, r, world)."""
    wl, pkg = ctx.pkg.workloads, ctx.pkg
    seed = 1 if strong else 1 + ctx.rank
    key = (workload, n, seed, strong)
    cache = getattr(ctx, "batch_cache", None)
    if cache is not None and cache[0] == key:  # (the strong list is run twice at N > 1: full and compact exchange)
        b_local, b_dtype, b_extra, b_full = cache[1]
        return b_local, b_dtype, dict(b_extra), b_full
    extra = {}
    if workload == "cfg3":
        batch, dtype = wl.cfg3_convex_convex(n=n, seed=seed), "f32"
    elif workload == "cfg3u":
        batch, dtype = wl.cfg3_unique_hulls(n=n, seed=seed), "f32"
    elif workload == "cfg1":
        batch, dtype = wl.cfg1_sphere_sphere(n=n, seed=seed), "f64"
    elif workload == "cfg2":
        batch, dtype = wl.cfg2_box_capsule(n=n, seed=seed), "f64"
    elif workload == "cfg2f":
        batch, dtype = wl.cfg2_box_capsule(n=n, seed=seed), "f32"
    elif workload == "cfg5":
        # pair list = host broadphase over a scene of n/10 posed objects (not in the timed region); the list is cut to
        # exactly n pairs.  Several ranks on one host share its cores.
        over = 1.15
        threads = max(1, (os.cpu_count() or 1) // max(1, ctx.world))
        while True:
            t_bp = time.perf_counter()
            batch = wl.cfg5_broadphase_scene(n_objects=max(n // 10, 100), target_pairs=int(over * n), seed=seed, n_threads=threads)
            t_bp = time.perf_counter() - t_bp
            if len(batch) >= n:
                break
            over *= 1.3
        extra = {"objects": batch.scene["n_objects"], "broadphase_pairs": len(batch), "host_broadphase_s": t_bp}
        batch = batch.slice(0, n)
        dtype = "f64"
    elif workload == "cfg4d":
        batch, dtype = wl.cfg4_mesh_mesh_distance(n=n, seed=seed), "f64"
    elif workload == "cfg4s":
        batch, dtype = wl.mesh_vs_solid("mixed", n=n, seed=seed), "f64"
    elif workload == "cfgmix":
        batch, dtype = wl.mixed_scene(n=n, seed=seed), "f64"
        extra = {"mix": batch.mix}
    else:
        batch, dtype = wl.cfg4_mesh_mesh(n=n, seed=seed), "f64"
    if strong:
        lo, hi = pkg.sharding.shard_range(n, ctx.rank, ctx.world)
        extra["shard"] = [lo, hi]
        batch_local = batch.slice(lo, hi)
        batch_local.meshes = getattr(batch, "meshes", None)
        ctx.batch_cache = (key, (batch_local, dtype, dict(extra), batch))
        return batch_local, dtype, extra, batch
    return batch, dtype, extra, batch


def cpu_baseline(ctx, workload, batch, req, sample, budget_s, all_cores=True, native=False):
    """The fp64 CPU oracle (oracle/, a restatement of the reference's algorithm: kind "port") on a bounded sample of
    the same workload: 1 thread (the reference is single-threaded) and, for scale, all host threads."""
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import oracle_binding as ob  # checker/baseline only -- never on the product path
    ns = min(sample, len(batch))
    sb = batch.slice(0, ns)
    tf1, tf2 = sb.tf1, sb.tf2
    if workload == "cfg4":
        MLc = ctx.pkg.bvh_builder.MeshLibrary(batch.meshes)

        def run_cpu(lo, hi, threads):
            ob.bvh_collide_batch(MLc, sb.s1[lo:hi], sb.s2[lo:hi], tf1[lo:hi], tf2[lo:hi], req, n_threads=threads)
    elif workload == "cfg4d":
        MLc = ctx.pkg.bvh_builder.MeshLibrary(batch.meshes)

        def run_cpu(lo, hi, threads):
            ob.bvh_distance_batch(MLc, sb.s1[lo:hi], sb.s2[lo:hi], tf1[lo:hi], tf2[lo:hi], n_threads=threads)
    elif workload in ("cfg4s", "cfgmix"):
        MLc = ctx.pkg.bvh_builder.MeshLibrary(batch.meshes)

        def run_cpu(lo, hi, threads):
            ob.mixed_collide_batch(sb.shapes, sb.verts, MLc, sb.s1[lo:hi], sb.s2[lo:hi], tf1[lo:hi], tf2[lo:hi], req, n_threads=threads)
    else:
        fn = ob.distance_batch if sb.kind == "distance" else ob.collide_batch

        def run_cpu(lo, hi, threads):
            fn(sb.shapes, sb.verts, sb.s1[lo:hi], sb.s2[lo:hi], tf1[lo:hi], tf2[lo:hi], req, n_threads=threads)
    run_cpu(0, min(1000, ns), 1)  # warm-up
    reps, t_cpu = 0, 0.0
    while t_cpu < budget_s and reps < 5:
        t1 = time.perf_counter()
        run_cpu(0, ns, 1)
        t_cpu += time.perf_counter() - t1
        reps += 1
    out = {"value": reps * ns / t_cpu, "unit": "queries/s", "cores": 1, "kind": "port",
           "sample": "%d pairs of the same workload x %d repeats, fp64 CPU oracle (oracle/: g++ -O3 -march=x86-64-v2, no FMA -- the reference's default "
                     "arithmetic, not -march=native), 1 thread" % (ns, reps)}
    if all_cores:
        cores_all = os.cpu_count() or 1
        t1 = time.perf_counter()
        run_cpu(0, ns, cores_all)
        out["all_cores"] = {"value": ns / (time.perf_counter() - t1), "cores": cores_all}
    if native:
        # BASELINE.md section 3 builds the CPU side with -march=native; the figure above is the reference's DEFAULT arithmetic (no FMA),
        # which is what the parity tests hold the device to.  Both are stated: a second build of the same sources, made on this host.
        try:
            with ob.native_build():
                run_cpu(0, min(1000, ns), 1)
                t1 = time.perf_counter()
                run_cpu(0, ns, 1)
                out["native"] = {"value": ns / (time.perf_counter() - t1), "cores": 1, "build": "g++ -O3 -march=native (FMA contraction on)"}
        except Exception as e:  # (no compiler on the box: the figure is absent, the line is not)
            out["native"] = {"value": None, "cores": 1, "build": "failed: " + repr(e)[:60]}
    return out


def host_boundary(pkg, lib, batch, req, device_records, f32=False):
    """hfcl_collide_batch / hfcl_distance_batch (f32: hfcl_*_batch_f32, 7-float poses and 44-byte records) on pageable host arrays:
    queries/s, and the time against what the link admits (inputs in, records out, both directions at once)."""
    import ctypes as C
    abi = pkg.abi
    n = len(batch)
    dll = pkg.engine.dll()
    s1, s2 = batch.s1.astype(np.uint32), batch.s2.astype(np.uint32)
    if f32:
        cfn = dll.hfcl_distance_batch_f32 if batch.kind == "distance" else dll.hfcl_collide_batch_f32
        tf1, tf2 = np.ascontiguousarray(batch.pose1_f32), np.ascontiguousarray(batch.pose2_f32)
        out = np.zeros(n, dtype=abi.RESULT_F32_DTYPE)
    else:
        cfn = dll.hfcl_distance_batch if batch.kind == "distance" else dll.hfcl_collide_batch
        tf1, tf2 = np.ascontiguousarray(batch.tf1), np.ascontiguousarray(batch.tf2)
        out = np.zeros(n, dtype=abi.RESULT_DTYPE)
    out[:] = out  # touched: first-touch page faults of a fresh array are the allocator's, not the link's
    ts = []
    for _ in range(5):
        t1 = time.perf_counter()
        if f32:
            rc = cfn(lib._h, abi.ptr(s1), abi.ptr(s2), abi.ptr(tf1), abi.ptr(tf2), C.c_size_t(n), C.byref(req), abi.ptr(out))
        else:
            rc = cfn(lib._h, abi.ptr(s1), abi.ptr(s2), abi.ptr(tf1), abi.ptr(tf2), C.c_size_t(n), C.byref(req), abi.ptr(out), None, None)
        ts.append(time.perf_counter() - t1)
        assert rc == 0, pkg.engine.last_error()
    t_h = min(ts[1:])
    b_in, b_out = (8 + 56, 44) if f32 else (8 + 192, 96)
    bound_s = max(n * b_in / (LINK_H2D_GBS * 1e9), n * b_out / (LINK_D2H_GBS * 1e9), n * (b_in + b_out) / (LINK_BOTH_GBS * 1e9))
    r = {"pairs": n, "value": n / t_h, "unit": "queries/s", "ms_per_call": 1e3 * t_h, "ms_first_call": 1e3 * ts[0],
         "bytes_in_per_pair": b_in, "bytes_out_per_pair": b_out,
         "link_GBps_measured": {"h2d": LINK_H2D_GBS, "d2h": LINK_D2H_GBS, "both_directions_total": LINK_BOTH_GBS},
         "link_bound_ms": 1e3 * bound_s, "frac_of_link_bound": bound_s / t_h,
         "note": "hfcl_%s_batch%s on pageable host arrays: chunked H2D | kernels | D2H pipeline; PCIe inclusive; best of 4 calls on "
                 "the same arrays (the first call on fresh arrays also pays the pinning of their pages: ms_first_call)" % (batch.kind, "_f32" if f32 else "")}
    if device_records is not None:
        r["records_identical_to_device_path"] = bool(np.array_equal(out.view(np.int32), device_records.view(np.int32)))
    return r


def run_workload(ctx, workload, n_arg, steps, warmup, strong=False, cpu_budget_s=10.0, cpu_sample=1_000_000, host_buffers=False,
                 gather=None, two_streams=False):
    """One timed region: `steps` passes of the hot path over this rank's batch.  Returns the result dict on rank 0."""
    import torch
    import torch.distributed as dist
    args, pkg, dev = ctx.args, ctx.pkg, ctx.dev
    abi, wl = pkg.abi, pkg.workloads
    n_total = n_arg or DEFAULT_PAIRS.get(workload, 1_000_000)
    batch, dtype, extra_cfg, full = make_batch(ctx, workload, n_total, strong)
    n = len(batch)  # pairs of this rank per step
    req = wl.make_request(batch, abi)
    lib = wl.make_library(pkg, full, device=ctx.device_index)
    if args.split:
        lib.set_split(args.split)
    # two_streams (a secondary line only, never the headline): even steps on the caller's stream, odd steps on a second
    # stream through a second library object (own workspace), as a throughput-oriented caller with independent batches
    # would run them -- the GJK kernels of one batch fill the drain of the other's EPA kernels
    lib2 = wl.make_library(pkg, full, device=ctx.device_index) if two_streams else None
    stream2 = torch.cuda.Stream(device=dev) if two_streams else None

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
    # strong scaling: shards may be ragged; the collective moves equal-sized (padded) blocks
    per = pkg.sharding.padded_shard_len(n_total, ctx.world) if strong else n
    outs = [torch.zeros(per * rec_words, dtype=torch.int32, device=dev) for _ in range(2)]
    gather_mode = ("none" if args.no_gather else (gather or args.gather)) if ctx.dist_on else "none"
    xch = pkg.multigpu.RecordExchange(lib, dev, per, dtype, gather_mode if ctx.dist_on else "none",
                                      dist=dist if ctx.dist_on else None, staged=ctx.backend == "gloo")
    gather = xch.dist is not None
    stream = torch.cuda.current_stream()
    kernel_ms = {}
    sent = {}

    def one_step(i, record_times, exchange=True):
        buf = i & 1
        if exchange:
            # double buffering: the kernels of step i overwrite the buffer the exchange of step i-2 read
            xch.before_launch(buf, stream)
        if n and two_streams and buf and not record_times:
            launch2(d_s1, d_s2, d_p1, d_p2, n, req, outs[buf], stream=stream2.cuda_stream)
        elif n:
            launch(d_s1, d_s2, d_p1, d_p2, n, req, outs[buf], stream=stream.cuda_stream)
        if exchange:
            # results of this step travel over xGMI while the next step's kernels run
            sent[buf] = xch.after_launch(buf, outs[buf], n, stream)
        if record_times and n:
            for name, ms in lib.last_kernel_breakdown():  # HIP events on the launch stream
                kernel_ms.setdefault(name, []).append(ms)

    def sync_all():
        torch.cuda.synchronize()
        if ctx.dist_on:
            dist.barrier()
        torch.cuda.synchronize()

    lib.set_kernel_timing(False)  # the timed region runs without the per-kernel event markers
    if two_streams:
        assert not gather, "the two-stream form is a single-GPU secondary line"
        lib2.set_kernel_timing(False)
        kind_fn = ("distance" if batch.kind == "distance" else "collide") + "_device" + ("_f32" if dtype == "f32" else "")
        launch2 = getattr(lib2, kind_fn)
    for i in range(warmup):
        one_step(i, False)
        xch.drain()
    sync_all()
    t0 = time.perf_counter()
    for i in range(steps):
        one_step(i, False)
    xch.drain()
    sync_all()
    elapsed = time.perf_counter() - t0
    if ctx.dist_on:
        t = torch.tensor([elapsed], dtype=torch.float64, device=dev if ctx.backend != "gloo" else "cpu")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())

    # what did the exchange deliver?  (outside the timed region)
    last = (steps - 1) & 1
    gather_check = None
    if gather and steps > 0:
        gather_check = {"block_checksums": xch.verify(last, sent[last], ctx.rank)}
        verify = args.verify_gather if args.verify_gather is not None else ctx.world > 1  # (on by default for every real multi-rank run)
        if verify and not strong and workload in WALK_BYTES:
            verify = False  # (weak scaling regenerates every rank's batch on rank 0: done for the pair-list workloads, not for the mesh scenes)
        if verify and not strong:
            # weak scaling: every rank ran its own list (seed 1 + rank).  Rank 0 regenerates each of them, runs it through the same entry
            # point alone, and compares with the rank's block of the gathered buffer: bytes equal to single-rank runs
            ok = None
            if ctx.rank == 0:
                ok = True
                per_words = xch.gathered[last].numel() // ctx.world
                for r in range(ctx.world):
                    c2 = Ctx()
                    c2.__dict__.update(ctx.__dict__)
                    c2.rank, c2.batch_cache = r, None
                    b_r = make_batch(c2, workload, n, False)[0]
                    lib_r = wl.make_library(pkg, b_r, device=ctx.device_index)  # (the rank's own shape library: its seed makes the hulls too)
                    name = ("distance" if b_r.kind == "distance" else "collide") + ("_device_f32" if dtype == "f32" else "_device")
                    launch_r = getattr(lib_r, name)
                    f_s1 = torch.from_numpy(b_r.s1.astype(np.int32)).to(dev)
                    f_s2 = torch.from_numpy(b_r.s2.astype(np.int32)).to(dev)
                    f_p1 = torch.from_numpy(b_r.pose1_f32 if dtype == "f32" else b_r.tf1).to(dev)
                    f_p2 = torch.from_numpy(b_r.pose2_f32 if dtype == "f32" else b_r.tf2).to(dev)
                    f_out = torch.zeros(n * rec_words, dtype=torch.int32, device=dev)
                    launch_r(f_s1, f_s2, f_p1, f_p2, n, req, f_out, stream=stream.cuda_stream)
                    torch.cuda.synchronize()
                    rec = f_out.cpu().numpy().view(abi.RESULT_F32_DTYPE if dtype == "f32" else abi.RESULT_DTYPE)
                    want = pkg.multigpu.expected_exchange(rec, dtype, gather_mode)
                    got = xch.gathered[last][r * per_words:r * per_words + want.size].cpu().numpy()
                    ok = ok and bool(np.array_equal(got, want))
                    lib_r.close()
                    del f_s1, f_s2, f_p1, f_p2, f_out
            gather_check["equals_single_rank_run"] = ok
        if verify and strong:
            # the whole list on ONE rank, through the same entry point: the gathered buffer must hold exactly these bytes
            ok = None
            if ctx.rank == 0:
                f_s1 = torch.from_numpy(full.s1.astype(np.int32)).to(dev)
                f_s2 = torch.from_numpy(full.s2.astype(np.int32)).to(dev)
                f_p1 = torch.from_numpy(full.pose1_f32 if dtype == "f32" else full.tf1).to(dev)
                f_p2 = torch.from_numpy(full.pose2_f32 if dtype == "f32" else full.tf2).to(dev)
                f_out = torch.zeros(n_total * rec_words, dtype=torch.int32, device=dev)
                launch(f_s1, f_s2, f_p1, f_p2, n_total, req, f_out, stream=stream.cuda_stream)
                torch.cuda.synchronize()
                rec = f_out.cpu().numpy().view(abi.RESULT_F32_DTYPE if dtype == "f32" else abi.RESULT_DTYPE)
                want = pkg.multigpu.expected_exchange(rec, dtype, gather_mode)
                got = xch.gathered[last][:want.size].cpu().numpy()
                ok = bool(np.array_equal(got, want))
                del f_s1, f_s2, f_p1, f_p2, f_out
            gather_check["equals_single_rank_run"] = ok

    # N > 1: what the run can say about itself, outside the timed region -- the ranks the backend reached, the exchange alone, and every
    # rank's steps WITHOUT the exchange (max / min over ranks: weak scaling's per-rank figure is the N = 1 bench's ms_per_step, so the
    # driver can hold a SCALE line against its BENCH line)
    exchange_probe = xch.probe(sent[last]) if (gather and steps > 0) else None
    rank_ms = None
    if ctx.dist_on and steps > 0:
        sync_all()
        t1 = time.perf_counter()
        for i in range(steps):
            one_step(i, False, exchange=False)
        torch.cuda.synchronize()
        mine = 1e3 * (time.perf_counter() - t1) / steps
        t = torch.tensor([mine, -mine], dtype=torch.float64, device=dev if ctx.backend != "gloo" else "cpu")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        rank_ms = {"max": float(t[0].item()), "min": -float(t[1].item()), "rank0": mine,
                   "note": "ms per step of a rank's own kernels with no exchange in flight (weak scaling: the N = 1 bench's ms_per_step)"}
        sync_all()

    # per-kernel durations (HIP events inside the library, on the launch stream), separate pass so
    # the event reads do not serialise the timed region
    lib.set_kernel_timing(True)
    for i in range(min(steps, 10)):
        one_step(i, True, exchange=False)
    torch.cuda.synchronize()

    result = None
    if ctx.rank == 0:
        res = outs[(steps - 1) & 1][:n * rec_words].cpu().numpy()
        status = res.view(abi.RESULT_F32_DTYPE if dtype == "f32" else abi.RESULT_DTYPE)["status"]
        contact_frac = float(abi.status_contact(status).mean()) if n else 0.0
        buckets = lib.last_bucket_counts()
        total_q = steps * (n_total if strong else n * ctx.world)
        qps = total_q / elapsed
        ms_per_step = 1e3 * elapsed / steps
        avg = {k: float(np.mean(v)) for k, v in kernel_ms.items() if np.mean(v) > 0}
        dominant = max(avg, key=avg.get) if avg else ""
        bpq = BYTES_PER_QUERY[workload]
        if workload in WALK_BYTES:
            sys.path.insert(0, os.path.join(ROOT, "tests"))
            import oracle_binding as ob0  # measures N_bv / N_leaf of the reference DFS on a sample (outside the timed region)
            ns0 = min(2000, n)
            ML = pkg.bvh_builder.MeshLibrary(batch.meshes)
            sl = slice(0, ns0)
            if workload == "cfg4":
                _, st0 = ob0.bvh_collide_batch(ML, batch.s1[sl], batch.s2[sl], batch.tf1[sl], batch.tf2[sl], req,
                                               n_threads=os.cpu_count() or 1, want_stats=True)
            elif workload == "cfg4d":
                _, st0 = ob0.bvh_distance_batch(ML, batch.s1[sl], batch.s2[sl], batch.tf1[sl], batch.tf2[sl],
                                                n_threads=os.cpu_count() or 1, want_stats=True)
            else:
                _, st0 = ob0.mixed_collide_batch(batch.shapes, batch.verts, ML, batch.s1[sl], batch.s2[sl], batch.tf1[sl], batch.tf2[sl], req,
                                                 n_threads=os.cpu_count() or 1, want_stats=True)
            nbv, nleaf = float(st0[:, 0].mean()), float(st0[:, 1].mean())
            bpq = bpq + WALK_BYTES[workload][0] * nbv + WALK_BYTES[workload][1] * nleaf
            extra_cfg.update({"mean_bv_tests": nbv, "mean_leaf_tests": nleaf, "stats_sample": ns0})
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
        traffic, traffic_note = load_traffic(workload, dominant, n)
        valu_insts, sq = load_traffic(workload, dominant, n, want="valu")
        valu = None
        if valu_insts and dom_ms == dom_ms and dom_ms > 0:
            # the bound that actually applies to the iterative kernels: wave-level VALU instructions issued per
            # launch (SQ_INSTS_VALU, committed PMC pass of this code) against the MEASURED issue peak over the live kernel duration
            valu = {"insts_per_launch": valu_insts, "issue_peak_per_s": VALU_ISSUE_PEAK,
                    "frac": valu_insts / (dom_ms * 1e-3) / VALU_ISSUE_PEAK, "sq_counters_per_launch": sq,
                    "note": "wave-level VALU instructions / (kernel time x measured chip-wide issue peak, tools/valu_peak.hip); "
                            "fp64 arithmetic issues at 0.56 of that rate"}
        names = rocprof_names(workload, dominant)
        step_traffic = load_step_traffic(workload, n)
        useful = None
        if workload in ("cfg3", "cfg3u") and n:
            g_it = float(abi.status_gjk_iters(status).mean())
            e_it = float((abi.status_epa_iters(status) * (abi.status_epa(status) != 15)).mean())
            ran_epa = float((abi.status_epa(status) != 15).mean())
            useful = g_it * FLOP_GJK_ITER + e_it * FLOP_EPA_ITER + ran_epa * FLOP_EPA_SETUP
            extra_cfg.update({"mean_gjk_iterations": g_it, "mean_epa_iterations_over_all_pairs": e_it})
        roofline = {
            "bound": "hbm", "kernel": (names[0].split("(")[0].replace("void ", "").strip() if names else dominant), "timer_label": dominant,
            "kernels_under_label": names,
            "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
            "frac": (achieved / HBM_PEAK_GBS) if achieved else None, "traffic": traffic, "traffic_note": traffic_note,
            "traffic_ratio": (traffic / (units * bpq)) if (traffic and units) else None,
            "traffic_source": ("profiles/traffic_%s.json (PMC pass of this device code, source_sha %s; replayed, not collected in this run)" % (
                workload, kernel_source_sha())) if traffic else None,
            "bytes_per_query": bpq, "units_per_launch": units, "kernel_ms": dom_ms,
            "pipeline_ms": pipeline_ms, "pipeline_achieved": (n * bpq) / (ms_per_step * 1e-3) / 1e9,
            "step_traffic": step_traffic, "step_traffic_ratio": (step_traffic / (n * bpq)) if (step_traffic and n) else None,
            "useful_flop_per_query": useful,
            "useful_flop_frac": (useful * n / (ms_per_step * 1e-3) / (FP32_VECTOR_PEAK_TFLOPS * 1e12)) if useful else None,
            "kernels_ms": avg, "valu_issue": valu,
        }
        if workload in WALK_BYTES and achieved:
            # The walks read node records that live in L2 / MALL (8 models x 1.28 MB here): the "algorithmic" rate above is what the
            # LANES consume (an L2-side figure), the HBM-side figure is what the PMC pass saw leave the memory controllers.  Both
            # against the HBM peak, both stated -- the single `frac` of earlier rounds mixed them.
            roofline["l2_side"] = {"GBps": _r(achieved, 4), "frac_of_hbm_peak": _r(achieved / HBM_PEAK_GBS, 3),
                                   "note": "visited node / triangle records x record size / kernel time"}
            roofline["hbm_side"] = ({"GBps": _r(traffic / (dom_ms * 1e-3) / 1e9, 4), "frac_of_hbm_peak": _r(traffic / (dom_ms * 1e-3) / 1e9 / HBM_PEAK_GBS, 3),
                                     "note": "PMC bytes (2 x FETCH_SIZE + WRITE_SIZE) / kernel time"} if traffic else None)
        cpu = None
        if not args.no_cpu_baseline and ctx.world == 1 and cpu_budget_s > 0:  # reported on rank 0 at N=1 only
            cpu = cpu_baseline(ctx, workload, batch, req, cpu_sample, cpu_budget_s, native=cpu_budget_s >= 5.0)
        if two_streams:
            # a throughput-of-two-batches figure: the per-kernel durations above come from the one-stream event pass and do
            # not describe kernels that overlap another batch's, so this row carries no kernel / roofline figure of its own
            roofline = {"bound": "hbm", "kernel": None, "achieved": None, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": None,
                        "traffic": None, "bytes_per_query": bpq, "pipeline_achieved": roofline["pipeline_achieved"],
                        "note": "two batches in flight; see the one-stream row of the same workload for the kernels"}
        result = {
            "workload": batch.name + (" (two batches in flight: even / odd steps on two streams)" if two_streams else ""),
            "tag": workload + ("/strong" if strong else "") + ("/2streams" if two_streams else "") + (
                "/%dk" % (n // 1000) if (n_arg and not strong and n != DEFAULT_PAIRS.get(workload, 1_000_000)) else "") + (
                "/gather=" + gather_mode if ctx.dist_on else ""),
            "batches_in_flight": 2 if two_streams else 1,
            "value": qps, "unit": "queries/s", "ms_per_step": ms_per_step, "steps": steps, "warmup": warmup,
            "dtype": dtype, "scaling": "strong" if strong else "weak",
            "config": {"workload": batch.name, **extra_cfg, "baseline_config": BASELINE_CONFIG[workload],
                       "pairs_per_gpu_per_step": n, "pairs_per_step_all_gpus": n_total if strong else n * ctx.world,
                       "contact_fraction": contact_frac, "buckets": buckets,
                       "request": batch.kind, "all_gather_results": bool(gather), "gather": gather_mode,
                       "gather_bytes_per_rank_per_step": dict(zip(("sent", "received"), xch.bytes_per_rank_per_step())),
                       "gather_check": gather_check, "backend": ctx.backend if ctx.dist_on else None,
                       "exchange": ({**exchange_probe, "frac_of_link_budget": (exchange_probe["bus_GBps"] / XGMI_IN_GBPS) if exchange_probe.get("bus_GBps") else None,
                                     "link_budget_GBps": XGMI_IN_GBPS} if exchange_probe else None),
                       "per_rank_ms_no_exchange": rank_ms,
                       "split_parts": lib.last_split_parts(),
                       "lane_group_width": os.environ.get("HFCL_CVX_W", "auto (2; fp64 convex-convex 4)")},
            "roofline": roofline, "cpu_baseline": cpu,
        }
        if host_buffers and ctx.world == 1:
            # the same batch through the host-buffer boundary (what a hpp::fcl::collide()/distance() caller gets):
            # PCIe inclusive, never the headline value
            result["host_buffers"] = host_boundary(pkg, lib, batch, req, res, f32=(dtype == "f32"))
            if workload == "cfg2":  # ... and at a quarter / four times the size
                result["host_buffers"]["other_sizes"] = [
                    host_boundary(pkg, lib, wl.cfg2_box_capsule(n=m, seed=1 + ctx.rank), req, None) for m in (250_000, 4_000_000)]
    lib.close()
    if lib2 is not None:
        lib2.close()
    del d_s1, d_s2, d_p1, d_p2, outs, xch
    torch.cuda.empty_cache()
    return result


def main():
    args = parse()
    if "WORLD_SIZE" not in os.environ and args.gpus > 1:
        # started as a plain process: become the launcher of the N ranks (one process per GPU) and pass their output through
        pkg = load_pkg()
        if not args.device_map:
            import torch
            have = torch.cuda.device_count() if torch.cuda.is_available() else 0
            if have < args.gpus:
                raise SystemExit("bench.py --gpus %d: only %d GPU(s) visible on this node (several ranks can share a device "
                                 "with --backend gloo --device-map 0,0,...)" % (args.gpus, have))
        raise SystemExit(pkg.multigpu.spawn_ranks(os.path.abspath(__file__), sys.argv[1:], args.gpus))
    import torch
    import torch.distributed as dist

    ctx = Ctx()
    ctx.args = args
    ctx.rank = int(os.environ.get("RANK", "0"))
    ctx.local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    ctx.world = int(os.environ.get("WORLD_SIZE", "1"))
    ctx.backend = args.backend
    if args.gpus != ctx.world:
        raise SystemExit("bench.py: --gpus %d but the launcher started %d rank(s)" % (args.gpus, ctx.world))
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU (the engine has no CPU fallback)")
    ctx.pkg = load_pkg()
    ctx.device_index = ctx.pkg.multigpu.device_of_rank(ctx.local_rank, args.device_map)
    torch.cuda.set_device(ctx.device_index)
    ctx.dev = torch.device("cuda", ctx.device_index)
    # HFCL_BENCH_FORCE_DIST=1: run the N>1 code path (process group, comm stream, async all-gather, barriers, max over
    # ranks) with a single rank -- a dry run of the multi-GPU plumbing on a 1-GPU box
    ctx.dist_on = ctx.world > 1 or os.environ.get("HFCL_BENCH_FORCE_DIST") == "1"
    if ctx.dist_on:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29533")
        import datetime
        limit = datetime.timedelta(minutes=15)  # a rank that dies must end the job, not hang it
        if args.backend == "nccl":
            dist.init_process_group("nccl", device_id=ctx.dev, rank=ctx.rank, world_size=ctx.world, timeout=limit)
        else:
            dist.init_process_group("gloo", rank=ctx.rank, world_size=ctx.world, timeout=limit)

    headline_wl = args.workload or "cfg3"
    strong = args.scaling == "strong"
    head = run_workload(ctx, headline_wl, args.pairs, args.steps, args.warmup, strong=strong,
                        cpu_budget_s=10.0, cpu_sample=args.cpu_sample, host_buffers=args.host_buffers or args.workload is None,  # (the default line: the headline's PCIe-inclusive rate beside it, never `value`)
                        two_streams=bool(args.two_streams and args.workload and ctx.world == 1))
    secondary = []
    if args.workload is None and not args.no_secondary:
        sec_steps = max(3, min(args.steps, 10))
        plan = [("cfg2", 0, False, dict(host_buffers=True)), ("cfg4", 0, False, {}), ("cfg5", 0, False, {}),
                ("cfg5", 10_000_000, True, {}), ("cfg3u", 0, False, {}), ("cfg2f", 0, False, {})]
        if ctx.world == 1:  # what the drain phases of the headline's kernels cost: the same steps, two batches in flight
            plan += [("cfg3", 0, False, dict(two_streams=True)), ("cfg4", 0, False, dict(two_streams=True))]
            plan.append(("cfg4d", 0, False, {}))  # configs[3]'s distance() variant: ~3 400 RSS tests + 215 triangle pairs per query
            plan.append(("cfg4s", 0, False, {}))  # configs[3]'s models against convex solids
            plan.append(("cfgmix", 0, False, {}))  # meshes and solids in one batch: the mesh walks beside the solids' kernels
            # what a planner sends: 20 000 pairs per batch -- every kernel a chain of dependent steps on a chip it does not fill, independent chains beside each other
            plan.append(("cfg5", 20_000, False, {}))
        if ctx.world > 1:  # the same list with the 24-B exchange format, and the headline without any exchange
            plan += [("cfg5", 10_000_000, True, dict(gather="compact")), ("cfg3", 0, False, dict(gather="none"))]
        for wl_name, pairs, st, kw in plan:
            try:
                r = run_workload(ctx, wl_name, pairs, (sec_steps if not st else 5) if wl_name != "cfg4d" else 2, 2 if wl_name != "cfg4d" else 1,
                                 strong=st, cpu_budget_s=2.5,
                                 cpu_sample=100_000 if wl_name not in ("cfg4", "cfg4d", "cfgmix") else (20_000 if wl_name != "cfg4d" else 2_000), **kw)
            except Exception as e:  # a secondary must never take the headline down
                r = {"workload": wl_name + ("/strong" if st else ""), "error": repr(e)} if ctx.rank == 0 else None
            if r is not None:
                if st:
                    r["workload"] += " (one %d-pair list sharded over the ranks, gather=%s)" % (pairs, r["config"]["gather"]) \
                        if "config" in r else ""
                secondary.append(r)

    line = None
    if ctx.rank == 0:
        line = {
            "metric": "narrow-phase queries/s (collision+distance)", "value": head["value"], "unit": "queries/s",
            "n_gpus": ctx.world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": head["ms_per_step"],
            "higher_is_better": True, "scaling": head["scaling"], "vs_baseline": None, "dtype": head["dtype"], "data": "synthetic",
            "config": head["config"], "roofline": head["roofline"], "cpu_baseline": head["cpu_baseline"],
        }
        if "host_buffers" in head:
            line["host_buffers"] = head["host_buffers"]
        if secondary:
            line["secondary"] = secondary
    if ctx.dist_on:
        dist.barrier()
        dist.destroy_process_group()
    if line is not None:
        # the JSON line goes out last: RCCL writes a version banner to the C stdout, which would otherwise be
        # flushed after Python's buffer when stdout is a file or pipe
        import ctypes
        try:
            ctypes.CDLL(None).fflush(None)
        except Exception:
            pass
        sys.stdout.flush()
        full = json.dumps(line)
        try:
            with open(os.path.join(os.environ.get("HFCL_BENCH_FULL_DIR", ROOT), "bench_full.json"), "w") as f:
                f.write(full + "\n")
        except OSError:
            pass
        # (not to stderr: the driver's 8 KB tail holds stdout AND stderr)
        print(compact_line(line), flush=True)  # the last stdout line: <= 4 KB


if __name__ == "__main__":
    main()
