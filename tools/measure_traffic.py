#!/usr/bin/env python
"""HBM traffic per kernel launch from rocprofv3 PMC counters -> profiles/traffic_<workload>.json.

Runs on the GPU box (`gpurun -- python tools/measure_traffic.py --workload cfg3`).  Two separate
`--pmc` passes (FETCH_SIZE needs 3 of the 4 TCC slots, WRITE_SIZE 2: MI355X_MICROARCH.md "rocprofv3
PMC slots"), each with --kernel-trace only.  Values are KB per dispatch as rocprofv3 reports them;
`bench.py` applies the gfx950 correction (FETCH_SIZE tallies 128-B requests at 64 B -> x2 for
wide reads; narrower reads uncalibrated, so raw is a lower bound and 2x raw an upper bound)."""
import argparse
import collections
import glob
import json
import os
import sqlite3
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def one_pass(counters, workload, outdir, pairs):
    """One rocprofv3 pass collecting `counters` (str or list) -> {counter: {kernel: (mean per dispatch, dispatches)}}"""
    single = isinstance(counters, str)
    names = [counters] if single else list(counters)
    env = dict(os.environ, TMPDIR="/tmp")
    cmd = ["rocprofv3", "--kernel-trace", "--pmc"] + names + ["-d", outdir, "-o", "t", "--",
           sys.executable, os.path.join(ROOT, "bench.py"), "--workload", workload, "--steps", "3",
           "--warmup", "1", "--no-cpu-baseline"] + (["--pairs", str(pairs)] if pairs else [])
    subprocess.run(cmd, check=True, env=env, cwd=ROOT, stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL, timeout=900)
    dbs = glob.glob(os.path.join(outdir, "**", "*.db"), recursive=True)
    assert dbs, "rocprofv3 wrote no database under " + outdir
    con = sqlite3.connect(dbs[0])
    res = {}
    for c in names:
        # per dispatch: the counter summed over its instances (XCDs / SEs report separately)
        rows = con.execute("select kernel_name, dispatch_id, sum(value) from counters_collection "
                           "where counter_name=? group by kernel_name, dispatch_id", (c,))
        per = collections.defaultdict(list)
        for k, _, v in rows:
            per[k].append(v)
        # PER PASS OF THE PIPELINE, like the kernel times of the bench line (the HIP events of a timer label span all
        # launches of that kernel within one pass: the twelve task-level launches of k_bvh_collide are one figure
        # there): every counter summed over all dispatches of the kernel, divided by the number of passes = dispatches
        # of k_classify (one per batch, or per half of a split batch).  Round 2 averaged over the "heavy" dispatches
        # only, which did not match the summed times.
        passes = max([len(v) for k, v in per.items() if "k_classify" in k] or [1])
        res[c] = {k: (sum(vals) / passes, len(vals) / passes) for k, vals in per.items()}
    return res[names[0]] if single else res


if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    ap.add_argument("--workload", default="cfg3")
    ap.add_argument("--pairs", type=int, default=0)
    ap.add_argument("--out", default=None)
    a = ap.parse_args()
    scratch = os.path.join(ROOT, "gpurun_out", "traffic_" + a.workload)
    res = collections.defaultdict(dict)
    for c in ("FETCH_SIZE", "WRITE_SIZE"):
        for k, (v, n) in one_pass(c, a.workload, os.path.join(scratch, c), a.pairs).items():
            res[k][c + "_KB_per_dispatch"] = v  # (per pass of the pipeline: all dispatches of the kernel in one batch)
            res[k]["dispatches"] = n
    # issue-side counters in their own pass: wave instructions by type per dispatch (the iterative kernels are bound
    # by VALU issue, not by HBM: bench.py reports SQ_INSTS_VALU against the chip's issue peak next to the HBM fraction)
    sq = ("SQ_INSTS_VALU", "SQ_INSTS_SALU", "SQ_INSTS_LDS", "SQ_WAVE_CYCLES", "SQ_BUSY_CYCLES")
    try:
        for c, per_kernel in one_pass(sq, a.workload, os.path.join(scratch, "SQ"), a.pairs).items():
            for k, (v, n) in per_kernel.items():
                res[k][c + "_per_dispatch"] = v
    except Exception as e:  # the traffic figures stand on their own
        print("SQ pass failed:", e, file=sys.stderr)
    out = a.out or os.path.join(ROOT, "gpurun_out", "traffic_%s.json" % a.workload)
    sys.path.insert(0, ROOT)
    from bench import kernel_source_sha  # the pass is only quoted by bench.py for the device code it was taken on
    json.dump({"workload": a.workload, "pairs": a.pairs or None, "source_sha": kernel_source_sha(),
               "unit": "KB as reported by rocprofv3 (uncorrected), summed over the kernel's dispatches within one pass of the pipeline",
               "kernels": {k: v for k, v in res.items() if k.startswith("void k_") or k.startswith("k_")}},
              open(out, "w"), indent=1)
    print(open(out).read())
