#!/usr/bin/env python
"""Per-kernel PMC counters of one bench.py workload (rocprofv3 --kernel-trace --pmc, counters in passes of 4).

usage (on the GPU box): tools/pmc.py --workload cfg3 --kernels k_epa_stream,k_gjk_cvx [--lib build/ab/libX.so] C1 C2 ...
Prints, per kernel, the mean per dispatch of every counter."""
import argparse
import collections
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tools"))
from measure_traffic import one_pass  # noqa: E402

if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    ap.add_argument("--workload", default="cfg3")
    ap.add_argument("--pairs", type=int, default=0)
    ap.add_argument("--kernels", default="")
    ap.add_argument("--lib", default="")
    ap.add_argument("--tag", default="pmc")
    ap.add_argument("counters", nargs="+")
    a = ap.parse_args()
    if a.lib:
        os.environ["HFCL_LIB_PATH"] = os.path.abspath(a.lib)
    res = collections.defaultdict(dict)
    for i in range(0, len(a.counters), 4):
        grp = a.counters[i:i + 4]
        out = os.path.join(ROOT, "gpurun_out", "%s_%s_%d" % (a.tag, a.workload, i))
        try:
            for c, per_kernel in one_pass(grp, a.workload, out, a.pairs).items():
                for k, (v, n) in per_kernel.items():
                    res[k][c] = v
        except Exception as e:
            print("pass %s failed: %s" % (grp, e), file=sys.stderr)
    want = [w for w in a.kernels.split(",") if w]
    for k in sorted(res):
        if want and not any(w in k for w in want):
            continue
        print(k[:100])
        for c in a.counters:
            if c in res[k]:
                print("    %-28s %16.6g" % (c, res[k][c]))
