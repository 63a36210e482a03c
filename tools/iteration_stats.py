#!/usr/bin/env python
"""Where the iterations go, from the CPU oracle (no GPU needed): the statistics the lockstep kernels were re-arranged by.

  tools/iteration_stats.py cfg5   GJK and EPA iterations by pair of shape kinds on the mixed workload
                                  (profiles/r02_q: pairs with a curved shape take 2-3x the GJK and 4-6x the EPA iterations)
  tools/iteration_stats.py cfg3   GJK iterations against the separation of the pair (weak correlation: no class to sort by)
  tools/iteration_stats.py cfg4   cost of a mesh x mesh query in BV tests + 8 x triangle tests (profiles/r02_w)"""
import collections
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
from __graft_entry__ import load_pkg  # noqa: E402


def main():
    import oracle_binding as ob
    which = sys.argv[1] if len(sys.argv) > 1 else "cfg5"
    n = int(sys.argv[2]) if len(sys.argv) > 2 else (100_000 if which == "cfg4" else 200_000)
    pkg = load_pkg()
    abi, wl = pkg.abi, pkg.workloads
    threads = min(os.cpu_count() or 1, 64)
    if which == "cfg4":
        b = wl.cfg4_mesh_mesh(n=n)
        ML = pkg.bvh_builder.MeshLibrary(b.meshes)
        out, st = ob.bvh_collide_batch(ML, b.s1, b.s2, b.tf1, b.tf2, wl.make_request(b, abi), n_threads=threads, want_stats=True)
        cost = st[:, 0].astype(float) + 8 * st[:, 1].astype(float)
        hit = out["num_contacts"] > 0
        print("queries %d, colliding %.3f" % (n, hit.mean()))
        for name, x in (("BV tests", st[:, 0].astype(float)), ("triangle tests", st[:, 1].astype(float)), ("cost = BV + 8 x triangle", cost)):
            print("%-26s mean %.1f  p50 %.0f  p90 %.0f  p99 %.0f  p99.9 %.0f  max %.0f" % (
                name, x.mean(), np.median(x), np.quantile(x, .9), np.quantile(x, .99), np.quantile(x, .999), x.max()))
        for thr in (128, 256, 512, 1024, 2048):
            m = cost > thr
            print("cost > %4d: %5.2f %% of the queries hold %4.1f %% of the work (%.0f %% of them colliding)" % (
                thr, 100 * m.mean(), 100 * cost[m].sum() / cost.sum(), 100 * hit[m].mean() if m.any() else 0))
        return
    b = wl.cfg5_mixed(n=n) if which == "cfg5" else wl.cfg3_convex_convex(n=n)
    req = wl.make_request(b, abi)
    r = (ob.distance_batch if b.kind == "distance" else ob.collide_batch)(b.shapes, b.verts, b.s1, b.s2, b.tf1, b.tf2, req, n_threads=threads)
    gi = abi.status_gjk_iters(r["status"]).astype(float)
    ei = abi.status_epa_iters(r["status"]).astype(float)
    epa = abi.status_epa(r["status"]) != 15  # 15 = EPA did not run
    if which == "cfg3":
        d = r["distance"]
        print("GJK iterations: mean %.2f p50 %.0f p90 %.0f p99 %.0f max %.0f" % (gi.mean(), np.median(gi), np.quantile(gi, .9), np.quantile(gi, .99), gi.max()))
        qs = np.quantile(d, np.linspace(0, 1, 11))
        for lo, hi in zip(qs[:-1], qs[1:]):
            m = (d >= lo) & (d <= hi)
            print("distance in [%7.3f, %7.3f]: mean %.2f p90 %.0f p99 %.0f" % (lo, hi, gi[m].mean(), np.quantile(gi[m], .9), np.quantile(gi[m], .99)))
        return
    t1, t2 = b.shapes["type"][b.s1], b.shapes["type"][b.s2]
    names = {9: "box", 10: "sphere", 11: "capsule", 12: "cone", 13: "cylinder", 14: "convex", 19: "ellipsoid"}
    print("%d pairs, %.1f %% reach EPA" % (n, 100 * epa.mean()))
    for title, x, mask in (("GJK iterations", gi, gi > 0), ("EPA iterations", ei, epa)):
        print(title + " by pair of kinds (pairs that ran it)")
        groups = collections.defaultdict(list)
        for a, c, v in zip(t1[mask], t2[mask], x[mask]):
            groups[(min(a, c), max(a, c))].append(v)
        tot = sum(sum(v) for v in groups.values())
        for k, v in sorted(groups.items()):
            print("  %-10s x %-10s n %6d  mean %5.1f  p50 %3.0f  p90 %3.0f  p99 %3.0f  max %3.0f  (%4.1f %% of all iterations)" % (
                names.get(int(k[0]), k[0]), names.get(int(k[1]), k[1]), len(v), np.mean(v), np.median(v), np.quantile(v, .9),
                np.quantile(v, .99), max(v), 100 * sum(v) / tot))


if __name__ == "__main__":
    main()
