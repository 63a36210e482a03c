#!/usr/bin/env python
"""Cost of the two support forms of hulls far above the 32-vertex register form (k_gjk_large + full-capacity EPA):
the 16-lane vertex scan, and neighbour hill-climbing over a registered vertex adjacency (hfcl_lib_set_convex_neighbors).
Convex x convex distance() on pairs of V-vertex hulls, V = 64 ... 16384 ("bench a 1k- and 16k-vertex
hull workload").  Prints ms per batch, pairs/s and the time per support evaluation (two per GJK iteration) for both,
and how far the two sets of results are apart.

usage (GPU box): tools/large_hull_bench.py [--pairs 100000]"""
import argparse
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from __graft_entry__ import load_pkg  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--pairs", type=int, default=100_000)
    ap.add_argument("--sizes", default="64,256,1024,4096,16384")
    a = ap.parse_args()
    import torch
    pkg = load_pkg()
    abi, wl, g = pkg.abi, pkg.workloads, pkg.geometry
    dev = torch.device("cuda:0")
    rng = np.random.default_rng(3)
    n = a.pairs
    for V in [int(x) for x in a.sizes.split(",")]:
        nlib = 16
        lib = g.ShapeLibrary()
        for radii in rng.uniform(0.3, 1.0, (nlib, 3)):
            d = rng.normal(size=(V, 3))
            lib.add_convex(d / np.linalg.norm(d, axis=1, keepdims=True) * radii)  # every point a hull vertex
        s1, s2 = rng.integers(0, nlib, n), rng.integers(0, nlib, n)
        q1, T1, q2, T2 = wl._poses(rng, n, 1.2)
        b = wl.Batch("large_%d" % V, lib, s1, s2, q1, T1, q2, T2, "distance")
        req = wl.make_request(b, abi)
        d_in = [torch.from_numpy(x).to(dev) for x in (b.s1.astype(np.int32), b.s2.astype(np.int32), b.tf1, b.tf2)]
        st = torch.cuda.current_stream().cuda_stream
        recs = {}
        for form in ("scan", "climb"):
            os.environ["HFCL_CLIMB_MIN"] = "0"
            L = pkg.Library(lib)
            if form == "climb":
                t0 = time.perf_counter()
                wl.register_adjacency(L, b.shapes, b.verts)
                t_adj = time.perf_counter() - t0
            out = torch.zeros(n * 24, dtype=torch.int32, device=dev)
            L.distance_device(*d_in, n, req, out, stream=st)
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            reps = 3
            for _ in range(reps):
                L.distance_device(*d_in, n, req, out, stream=st)
            torch.cuda.synchronize()
            t = (time.perf_counter() - t0) / reps
            kb = dict(L.last_kernel_breakdown())
            rec = out.cpu().numpy().view(abi.RESULT_DTYPE).copy()
            recs[form] = rec
            it = abi.status_gjk_iters(rec["status"]).astype(np.float64)
            pen = (rec["distance"] < 0).mean()
            gjk_ms = kb.get("k_gjk_large", float("nan"))
            # supports per pair in GJK: two per iteration; lanes per pair: 16; groups resident: 256 CUs x 8 waves x 4
            sup = 2 * it.mean() * n
            print("V=%6d %-5s %8.3f ms per %d pairs = %7.2f M pairs/s  (k_gjk_large %.3f ms, EPA %.3f ms; %.1f GJK iterations, %.0f %% penetrating)  "
                  "=> %.2f us of one 16-lane group per support" % (
                      V, form, 1e3 * t, n, n / t / 1e6, gjk_ms, kb.get("k_epa<full>", 0.0), it.mean(), 100 * pen,
                      gjk_ms * 1e3 / (sup / (256 * 8 * 4))))
            L.close()
        a, c = recs["scan"], recs["climb"]
        dd = np.abs(a["distance"].astype(np.float64) - c["distance"].astype(np.float64))
        same = (a["status"] == c["status"]).mean()
        print("          scan vs climb: max |d distance| %.3g, p99.9 %.3g, identical status words %.4f %%, identical records %.4f %%  (adjacency of %d hulls built in %.2f s)" % (
            dd.max(), np.quantile(dd, 0.999), 100 * same,
            100 * np.mean([x.tobytes() == y.tobytes() for x, y in zip(a[:20000], c[:20000])]), nlib, t_adj))


if __name__ == "__main__":
    main()
