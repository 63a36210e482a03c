#!/usr/bin/env python
"""Soak of mesh distance(): the default kernels (long walks continued several to a wave, their tests pooled; walks whose reported pair
could hang on a rounding error walked again in order) against the lanes' sequential walk (budget 0 = the reference's order of visits),
on fresh random batches: EVERY BYTE of every record must agree -- distances, triangle ids, witness points.  No oracle involved.

  tools/distance_order_soak.py [--seeds 4] [--n 100000] [--rerun 1]     (--rerun 2: every pooled walk takes the ordered mode as well)"""
import argparse
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from __graft_entry__ import load_pkg  # noqa: E402


def run(pkg, b, env):
    abi, wl = pkg.abi, pkg.workloads
    old = {k: os.environ.get(k) for k in env}
    os.environ.update(env)
    try:
        lib = wl.make_library(pkg, b)
    finally:
        for k, v in old.items():
            os.environ.pop(k, None) if v is None else os.environ.__setitem__(k, v)
    try:
        r = lib.distance(b.s1, b.s2, b.tf1, b.tf2, abi.default_distance_request())
        return r, lib.last_ordered_reruns()
    finally:
        lib.close()


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--seeds", type=int, default=4)
    ap.add_argument("--n", type=int, default=100000)
    ap.add_argument("--rerun", default="1")
    a = ap.parse_args()
    pkg = load_pkg()
    wl = pkg.workloads
    cases = [("mesh x mesh", lambda s: wl.cfg4_mesh_mesh_distance(n=a.n, seed=700 + s), dict(HFCL_BVHD_BUDGET="0"), {})]
    for kind in ("mixed", "box", "cylinder", "cone", "sphere", "capsule", "ellipsoid", "convex32"):
        n = a.n if kind == "mixed" else a.n // 4
        cases.append(("mesh x %s" % kind, lambda s, kind=kind, n=n: wl.mesh_vs_solid(kind, n=n, seed=800 + s, half_width=2.0), dict(HFCL_SHAPE_DIST_BUDGET="0"), {}))
        cases.append(("mesh x %s, budget 16" % kind, lambda s, kind=kind, n=n: wl.mesh_vs_solid(kind, n=n // 4, seed=900 + s, half_width=2.0),
                      dict(HFCL_SHAPE_DIST_BUDGET="0"), dict(HFCL_SHAPE_DIST_BUDGET="16")))
    cases.append(("mesh x mesh, budget 16", lambda s: wl.cfg4_mesh_mesh_distance(n=a.n // 4, seed=1000 + s), dict(HFCL_BVHD_BUDGET="0"), dict(HFCL_BVHD_BUDGET="16")))
    print("# default kernels (HFCL_POOL_RERUN=%s) against the lanes' sequential walk; %d seeds" % (a.rerun, a.seeds))
    print("%-30s %10s %12s %10s %12s" % ("case", "queries", "continued", "re-run", "bytes differ"))
    bad_total = 0
    for name, gen, seq_env, env in cases:
        tot = cont = rer = bad = 0
        for s in range(a.seeds):
            try:
                b = gen(s)
            except Exception as e:  # a solid kind the scene generator does not make
                print("%-30s skipped (%s)" % (name, e))
                break
            r0, rr = run(pkg, b, dict(env, HFCL_POOL_RERUN=a.rerun))
            r1, _ = run(pkg, b, seq_env)
            v0, v1 = r0.view(np.uint8).reshape(len(r0), -1), r1.view(np.uint8).reshape(len(r1), -1)
            d = np.flatnonzero((v0 != v1).any(axis=1))
            tot += len(b)
            cont += rr["mesh_continued"] + rr["solid_continued"]
            rer += rr["mesh_rerun"] + rr["solid_rerun"]
            bad += len(d)
            for k in d[:3]:
                print("   seed %d record %d: default b1 %d b2 %d d %.17g | sequential b1 %d b2 %d d %.17g" % (
                    s, k, r0["b1"][k], r0["b2"][k], r0["distance"][k], r1["b1"][k], r1["b2"][k], r1["distance"][k]))
        else:
            print("%-30s %10d %12d %10d %12d" % (name, tot, cont, rer, bad), flush=True)
            bad_total += bad
    print("TOTAL records that differ: %d" % bad_total)
    return 1 if bad_total else 0


if __name__ == "__main__":
    sys.exit(main())
