#!/usr/bin/env python
"""collide() with the long walks of k_bvh_coop / k_bvh_shape_coop cut into chunks (BvhSplit::cut_ticks) against the same walks in one piece:
every field of every record must be equal, wherever the cuts fall -- the GPU against itself.  tools/cut_check.py [n] [seeds]"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from __graft_entry__ import load_pkg  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 50000
seeds = int(sys.argv[2]) if len(sys.argv) > 2 else 2
only = sys.argv[3].split(",") if len(sys.argv) > 3 else None  # e.g. mesh,mixed,ellipsoid
pkg = load_pkg()
abi, wl = pkg.abi, pkg.workloads


def run(b, env):
    old = {k: os.environ.get(k) for k in env}
    os.environ.update(env)
    try:
        lib = wl.make_library(pkg, b)
    finally:
        for k, v in old.items():
            os.environ.pop(k, None) if v is None else os.environ.__setitem__(k, v)
    try:
        return lib.collide(b.s1, b.s2, b.tf1, b.tf2, abi.default_collision_request())
    finally:
        lib.close()


def differing(got, ref):
    diff = np.zeros(len(got), dtype=bool)
    for f in got.dtype.names:
        a, c = got[f], ref[f]
        d = ~((a == c) | (np.isnan(a) & np.isnan(c))) if a.dtype.kind == "f" else a != c
        diff |= d.reshape(len(got), -1).any(axis=1)
    return diff


bad_total = 0
cases = [("mesh x mesh", lambda s: wl.cfg4_mesh_mesh(n=n, seed=400 + s), "HFCL_BVH_CUT_TICKS")]
for kind in ("mixed", "sphere", "box", "capsule", "cylinder", "ellipsoid", "convex32", "cone"):
    cases.append(("mesh x " + kind, (lambda kd: lambda s: wl.mesh_vs_solid(kd, n=n, seed=300 + s))(kind), "HFCL_SHAPE_CUT_TICKS"))
if only:
    cases = [c for c in cases if c[0].split(" x ")[-1] in only]
for name, make, knob in cases:
    for s in range(seeds):
        b = make(s)
        ref = run(b, {knob: "0"})
        for ticks in ("600000", "100000", "15000"):
            got = run(b, {knob: ticks})
            diff = differing(got, ref)
            bad_total += int(diff.sum())
            print("%-18s seed %d cut after %7s ticks: %6d of %d records differ from the walks in one piece (contacts %.3f)" % (
                name, s, ticks, int(diff.sum()), len(got), float((ref["num_contacts"] > 0).mean())), flush=True)
            if diff.any():
                i = int(np.nonzero(diff)[0][0])
                print("   first:", i, {f: (got[f][i].tolist(), ref[f][i].tolist()) for f in got.dtype.names if np.any(got[f][i] != ref[f][i])})
print("TOTAL differing records:", bad_total)
