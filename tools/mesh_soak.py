#!/usr/bin/env python
"""Soak of the mesh walks: on fresh random batches the forms a long walk can take -- continued by a wave, and walked in one
piece by its lane -- must give the same records (no oracle involved: the GPU against itself, millions of queries).

  tools/mesh_soak.py [--seeds 8] [--n 50000]"""
import argparse
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from __graft_entry__ import load_pkg  # noqa: E402


def run(pkg, b, kind, env):
    abi, wl = pkg.abi, pkg.workloads
    old = {k: os.environ.get(k) for k in env}
    os.environ.update(env)
    try:
        lib = wl.make_library(pkg, b)
    finally:
        for k, v in old.items():
            os.environ.pop(k, None) if v is None else os.environ.__setitem__(k, v)
    try:
        if kind == "collide":
            return lib.collide(b.s1, b.s2, b.tf1, b.tf2, abi.default_collision_request())
        return lib.distance(b.s1, b.s2, b.tf1, b.tf2, abi.default_distance_request())
    finally:
        lib.close()


def compare(a, c, kind, exact=False):
    """-> (queries whose decision / triangle ids differ, max |distance difference|, queries with another triangle at the same distance)"""
    fin = (np.abs(a["distance"]) < 1e300) & (np.abs(c["distance"]) < 1e300)
    dd = float(np.abs(a["distance"][fin] - c["distance"][fin]).max()) if fin.any() else 0.0
    hard = (a["status"] != c["status"]) | (a["num_contacts"] != c["num_contacts"]) | (fin != (np.abs(a["distance"]) < 1e300))
    ids = (a["b1"] != c["b1"]) | (a["b2"] != c["b2"])
    if kind == "collide":
        return int((hard | ids).sum()), dd, 0
    if exact:
        # mesh x mesh distance() (round 4: its unit is built without contraction): the distances of the two forms agree to 0 ulp; another
        # triangle pair may be reported only AT that distance (the enumerated class of tests/test_gpu_parity.py: _check_distance_records)
        tie = ids & (a["distance"] == c["distance"])
        return int((hard | (ids & ~tie) | (a["distance"] != c["distance"])).sum()), dd, int(tie.sum())
    tie = ids & (np.abs(a["distance"] - c["distance"]) < 1e-12)  # mesh x solid distance(): tied triangles, an ulp apart between two inlined leaves
    return int((hard | (ids & ~tie)).sum()), dd, int(tie.sum())


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--seeds", type=int, default=8)
    ap.add_argument("--n", type=int, default=50000)
    a = ap.parse_args()
    pkg = load_pkg()
    wl = pkg.workloads
    cases = [
        ("mesh x mesh collide", lambda s: wl.cfg4_mesh_mesh(n=a.n, seed=100 + s), "collide", dict(HFCL_BVH_COOP="0", HFCL_BVH_LEVELS="1")),
        ("mesh x solid collide", lambda s: wl.mesh_vs_solid("mixed", n=a.n, seed=200 + s), "collide", dict(HFCL_SHAPE_LEVELS="1")),
        ("mesh x mesh distance", lambda s: wl.cfg4_mesh_mesh_distance(n=a.n // 10, seed=300 + s), "distance", dict(HFCL_BVHD_BUDGET="0")),
        ("mesh x solid distance", lambda s: wl.mesh_vs_solid("mixed", n=a.n // 2, seed=400 + s, half_width=2.0), "distance", dict(HFCL_SHAPE_DIST_BUDGET="0")),
    ]
    print("# wave continuation (default) against walks in one piece; %d seeds" % a.seeds)
    print("%-24s %10s %10s %14s %8s" % ("case", "queries", "differ", "max |dd|", "ties"))
    for name, gen, kind, env in cases:
        tot = bad = ties = 0
        worst = 0.0
        for s in range(a.seeds):
            b = gen(s)
            r0, r1 = run(pkg, b, kind, {}), run(pkg, b, kind, env)
            nb, dd, nt = compare(r0, r1, kind, exact=name == "mesh x mesh distance")
            tot += len(b)
            bad += nb
            ties += nt
            worst = max(worst, dd)
        print("%-24s %10d %10d %14.3e %8d" % (name, tot, bad, worst, ties), flush=True)


if __name__ == "__main__":
    main()
