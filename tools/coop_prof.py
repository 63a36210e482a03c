#!/usr/bin/env python
"""How long the walks of k_bvh_shape_coop / k_bvh_coop and their waves live (a library built with -DHFCL_COOP_PROF: tools/build_variant.sh
cprof k_bvh -DHFCL_COOP_PROF, selected with HFCL_LIB_PATH): the balance of the continuation kernels.  usage (GPU box): tools/coop_prof.py [n] [kinds]
kinds: mesh (cfg4's mesh x mesh collide) or a solid kind of tools/mesh_solid_bench.py (mixed, sphere, ...)"""
import ctypes as C
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from __graft_entry__ import load_pkg  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 100000
kinds = sys.argv[2].split(",") if len(sys.argv) > 2 else ["mesh", "mixed", "ellipsoid"]
pkg = load_pkg()
wl = pkg.workloads
dll = pkg.engine.dll()
print("knobs:", {k: x for k, x in os.environ.items() if k.startswith("HFCL_") and k != "HFCL_LIB_PATH"})
for kind in kinds:
    b = wl.cfg4_mesh_mesh(n=n, seed=1) if kind == "mesh" else wl.mesh_vs_solid(kind, n=n, seg=50)
    lib = wl.make_library(pkg, b)
    out = (C.c_ulonglong * 32)()
    lib.collide(b.s1, b.s2, b.tf1, b.tf2)
    dll.hfcl_debug_coop_prof(out, 1)
    lib.collide(b.s1, b.s2, b.tf1, b.tf2)
    dll.hfcl_debug_coop_prof(out, 1)
    v = np.array(list(out), dtype=np.float64)
    print("%-10s %d queries: units %d (cut %d): longest %.0f ticks, mean %.0f; waves with a unit %d: longest %.0f ticks, mean %.0f, sum %.3g" % (
        kind, n, v[2], v[6], v[0], v[1] / max(v[2], 1), v[5], v[3], v[4] / max(v[5], 1), v[4]))
    if v[7] > 0:  # k_bvh_shape_coop's trips
        print("           trips %d (%.1f per unit): box tests %.1f lanes / trip, %.0f ticks / trip; leaf batches %d (%.2f of trips), %.1f lanes, %.0f ticks each; "
              "share of wave time: boxes %.2f, leaf batches %.2f, contact trips' tail %.2f, drawing and loading units %.2f" % (
                  v[7], v[7] / max(v[2], 1), v[9] / v[7], v[8] / v[7], v[10], v[10] / v[7], v[12] / max(v[10], 1), v[11] / max(v[10], 1),
                  v[8] / v[4], v[11] / v[4], v[13] / v[4], v[14] / v[4]))
        print("           scans %.2f, witness part %.2f, stack rewrite %.2f, the query's record %.2f" % (v[15] / v[4], v[16] / v[4], v[17] / v[4], v[18] / v[4]))
        print("           per trip: window %.1f entries of a stack of %.1f: %.1f fresh boxes, %.1f disjoint boxes waiting, %.1f evaluated triangles waiting, %.1f triangles "
              "waiting for a batch; %.1f entries visited" % (v[19] / v[7], v[22] / v[7], v[9] / v[7] - v[23] / v[7], v[20] / v[7], v[21] / v[7], v[23] / v[7], v[24] / v[7]))
    lib.close()
