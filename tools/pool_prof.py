#!/usr/bin/env python
"""Phase clocks and event counts of k_bvh_distance_pool (a library built with -DHFCL_POOL_PROF: tools/build_variant.sh prof k_bvhd
-DHFCL_POOL_PROF, selected with HFCL_LIB_PATH) on cfg4's distance() workload.  usage (GPU box): tools/pool_prof.py [n]"""
import ctypes as C
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from __graft_entry__ import load_pkg  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 100000
pkg = load_pkg()
wl = pkg.workloads
b = wl.cfg4_mesh_mesh_distance(n=n, seed=1)
lib = wl.make_library(pkg, b)
dll = pkg.engine.dll()
out = (C.c_ulonglong * 16)()
lib.distance(b.s1, b.s2, b.tf1, b.tf2)
dll.hfcl_debug_pool_prof(out, 1)
lib.distance(b.s1, b.s2, b.tf1, b.tf2)
dll.hfcl_debug_pool_prof(out, 1)
v = np.array(list(out), dtype=np.float64)
names = ["scan", "box tests", "triangle tests", "write-back", "refill"]
tot = v[:5].sum()
print("knobs:", {k: x for k, x in os.environ.items() if k.startswith("HFCL_") and k != "HFCL_LIB_PATH"})
for i, nm in enumerate(names):
    print("  %-16s %6.1f %% of the waves' clocks" % (nm, 100 * v[i] / tot))
trips, rounds, tests, tpass, ttests, walks = v[8], v[9], v[10], v[11], v[12], v[13]
print("  walks %d  trips %d (%.1f per walk)  box tests %.0f per walk in %d rounds (%.1f lanes per round, %.2f rounds per trip)" % (
    walks, trips, trips * 4 / max(walks, 1), tests / max(walks, 1), rounds, tests / max(rounds, 1), rounds / max(trips, 1)))
print("  triangle passes %d (one per %.1f trips), %.1f lanes per pass, %.0f triangle tests per walk" % (
    tpass, trips / max(tpass, 1), ttests / max(tpass, 1), ttests / max(walks, 1)))
print("  clocks per box round %.0f, per triangle pass %.0f, per scan %.0f, per write-back %.0f" % (
    v[1] / max(rounds, 1), v[2] / max(tpass, 1), v[0] / max(trips, 1), v[3] / max(trips, 1)))
