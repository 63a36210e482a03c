#!/usr/bin/env python
"""distance() on cfg4's mesh pairs: triangle ids, distances and witness points of the device path against the oracle, record by
record (on the GPU box).  usage: tools/cfg4d_ids.py [n] [seed]; the HFCL_BVHD_* knobs of the environment apply."""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
from __graft_entry__ import load_pkg  # noqa: E402
import oracle_binding as ob  # noqa: E402  (checker)

n = int(sys.argv[1]) if len(sys.argv) > 1 else 20000
seed = int(sys.argv[2]) if len(sys.argv) > 2 else 1
pkg = load_pkg()
wl, bb = pkg.workloads, pkg.bvh_builder
b = wl.cfg4_mesh_mesh_distance(n=n, seed=seed)
ML = bb.MeshLibrary(b.meshes)
lib = wl.make_library(pkg, b)
got = lib.distance(b.s1, b.s2, b.tf1, b.tf2)
t0 = time.perf_counter()
got = lib.distance(b.s1, b.s2, b.tf1, b.tf2)
t_host = time.perf_counter() - t0
kb = lib.last_kernel_breakdown()
lib.close()
ref = ob.bvh_distance_batch(ML, b.s1, b.s2, b.tf1, b.tf2, n_threads=os.cpu_count() or 8)
pos = ref["distance"] > 0
same = (got["b1"] == ref["b1"]) & (got["b2"] == ref["b2"])
eq_d = got["distance"] == ref["distance"]
print("knobs:", {k: v for k, v in os.environ.items() if k.startswith("HFCL_")})
print("n=%d separated=%.3f  host call %.1f ms  kernels %s" % (n, pos.mean(), 1e3 * t_host, [(k, round(v, 2)) for k, v in kb if v > 0.05]))
print("ids equal: all %.5f  separated %.5f (%d differ)   distance bit-equal: %.5f   max |dd| %.3g" % (
    same.mean(), same[pos].mean(), int((~same[pos]).sum()), eq_d.mean(), np.abs(got["distance"] - ref["distance"]).max()))
print("witness max |dp| (records with equal ids): %.3g" % max(np.abs(got["p1"][same & pos] - ref["p1"][same & pos]).max(),
                                                               np.abs(got["p2"][same & pos] - ref["p2"][same & pos]).max()))
print("overflow flags:", int(((got["status"] >> 30) & 1).sum()))
bad = np.where(~same)[0][:8]
for k in bad:
    print("  pair %d: got (%d,%d) d=%.17g   ref (%d,%d) d=%.17g" % (k, got["b1"][k], got["b2"][k], got["distance"][k], ref["b1"][k], ref["b2"][k], ref["distance"][k]))
