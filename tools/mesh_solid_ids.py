#!/usr/bin/env python
"""distance() between meshes and solids: triangle ids, distances and witness points of the device path against the oracle, record by
record (on the GPU box).  usage: tools/mesh_solid_ids.py [n] [kinds,comma-separated|scene] ; HFCL_LIB_PATH / the HFCL_SHAPE_DIST_* knobs apply."""
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
kinds = (sys.argv[2] if len(sys.argv) > 2 else "scene,sphere,box,capsule,ellipsoid,convex32,mixed").split(",")
pkg = load_pkg()
abi, wl, bb = pkg.abi, pkg.workloads, pkg.bvh_builder
print("lib:", os.environ.get("HFCL_LIB_PATH", "in-tree"), " knobs:", {k: v for k, v in os.environ.items() if k.startswith("HFCL_") and k != "HFCL_LIB_PATH"})
for kind in kinds:
    b = wl.mesh_vs_shapes(n=n, seed=10, half_width=1.3) if kind == "scene" else wl.mesh_vs_solid(kind, n=n, seed=5, half_width=2.0)
    ML = bb.MeshLibrary(b.meshes)
    req = abi.default_distance_request()
    lib = wl.make_library(pkg, b)
    got = lib.distance(b.s1, b.s2, b.tf1, b.tf2, req)
    t0 = time.perf_counter()
    got = lib.distance(b.s1, b.s2, b.tf1, b.tf2, req)
    t_host = time.perf_counter() - t0
    kb = lib.last_kernel_breakdown()
    lib.close()
    ref = ob.mixed_distance_batch(b.shapes, b.verts, ML, b.s1, b.s2, b.tf1, b.tf2, req, n_threads=os.cpu_count() or 8)
    k = b.shapes["type"]
    mixed = (k[b.s1] == abi.BV_OBBRSS) != (k[b.s2] == abi.BV_OBBRSS)
    sep = mixed & (ref["distance"] > 0)
    same = (got["b1"] == ref["b1"]) & (got["b2"] == ref["b2"])
    eq_d = got["distance"] == ref["distance"]
    eq_w = (got["p1"] == ref["p1"]).all(axis=1) & (got["p2"] == ref["p2"]).all(axis=1)
    eq_s = got["status"] == ref["status"]
    print("%-10s n=%d mixed=%d separated=%.3f | ids equal: mixed %.5f separated %.5f (%d differ) | distance bit-equal %.5f (sep %.5f) | witness bit-equal sep %.5f | "
          "status equal %.5f | max|dd| %.3g | kernels %s" % (
              kind, n, int(mixed.sum()), sep.sum() / max(1, mixed.sum()), same[mixed].mean(), same[sep].mean(), int((~same[sep]).sum()),
              eq_d[mixed].mean(), eq_d[sep].mean(), eq_w[sep].mean(), eq_s[mixed].mean(),
              np.nanmax(np.abs(got["distance"][mixed] - ref["distance"][mixed])), [(a, round(v, 2)) for a, v in kb if v > 0.05]))
    bad = np.flatnonzero(mixed & ~same)
    nt = 0
    for k in bad[:40]:
        dg = ob.mixed_leaf_distance(b.shapes, b.verts, ML, b.s1[k], b.s2[k], b.tf1[k], b.tf2[k], got["b1"][k], req)
        nt += dg == ref["distance"][k]
        if k in bad[:6] or dg != ref["distance"][k]:
            print("           record %d: device triangle %d (oracle leaf value %.17g, device distance %.17g), oracle triangle %d at %.17g" % (
                k, got["b1"][k], dg, got["distance"][k], ref["b1"][k], ref["distance"][k]))
    if len(bad):
        print("           %d of the first %d differing records report a triangle at exactly the oracle's distance (0-ulp ties)" % (nt, min(40, len(bad))))
    pen = mixed & ~sep
    if pen.any():
        print("           penetrating: ids equal %.5f  distance bit-equal %.5f  max|dd| %.3g" % (
            same[pen].mean(), eq_d[pen].mean(), np.nanmax(np.abs(got["distance"][pen] - ref["distance"][pen]))))
