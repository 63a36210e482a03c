#!/usr/bin/env python
"""How long the walks of mesh x solid collide() queries are (host build of the device headers, tests/hostsim): nodes popped
and triangles tested per query on tools/mesh_solid_bench.py's scenes.  The distribution has a heavy tail -- most queries end
at the root, a few walk thousands of nodes with hundreds of leaf tests -- which is why the device form cuts walks into tasks."""
import ctypes as C
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
from __graft_entry__ import load_pkg  # noqa: E402
import hostsim_binding as hs  # noqa: E402


def main():
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 4000
    pkg = load_pkg()
    abi, bb = pkg.abi, pkg.bvh_builder
    hs.set_shape_lane(True)
    L = hs.lib()
    L.sim_shape_walk_steps.restype = C.POINTER(C.c_uint32)
    L.sim_shape_walk_leaves.restype = C.POINTER(C.c_uint32)
    print("%-10s %28s | %22s | %s" % ("solid", "nodes: mean p50 p90 p99 max", "leaves: mean p99 max", "share of the nodes in the longest 1 % of the walks"))
    for kind in "sphere,box,capsule,cylinder,ellipsoid,convex32".split(","):
        b = pkg.workloads.mesh_vs_solid(kind, n=n)
        ML = bb.MeshLibrary(b.meshes)
        hs.mesh_shape_collide_f64(abi, b.shapes, b.verts, ML, b.s1, b.s2, b.tf1, b.tf2, abi.default_collision_request())
        st = np.ctypeslib.as_array(L.sim_shape_walk_steps(), (n,)).copy()
        lv = np.ctypeslib.as_array(L.sim_shape_walk_leaves(), (n,)).copy()
        print("%-10s %6.0f %5d %5d %5d %5d | %8.1f %5d %5d | %.2f" % (kind, st.mean(), np.median(st), np.quantile(st, .9), np.quantile(st, .99), st.max(),
                                                                   lv.mean(), np.quantile(lv, .99), lv.max(), np.sort(st)[-n // 100:].sum() / st.sum()))


if __name__ == "__main__":
    main()
