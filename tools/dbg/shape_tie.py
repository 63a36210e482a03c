import os, sys
sys.path.insert(0, "/root/repo"); sys.path.insert(0, "/root/repo/tests")
import numpy as np
from __graft_entry__ import load_pkg
import oracle_binding as ob
pkg = load_pkg()
abi, wl, bb = pkg.abi, pkg.workloads, pkg.bvh_builder
kind = sys.argv[1] if len(sys.argv) > 1 else "box"
b = wl.mesh_vs_solid(kind, n=3000, seed=5, half_width=2.0)
ML = bb.MeshLibrary(b.meshes)
req = abi.default_distance_request()
ref = ob.mixed_distance_batch(b.shapes, b.verts, ML, b.s1, b.s2, b.tf1, b.tf2, req, n_threads=16)
for name, env in (("default", {}), ("pool-16", dict(HFCL_SHAPE_DIST_BUDGET="16")), ("pool-16 no rerun", dict(HFCL_SHAPE_DIST_BUDGET="16", HFCL_POOL_RERUN="0")),
                  ("pool-16 rerun all", dict(HFCL_SHAPE_DIST_BUDGET="16", HFCL_POOL_RERUN="2")), ("default rerun all", dict(HFCL_POOL_RERUN="2")), ("lanes", dict(HFCL_SHAPE_DIST_BUDGET="0"))):
    os.environ.update(env)
    lib = wl.make_library(pkg, b)
    for k in env: os.environ.pop(k)
    got = lib.distance(b.s1, b.s2, b.tf1, b.tf2, req)
    rr = lib.last_ordered_reruns()
    lib.close()
    bad = np.flatnonzero(got["b1"] != ref["b1"])
    print(name, rr, "bad", bad[:10])
    for k in bad[:5]:
        print("   rec", k, "got b1", got["b1"][k], "ref b1", ref["b1"][k], "dist", repr(got["distance"][k]), repr(ref["distance"][k]), "status", hex(got["status"][k]), hex(ref["status"][k]))
        for t in (got["b1"][k], ref["b1"][k]):
            print("      leaf", t, repr(ob.mixed_leaf_distance(b.shapes, b.verts, ML, b.s1[k], b.s2[k], b.tf1[k], b.tf2[k], t, req)))
