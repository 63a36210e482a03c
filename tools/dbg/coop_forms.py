import os, sys
sys.path.insert(0, "/root/repo"); sys.path.insert(0, "/root/repo/tests")
import numpy as np
from __graft_entry__ import load_pkg
pkg = load_pkg()
abi, wl = pkg.abi, pkg.workloads
from test_bvh_shape import _device_collide
kind = sys.argv[1] if len(sys.argv) > 1 else "ellipsoid"
b = wl.mesh_vs_solid(kind, n=6000, seed=3)
req = abi.default_collision_request()
got = _device_collide(pkg, b, req)
forms = {"tiny": dict(HFCL_SHAPE_BUDGET0="8", HFCL_SHAPE_BUDGET="8", HFCL_SHAPE_LEAF_COST="8"), "whole": dict(HFCL_SHAPE_LEVELS="1")}
for name, env in forms.items():
    o = _device_collide(pkg, b, req, env=env)
    for f in ("distance", "p1", "p2", "normal"):
        a, c = o[f], got[f]
        fin = np.isfinite(a) & np.isfinite(c)
        d = np.abs(a[fin] - c[fin])
        print(kind, name, f, "max diff %.3g" % (d.max() if d.size else 0), "n>1e-12:", int((d > 1e-12).sum()), "nan pattern equal:", np.array_equal(np.isnan(a), np.isnan(c)))
    print("  ids equal", np.array_equal(o["b1"], got["b1"]), "contacts equal", np.array_equal(o["num_contacts"], got["num_contacts"]))
