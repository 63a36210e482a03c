#!/usr/bin/env python
"""How close to bit-exact is the fp64 device path?  Records of the HIP kernels against the CPU oracle on the parity suite's
workloads: whole records equal, statuses (incl. iteration counts) equal, distances equal.  With HFCL_LIB_PATH set: that build.
usage (GPU box): tools/fp64_exactness.py [n]"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
from __graft_entry__ import load_pkg  # noqa: E402
import oracle_binding as ob  # noqa: E402  (checker)

n = int(sys.argv[1]) if len(sys.argv) > 1 else 100000
pkg = load_pkg()
abi, wl = pkg.abi, pkg.workloads
print("lib:", os.environ.get("HFCL_LIB_PATH", "in-tree"))
cases = [("cfg2_box_capsule", {}, None), ("cfg3_convex_convex", {}, None), ("cfg5_mixed", {}, None), ("all_primitives", {}, None),
         ("cfg5_mixed", {"seed": 6}, "bvguess"), ("cfg5_mixed", {"seed": 12}, "crit10"), ("cfg5_mixed", {"seed": 12}, "crit20")]
for case, kw, mode in cases:
    b = getattr(wl, case)(n=n, **kw)
    if mode and mode.startswith("crit"):
        b.kind = "distance"
        req = abi.default_distance_request()
        req.q.gjk_variant = 2
        req.q.gjk_convergence_criterion, req.q.gjk_convergence_criterion_type = int(mode[4]), int(mode[5])
    else:
        req = wl.make_request(b, abi)
        if mode == "bvguess":
            req.q.gjk_initial_guess = abi.BoundingVolumeGuess
    fn = ob.distance_batch if b.kind == "distance" else ob.collide_batch
    ref = fn(b.shapes, b.verts, b.s1, b.s2, b.tf1, b.tf2, req, n_threads=min(64, os.cpu_count() or 8))
    lib = pkg.Library(b.lib, device=0)
    got = lib.distance(b.s1, b.s2, b.tf1, b.tf2, req) if b.kind == "distance" else lib.collide(b.s1, b.s2, b.tf1, b.tf2, req)
    lib.close()
    rec_eq = (got.view(np.uint8).reshape(len(got), -1) == ref.view(np.uint8).reshape(len(ref), -1)).all(axis=1)
    nanok = lambda a, c: (a == c) | (np.isnan(a) & np.isnan(c))  # noqa: E731
    d_eq = nanok(got["distance"], ref["distance"])
    st_eq = got["status"] == ref["status"]
    gi_eq = abi.status_gjk_iters(got["status"]) == abi.status_gjk_iters(ref["status"])
    ei_eq = abi.status_epa_iters(got["status"]) == abi.status_epa_iters(ref["status"])
    fl_eq = abi.status_contact(got["status"]) == abi.status_contact(ref["status"])
    w_eq = nanok(got["p1"], ref["p1"]).all(axis=1) & nanok(got["p2"], ref["p2"]).all(axis=1) & nanok(got["normal"], ref["normal"]).all(axis=1)
    dd = np.abs(got["distance"] - ref["distance"])
    dw = max(np.nanmax(np.abs(got[f] - ref[f])) for f in ("p1", "p2", "normal"))
    print("%-20s %-8s n=%d | status equal %.6f | gjk iters equal %.6f | epa iters equal %.6f | contact flags differ %d | distance bit-equal %.6f | "
          "witness+normal bit-equal %.6f | max|dd| %.3g | max witness/normal diff %.3g" % (case, mode or "", len(got), st_eq.mean(), gi_eq.mean(), ei_eq.mean(), int((~fl_eq).sum()),
                                                              d_eq.mean(), w_eq.mean(), np.nanmax(dd), dw))
