"""mesh x solid collide(): the three-kernel phase (shape_walk = 1) against k_bvh_collide's SOLID form (shape_walk = 0) at several batch sizes -- records byte for byte, ms per batch.
usage (GPU box): tools/dbg/shape_walk_sizes.py [sizes, comma-separated]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
from __graft_entry__ import load_pkg
pkg = load_pkg(); wl = pkg.workloads
sizes = [int(x) for x in sys.argv[1].split(",")] if len(sys.argv) > 1 else [300, 5000, 50000, 400000, 1000000]
for n in sizes:
    b = wl.mesh_vs_solid("mixed", n=n, seed=7)
    recs, ms = {}, {}
    for sw in (1, 0):
        lib = wl.make_library(pkg, b, options={"shape_walk": sw})
        try:
            lib.collide(b.s1, b.s2, b.tf1, b.tf2)
            t0 = time.perf_counter()
            for _ in range(3):
                recs[sw] = lib.collide(b.s1, b.s2, b.tf1, b.tf2)
            ms[sw] = (time.perf_counter() - t0) / 3 * 1e3
        finally:
            lib.close()
    same = recs[1].tobytes() == recs[0].tobytes()
    print("n=%8d  records identical: %s  contacts %.3f  host-boundary ms per batch (incl. copies): three kernels %.2f, SOLID form %.2f" % (
        n, same, (recs[1]["num_contacts"] > 0).mean(), ms[1], ms[0]), flush=True)
    assert same
