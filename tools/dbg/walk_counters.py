"""The counters of the walk / leaves / resolve phase of a workload's last batch (hfcl_debug_walk_counters): queries, items listed, leaves that ended a walk
needing EPA, queries handed to the continuation kernel.  usage (GPU box): tools/dbg/walk_counters.py [cfg4s|cfg4|cfgmix] [n]"""
import ctypes as C, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
from __graft_entry__ import load_pkg
pkg = load_pkg(); wl = pkg.workloads
dll = pkg.engine.dll()
name = sys.argv[1] if len(sys.argv) > 1 else "cfg4s"
n = int(sys.argv[2]) if len(sys.argv) > 2 else 100000
b = {"cfg4s": lambda: wl.mesh_vs_solid("mixed", n=n), "cfg4": lambda: wl.cfg4_mesh_mesh(n=n, seed=1), "cfgmix": lambda: wl.mixed_scene(n=n, seed=1)}[name]()
lib = wl.make_library(pkg, b)
rec = lib.collide(b.s1, b.s2, b.tf1, b.tf2)
out = (C.c_uint32 * 34)()
for solid in (1, 0):
    dll.hfcl_debug_walk_counters(lib._h if hasattr(lib, "_h") else lib.handle, solid, out)
    v = list(out)
    print("%s walks: round 0: ticket %d, items %d, next-round queries %d, redo (EPA enders) %d; tasks %d, suspended queries %d" % ("mesh x solid" if solid else "mesh x mesh", v[0], v[1], v[2], v[5], v[32], v[33]))
print("records with a contact: %.3f" % (rec["num_contacts"] > 0).mean())
lib.close()
