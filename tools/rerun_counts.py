#!/usr/bin/env python
"""distance() on meshes: how many walks a wave continued and how many of those were re-run in the reference's order (GPU).
usage: python tools/rerun_counts.py [n]"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from __graft_entry__ import load_pkg  # noqa: E402

pkg = load_pkg()
abi, wl = pkg.abi, pkg.workloads
n = int(sys.argv[1]) if len(sys.argv) > 1 else 100_000
for name, b in (("cfg4d mesh x mesh", wl.cfg4_mesh_mesh_distance(n=n, seed=1)), ("mesh x solid mixed", wl.mesh_vs_solid("mixed", n=n, seed=2))):
    lib = wl.make_library(pkg, b)
    lib.distance(b.s1, b.s2, b.tf1, b.tf2, abi.default_distance_request())
    print(name, n, lib.last_ordered_reruns())
    lib.close()
