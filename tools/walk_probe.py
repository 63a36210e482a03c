#!/usr/bin/env python
"""Host-side experiments on cfg4's distance() walks with the oracle's arithmetic (CPU only; tools/walk_probe.cpp is compiled against
oracle/*.cpp into /tmp):
  ties     the sequential walk against "global minimum, the first pair in DFS order among equals" (what an order-free evaluation
           with a DFS tie-break computes): same distance / ids?  how many queries have several pairs at exactly the minimum?
  predict  what is known after 64 steps of a walk against the box tests it still needs (scheduling order, profiles/r04_h)
usage: tools/walk_probe.py ties|predict [n]"""
import ctypes as C
import os
import subprocess
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from __graft_entry__ import load_pkg  # noqa: E402

so = "/tmp/hfcl_walk_probe.so"
srcs = [os.path.join(ROOT, "tools", "walk_probe.cpp")] + [os.path.join(ROOT, "oracle", f) for f in
                                                          ("gjk.cpp", "narrowphase.cpp", "bvh.cpp", "bvh_build.cpp", "bvh_shape.cpp")]
subprocess.check_call(["g++", "-O3", "-DNDEBUG", "-march=x86-64-v2", "-std=c++17", "-fPIC", "-ffp-contract=off", "-shared", "-o", so] + srcs +
                      ["-I" + os.path.join(ROOT, "oracle"), "-lpthread"])
what = sys.argv[1] if len(sys.argv) > 1 else "ties"
n = int(sys.argv[2]) if len(sys.argv) > 2 else 20000
pkg = load_pkg()
abi, wl = pkg.abi, pkg.workloads
b = wl.cfg4_mesh_mesh_distance(n=n, seed=1)
ML = pkg.bvh_builder.MeshLibrary(b.meshes)
L = C.CDLL(so)
nodes = np.ascontiguousarray(ML.nodes)
m1, m2 = np.ascontiguousarray(b.s1, dtype=np.uint32), np.ascontiguousarray(b.s2, dtype=np.uint32)
tf1 = np.ascontiguousarray(b.tf1, dtype=np.float64).reshape(-1, 12)
tf2 = np.ascontiguousarray(b.tf2, dtype=np.float64).reshape(-1, 12)
args = (abi.ptr(nodes), abi.ptr(ML.verts), abi.ptr(ML.tris), abi.ptr(ML.table), C.c_size_t(len(ML.table)), abi.ptr(m1), abi.ptr(m2),
        abi.ptr(tf1), abi.ptr(tf2), C.c_size_t(n))
threads = C.c_int(os.cpu_count() or 8)
if what == "ties":
    out = np.zeros((n, 10))
    L.tie_probe(*args, abi.ptr(out), threads)
    sep = out[:, 0] > 0
    same_d = out[:, 0] == out[:, 1]
    same_id = (out[:, 2] == out[:, 4]) & (out[:, 3] == out[:, 5])
    print("%d queries, %.1f %% separated" % (n, 100 * sep.mean()))
    print("sequential walk vs minimum with DFS tie-break: distance differs in %d, ids differ in %d queries" % ((~same_d).sum(), (~same_id).sum()))
    print("separated queries with several triangle pairs EXACTLY at the minimum: %.1f %%; with a pair within 1e-12 but not equal: %.1f %%" % (
        100 * (out[sep, 9] > 1).mean(), 100 * (out[sep, 6] > out[sep, 9]).mean()))
    print("box tests per query: sequential %.0f, pruning relaxed by 1e-9: %.0f" % (out[:, 7].mean(), out[:, 8].mean()))
else:
    out = np.zeros((n, 6))
    L.walk_predict(*args, C.c_int(64), abi.ptr(out), threads)
    mind64, minb, ssz, nbv64, nbv, dfin = out.T
    rem, live = nbv - nbv64, ssz > 0
    rk = lambda v: np.argsort(np.argsort(v))  # noqa: E731
    for k, v in {"minimum after 64 steps": mind64, "smallest bound on the stack": minb, "their gap": mind64 - minb, "stack size": ssz, "final distance": dfin}.items():
        print("%-28s rank correlation with the box tests still to do: %.2f" % (k, np.corrcoef(rk(v[live]), rk(rem[live]))[0, 1]))
    print("box tests per walk: mean %.0f, p99 %.0f, max %.0f" % (nbv.mean(), np.percentile(nbv, 99), nbv.max()))
