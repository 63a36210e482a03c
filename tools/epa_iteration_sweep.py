#!/usr/bin/env python
"""Cost of the EPA kernels as a function of the request's iteration limit (GPU).  With the limit at 0 the
kernels still load the seed, the hulls, enclose the origin, build the first tetrahedron and write the
result record: the intercept is the per-polytope fixed cost, the slope the cost of the expansion loop.
usage: python tools/epa_iteration_sweep.py [cfg3|cfg5]"""
import importlib
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
pkg = importlib.import_module("hpp-fcl_amd")
abi, wl = pkg.abi, pkg.workloads
which = sys.argv[1] if len(sys.argv) > 1 else "cfg3"
n = 1_000_000
dev = torch.device("cuda", 0)
if which == "cfg3":
    b = wl.cfg3_convex_convex(n=n, seed=1)
else:
    b = wl.cfg5_mixed(n=n, seed=1)
lib = wl.make_library(pkg, b, device=0)
d_s1 = torch.from_numpy(b.s1.astype(np.int32)).to(dev)
d_s2 = torch.from_numpy(b.s2.astype(np.int32)).to(dev)
f32 = which == "cfg3"
if f32:
    d_p1, d_p2 = torch.from_numpy(b.pose1_f32).to(dev), torch.from_numpy(b.pose2_f32).to(dev)
    launch = lib.distance_device_f32 if b.kind == "distance" else lib.collide_device_f32
    words = 11
else:
    d_p1, d_p2 = torch.from_numpy(b.tf1).to(dev), torch.from_numpy(b.tf2).to(dev)
    launch = lib.distance_device if b.kind == "distance" else lib.collide_device
    words = 24
out = torch.zeros(n * words, dtype=torch.int32, device=dev)
print("%s: epa_max_iterations -> k_epa<fast> ms, k_epa<full> ms, queue lengths" % which)
for lim in (0, 1, 2, 3, 4, 6, 8, 12, 16, 24, 64):
    req = wl.make_request(b, abi, epa_max_iterations=lim)
    acc = {}
    for rep in range(6):
        launch(d_s1, d_s2, d_p1, d_p2, n, req, out)
        torch.cuda.synchronize()
        if rep >= 2:
            for name, ms in lib.last_kernel_breakdown():
                acc.setdefault(name, []).append(ms)
    c = lib.last_bucket_counts()
    print("%3d  %.3f  %.3f  queue=%d overflow=%d" % (lim, np.mean(acc["k_epa<fast>"]), np.mean(acc["k_epa<full>"]), c["epa_queue"], c["epa_overflow"]))
