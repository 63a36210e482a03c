#!/usr/bin/env python
"""Time of k_bvh_distance (BVHModel<OBBRSS> distance(), cfg4's 5 000-triangle meshes) per batch (GPU).
usage: python tools/bvh_distance_timing.py [n_queries ...]"""
import importlib
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
pkg = importlib.import_module("hpp-fcl_amd")
abi, wl = pkg.abi, pkg.workloads
sizes = [int(a) for a in sys.argv[1:]] or [100_000, 400_000]
dev = torch.device("cuda", 0)
for n in sizes:
    b = wl.cfg4_mesh_mesh(n=n, seed=1, half_width=2.2)
    lib = wl.make_library(pkg, b, device=0)
    req = abi.default_distance_request()
    d_s1 = torch.from_numpy(b.s1.astype(np.int32)).to(dev)
    d_s2 = torch.from_numpy(b.s2.astype(np.int32)).to(dev)
    d_p1, d_p2 = torch.from_numpy(b.tf1).to(dev), torch.from_numpy(b.tf2).to(dev)
    out = torch.zeros(n * 24, dtype=torch.int32, device=dev)
    ms = []
    for rep in range(4):
        lib.distance_device(d_s1, d_s2, d_p1, d_p2, n, req, out)
        torch.cuda.synchronize()
        if rep:
            ms.append(dict(lib.last_kernel_breakdown()).get("k_bvh_distance", float("nan")))
    print("bvh distance: %8d queries  k_bvh_distance %.3f ms  -> %.2f M queries/s" % (n, np.mean(ms), n / np.mean(ms) / 1e3))
    lib.close()
