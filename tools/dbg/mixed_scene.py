"""A batch that holds mesh pairs AND solid pairs (cfg5's mixed solids + cfg4's mesh x mesh queries in one library): the mesh walks beside the solids'
kernels (option mesh_beside) against one after the other."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
from __graft_entry__ import load_pkg
import torch
pkg = load_pkg(); wl, abi = pkg.workloads, pkg.abi
dev = torch.device("cuda:0")
b = wl.mesh_vs_shapes(n=int(sys.argv[1]) if len(sys.argv) > 1 else 200000, seed=4)
req = wl.make_request(b, abi)
d = [torch.from_numpy(x).to(dev) for x in (b.s1.astype(np.int32), b.s2.astype(np.int32), b.tf1, b.tf2)]
for beside in (1, 0, 1, 0):
    lib = wl.make_library(pkg, b, options={"mesh_beside": beside})
    out = torch.zeros(len(b) * 24, dtype=torch.int32, device=dev)
    for _ in range(3):
        lib.collide_device(*d, len(b), req, out)
    torch.cuda.synchronize()
    lib.set_kernel_timing(False)
    t0 = time.perf_counter()
    for _ in range(10):
        lib.collide_device(*d, len(b), req, out)
    torch.cuda.synchronize()
    print("mesh_beside %d: %.3f ms per batch of %d  buckets %s" % (beside, 1e2 * (time.perf_counter() - t0), len(b), {k: v for k, v in lib.last_bucket_counts().items() if v}))
    lib.close()
