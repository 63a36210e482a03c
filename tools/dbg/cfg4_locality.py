"""Does the footprint of the models' node records bound the walk?  cfg4 (100k queries, fp64) with 1, 2, 4, 8 distinct 5 000-triangle models (1.28 MB of
OBB node records each; an XCD's L2 is 4 MB): step time and the walk kernels' durations."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
from __graft_entry__ import load_pkg
import torch
pkg = load_pkg(); wl, abi = pkg.workloads, pkg.abi
dev = torch.device("cuda:0")
def run(b, tag):
    req = wl.make_request(b, abi)
    lib = wl.make_library(pkg, b)
    d = [torch.from_numpy(x).to(dev) for x in (b.s1.astype(np.int32), b.s2.astype(np.int32), b.tf1, b.tf2)]
    out = torch.zeros(len(b) * 24, dtype=torch.int32, device=dev)
    for _ in range(3):
        lib.collide_device(*d, len(b), req, out)
    torch.cuda.synchronize()
    kb = dict(lib.last_kernel_breakdown())
    lib.set_kernel_timing(False)
    t0 = time.perf_counter()
    for _ in range(10):
        lib.collide_device(*d, len(b), req, out)
    torch.cuda.synchronize()
    ms = 1e2 * (time.perf_counter() - t0)
    rec = out.cpu().numpy().view(abi.RESULT_DTYPE)
    print("%-28s step %.3f ms  k_bvh_collide (timer: walk + leaves + resolve + continuation) %.3f  contacts %.3f" % (tag, ms, kb.get("k_bvh_collide", 0), (rec["num_contacts"] > 0).mean()))
    lib.close()
for nv in (1, 2, 4, 8):
    run(wl.cfg4_mesh_mesh(n=100_000, seed=1, n_variants=nv), "%d models" % nv)
