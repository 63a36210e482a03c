"""What k_gjk_cvx costs before and after its loop: the kernel's duration on cfg3 with gjk_max_iterations = 1, 2, 4, 8, 128 (the records of the capped runs are
GJK failures; only the time is read)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
from __graft_entry__ import load_pkg
import torch
pkg = load_pkg(); wl, abi = pkg.workloads, pkg.abi
n = int(sys.argv[1]) if len(sys.argv) > 1 else 1_000_000
b = wl.cfg3_convex_convex(n=n, seed=1); dev = torch.device("cuda:0")
d = [torch.from_numpy(x).to(dev) for x in (b.s1.astype(np.int32), b.s2.astype(np.int32), b.pose1_f32, b.pose2_f32)]
lib = pkg.Library(b.lib)
out = torch.zeros(len(b) * 11, dtype=torch.int32, device=dev)
for mi in (1, 2, 3, 4, 6, 8, 12, 128):
    req = wl.make_request(b, abi, gjk_max_iterations=mi)
    ts = []
    for _ in range(4):
        lib.distance_device_f32(*d, len(b), req, out); torch.cuda.synchronize()
        ts.append(dict(lib.last_kernel_breakdown()).get("k_gjk_cvx<cc>"))
    rec = out.cpu().numpy().view(abi.RESULT_F32_DTYPE)
    print("gjk_max_iterations %3d: k_gjk_cvx<cc> %.3f ms (min of 3), mean iterations run %.2f, EPA queue %d" % (
        mi, min(ts[1:]), abi.status_gjk_iters(rec["status"]).mean(), lib.last_bucket_counts()["epa_queue"]))
lib.close()
