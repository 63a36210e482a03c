"""What k_epa_loop costs per iteration level: its duration on cfg3 with epa_max_iterations = 1 ... 64 (capped runs end as EPA failures / fall-backs; only the
time is read) -- the same reading as tools/dbg/gjk_setup_cost.py for the GJK kernel."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
from __graft_entry__ import load_pkg
import torch
pkg = load_pkg(); wl, abi = pkg.workloads, pkg.abi
b = wl.cfg3_convex_convex(n=1_000_000, seed=1); dev = torch.device("cuda:0")
d = [torch.from_numpy(x).to(dev) for x in (b.s1.astype(np.int32), b.s2.astype(np.int32), b.pose1_f32, b.pose2_f32)]
lib = pkg.Library(b.lib)
out = torch.zeros(len(b) * 11, dtype=torch.int32, device=dev)
for mi in (1, 2, 3, 4, 6, 8, 12, 17, 24, 64):
    req = wl.make_request(b, abi, epa_max_iterations=mi)
    ts = []
    for _ in range(4):
        lib.distance_device_f32(*d, len(b), req, out); torch.cuda.synchronize()
        ts.append(dict(lib.last_kernel_breakdown()))
    rec = out.cpu().numpy().view(abi.RESULT_F32_DTYPE)
    ei = abi.status_epa_iters(rec["status"])[abi.status_epa(rec["status"]) != 15]
    k = {n: min(t.get(n, 0) for t in ts[1:]) for n in ("k_epa_prepare", "k_epa<fast>", "k_epa_records", "k_epa_resume_cc", "k_epa<full>")}
    print("epa_max_iterations %3d: %s  mean iterations run %.2f" % (mi, "  ".join("%s %.3f" % kv for kv in k.items()), ei.mean() if len(ei) else 0))
lib.close()
