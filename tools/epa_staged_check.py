#!/usr/bin/env python
"""GPU: the staged convex x convex fast tier (k_epa_prepare / k_epa_loop / k_epa_records) against the one-kernel form
(k_epa_stream<.., CC>) on the same batch: records must be identical byte for byte (same arithmetic, other kernels).
usage: python tools/epa_staged_check.py [n] [seed]"""
import importlib
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
pkg = importlib.import_module("hpp-fcl_amd")
abi, wl = pkg.abi, pkg.workloads
n = int(sys.argv[1]) if len(sys.argv) > 1 else 300_000
seed = int(sys.argv[2]) if len(sys.argv) > 2 else 1
dev = torch.device("cuda", 0)
b = wl.cfg3_convex_convex(n=n, seed=seed)
req = wl.make_request(b, abi)
d_s1 = torch.from_numpy(b.s1.astype(np.int32)).to(dev)
d_s2 = torch.from_numpy(b.s2.astype(np.int32)).to(dev)
d_p1, d_p2 = torch.from_numpy(b.pose1_f32).to(dev), torch.from_numpy(b.pose2_f32).to(dev)
recs = {}
for staged in ("0", "1"):
    os.environ["HFCL_EPA_CC_STAGED"] = staged
    os.environ["HFCL_EPA_CC_STAGED_MIN"] = "0"
    lib = wl.make_library(pkg, b, device=0)
    out = torch.zeros(n * 11, dtype=torch.int32, device=dev)
    for rep in range(3):
        lib.distance_device_f32(d_s1, d_s2, d_p1, d_p2, n, req, out)
        torch.cuda.synchronize()
    recs[staged] = out.cpu().numpy().copy()
    print("staged=%s kernels=%s counts=%s" % (staged, [(k, round(v, 4)) for k, v in lib.last_kernel_breakdown()], lib.last_bucket_counts()))
    lib.close()
same = np.array_equal(recs["0"], recs["1"])
if not same:  # with k_epa_resume_cc the handed-over polytopes are continued by other code (parallel horizon): envelope, not identity
    A, B = recs["0"].reshape(n, 11), recs["1"].reshape(n, 11)
    d = np.abs(A[:, 0].view(np.float32) - B[:, 0].view(np.float32))
    bad = (A != B).any(axis=1)
    ei = (A[:, 10] >> 16) & 127
    print("differing records: %d, all with >= 17 EPA iterations in the one-kernel form: %s, max |dd| %.3g, statuses equal %d" % (
        bad.sum(), bool((ei[bad] >= 17).all()), d[bad].max(), int((A[bad, 10] == B[bad, 10]).sum())))
    same = bool((ei[bad] >= 17).all()) and d[bad].max() < 1e-5
a, c = recs["0"].reshape(n, 11), recs["1"].reshape(n, 11)
diff = np.nonzero((a != c).any(axis=1))[0]
print("records identical: %s (%d of %d differ)" % (same, len(diff), n))
for i in diff[:5]:
    print(i, a[i].view(np.float32)[:10], hex(a[i][10]), "|", c[i].view(np.float32)[:10], hex(c[i][10]))
sys.exit(0 if same else 1)
