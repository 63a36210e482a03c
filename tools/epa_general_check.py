#!/usr/bin/env python
"""GPU: the general three-stage EPA tier (k_epa_prepare_general / k_epa_loop_general / k_epa_records_general) against the one-kernel forms on
the same fp64 batch: records and cached guesses must be identical byte for byte; kernel times and queue populations of both.
usage: python tools/epa_general_check.py [cfg5|cfg2|all_primitives] [n] [seed]"""
import importlib
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
pkg = importlib.import_module("hpp-fcl_amd")
abi, wl = pkg.abi, pkg.workloads
which = sys.argv[1] if len(sys.argv) > 1 else "cfg5"
n = int(sys.argv[2]) if len(sys.argv) > 2 else 600_000
seed = int(sys.argv[3]) if len(sys.argv) > 3 else 1
dev = torch.device("cuda", 0)
b = {"cfg5": wl.cfg5_mixed, "cfg2": wl.cfg2_box_capsule, "all_primitives": wl.all_primitives}[which](n=n, seed=seed)
req = wl.make_request(b, abi)
d_s1 = torch.from_numpy(b.s1.astype(np.int32)).to(dev)
d_s2 = torch.from_numpy(b.s2.astype(np.int32)).to(dev)
d_p1, d_p2 = torch.from_numpy(b.tf1).to(dev), torch.from_numpy(b.tf2).to(dev)
recs = {}
os.environ.setdefault("HFCL_SPLIT", "1")
for staged in ("0", "1"):
    os.environ["HFCL_EPA_GENERAL_STAGED"] = staged
    os.environ["HFCL_EPA_GENERAL_STAGED_MIN"] = "0"
    lib = wl.make_library(pkg, b, device=0)
    out = torch.zeros(n * 24, dtype=torch.int32, device=dev)
    gout = torch.zeros(n * 8, dtype=torch.int32, device=dev)
    launch = lib.distance_device if b.kind == "distance" else lib.collide_device
    acc = {}
    for rep in range(5):
        launch(d_s1, d_s2, d_p1, d_p2, n, req, out, None, gout)
        torch.cuda.synchronize()
        if rep >= 2:
            for k, v in lib.last_kernel_breakdown():
                acc.setdefault(k, []).append(v)
    recs[staged] = (out.cpu().numpy().copy(), gout.cpu().numpy().copy())
    lib.set_kernel_timing(False)
    import time
    for rep in range(3):
        launch(d_s1, d_s2, d_p1, d_p2, n, req, out, None, gout)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for rep in range(20):
        launch(d_s1, d_s2, d_p1, d_p2, n, req, out, None, gout)
    torch.cuda.synchronize()
    print("staged=%s wall %.4f ms per step (20 steps, kernel timing off, HFCL_SPLIT=%s)" % (staged, 1e3 * (time.perf_counter() - t0) / 20, os.environ.get("HFCL_SPLIT")))
    print("staged=%s kernels=%s sum=%.3f counts=%s" % (staged, [(k, round(float(np.mean(v)), 4)) for k, v in acc.items() if np.mean(v) > 0.004],
                                                      sum(float(np.mean(v)) for v in acc.values()), {k: v for k, v in lib.last_bucket_counts().items() if v}))
    lib.close()
same = np.array_equal(recs["0"][0], recs["1"][0]) and np.array_equal(recs["0"][1], recs["1"][1])
a, c = recs["0"][0].reshape(n, 24), recs["1"][0].reshape(n, 24)
diff = np.nonzero((a != c).any(axis=1))[0]
print("records + guesses identical: %s (%d of %d records differ)" % (same, len(diff), n))
sys.exit(0 if same else 1)
