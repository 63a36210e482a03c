#!/usr/bin/env python
"""Mesh x solid (BVHModel<OBBRSS> against one convex shape): queries/s per solid kind, collide() and
distance(), cfg4-size models (5 000 triangles), device-resident inputs, HIP-event timing.

  tools/mesh_solid_bench.py [--n 100000] [--kinds box,sphere,...] [--reps 3] [--seg 50]"""
import argparse
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from __graft_entry__ import load_pkg  # noqa: E402


def scene(pkg, kind, n, seg):
    return pkg.workloads.mesh_vs_solid(kind, n=n, seg=seg)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--n", type=int, default=100_000)
    ap.add_argument("--kinds", default="sphere,box,capsule,cylinder,ellipsoid,convex32")
    ap.add_argument("--reps", type=int, default=3)
    ap.add_argument("--seg", type=int, default=50)
    ap.add_argument("--no-distance", action="store_true")
    a = ap.parse_args()
    import torch
    pkg = load_pkg()
    abi, wl = pkg.abi, pkg.workloads
    dev = torch.device("cuda:0")
    print("# mesh x solid, %d queries per kind, %d-triangle models; env: %s" % (
        a.n, 2 * a.seg * a.seg, " ".join("%s=%s" % kv for kv in sorted(os.environ.items()) if kv[0].startswith("HFCL_"))))
    print("%-10s %-9s %10s %10s %9s   %s" % ("solid", "call", "ms", "M q/s", "contacts", "kernels (ms)"))
    for kind in a.kinds.split(","):
        b = scene(pkg, kind, a.n, a.seg)
        lib = wl.make_library(pkg, b)
        s1 = torch.from_numpy(b.s1.astype(np.int32)).to(dev)
        s2 = torch.from_numpy(b.s2.astype(np.int32)).to(dev)
        p1, p2 = torch.from_numpy(b.tf1).to(dev), torch.from_numpy(b.tf2).to(dev)
        out = torch.zeros(a.n * 24, dtype=torch.int32, device=dev)
        st = torch.cuda.current_stream().cuda_stream
        for call in ("collide",) + (() if a.no_distance else ("distance",)):
            req = abi.default_collision_request() if call == "collide" else abi.default_distance_request()
            fn = lib.collide_device if call == "collide" else lib.distance_device
            lib.set_kernel_timing(True)
            fn(s1, s2, p1, p2, a.n, req, out, stream=st)
            torch.cuda.synchronize()
            br = lib.last_kernel_breakdown()
            lib.set_kernel_timing(False)
            best = 1e30
            for _ in range(a.reps):
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
                fn(s1, s2, p1, p2, a.n, req, out, stream=st)
                e1.record()
                torch.cuda.synchronize()
                best = min(best, e0.elapsed_time(e1))
            rec = out.cpu().numpy().view(abi.RESULT_DTYPE)
            hit = float((rec["num_contacts"] > 0).mean()) if call == "collide" else float((rec["distance"] <= 0).mean())
            ks = "  ".join("%s %.2f" % (k, v) for k, v in sorted(br, key=lambda kv: -kv[1])[:4])
            print("%-10s %-9s %10.2f %10.3f %9.3f   %s" % (kind, call, best, a.n / best / 1e3, hit, ks), flush=True)
        lib.close()


if __name__ == "__main__":
    main()
