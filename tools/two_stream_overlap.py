#!/usr/bin/env python
"""Does running two half-batches on two streams (two libraries = two workspaces) hide the kernel tails of one
full batch?  Throughput of 1 x n pairs on one stream vs 2 x n/2 on two streams, same pairs."""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from __graft_entry__ import load_pkg  # noqa: E402


def main():
    import torch
    pkg = load_pkg()
    abi, wl = pkg.abi, pkg.workloads
    dev = torch.device("cuda:0")
    for name, gen, f32, n in (("cfg3", wl.cfg3_convex_convex, True, 1_000_000), ("cfg2", wl.cfg2_box_capsule, False, 1_000_000),
                              ("cfg5", wl.cfg5_mixed, False, 1_250_000)):
        b = gen(n=n)
        req = wl.make_request(b, abi)
        libs = [wl.make_library(pkg, b) for _ in range(2)]
        for l in libs:
            l.set_kernel_timing(False)
        s1 = torch.from_numpy(b.s1.astype(np.int32)).to(dev)
        s2 = torch.from_numpy(b.s2.astype(np.int32)).to(dev)
        if f32:
            p1, p2, w, pw = torch.from_numpy(b.pose1_f32).to(dev), torch.from_numpy(b.pose2_f32).to(dev), 11, 7
            fns = [(l.distance_device_f32 if b.kind == "distance" else l.collide_device_f32) for l in libs]
        else:
            p1, p2, w, pw = torch.from_numpy(b.tf1).to(dev), torch.from_numpy(b.tf2).to(dev), 24, 12
            fns = [(l.distance_device if b.kind == "distance" else l.collide_device) for l in libs]
        out = torch.zeros(n * w, dtype=torch.int32, device=dev)
        streams = [torch.cuda.Stream(), torch.cuda.Stream()]
        h = n // 2

        def full():
            fns[0](s1, s2, p1, p2, n, req, out, stream=streams[0].cuda_stream)

        def halves():
            for k in range(2):
                lo = k * h
                fns[k](s1[lo:lo + h], s2[lo:lo + h], p1.view(-1, pw)[lo:lo + h], p2.view(-1, pw)[lo:lo + h], h, req,
                       out.view(-1, w)[lo:lo + h], stream=streams[k].cuda_stream)

        res = {}
        for label, f in (("1 stream x n", full), ("2 streams x n/2", halves)):
            for _ in range(3):
                f()
            torch.cuda.synchronize()
            reps = 20
            t0 = time.perf_counter()
            for _ in range(reps):
                f()
            torch.cuda.synchronize()
            res[label] = (time.perf_counter() - t0) / reps
        print("%s n=%d: %s" % (name, n, ", ".join("%s %.3f ms" % (k, 1e3 * v) for k, v in res.items())))
        for l in libs:
            l.close()


if __name__ == "__main__":
    main()
