#!/usr/bin/env python
"""Per-call time of the device-resident entry points for small batches (inputs resident, one stream, back-to-back
calls, kernel timing markers off): what a per-frame caller with a few hundred pairs pays."""
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
    for name, gen, f32 in (("cfg5_mixed fp64 collide", wl.cfg5_mixed, False), ("cfg3 fp32 distance", wl.cfg3_convex_convex, True),
                           ("cfg1 fp64 distance", wl.cfg1_sphere_sphere, False)):
        for n in (1, 64, 1024, 16384):
            b = gen(n=n)
            req = wl.make_request(b, abi)
            lib = wl.make_library(pkg, b)
            lib.set_kernel_timing(False)
            s1 = torch.from_numpy(b.s1.astype(np.int32)).to(dev)
            s2 = torch.from_numpy(b.s2.astype(np.int32)).to(dev)
            if f32:
                p1, p2 = torch.from_numpy(b.pose1_f32).to(dev), torch.from_numpy(b.pose2_f32).to(dev)
                out = torch.zeros(n * 11, dtype=torch.int32, device=dev)
                fn = lib.distance_device_f32 if b.kind == "distance" else lib.collide_device_f32
            else:
                p1, p2 = torch.from_numpy(b.tf1).to(dev), torch.from_numpy(b.tf2).to(dev)
                out = torch.zeros(n * 24, dtype=torch.int32, device=dev)
                fn = lib.distance_device if b.kind == "distance" else lib.collide_device
            st = torch.cuda.current_stream().cuda_stream
            for _ in range(20):
                fn(s1, s2, p1, p2, n, req, out, stream=st)
            torch.cuda.synchronize()
            reps = 200
            t0 = time.perf_counter()
            for _ in range(reps):
                fn(s1, s2, p1, p2, n, req, out, stream=st)
            t_issue = (time.perf_counter() - t0) / reps
            torch.cuda.synchronize()
            t_all = (time.perf_counter() - t0) / reps
            # one call, issue to completion
            lat = []
            for _ in range(20):
                t1 = time.perf_counter()
                fn(s1, s2, p1, p2, n, req, out, stream=st)
                torch.cuda.synchronize()
                lat.append(time.perf_counter() - t1)
            # the same call captured once into a HIP graph (the launch sequence of a batch is fixed for given buffers and
            # n; the workspace is allocated by the warm-up calls) and replayed
            graph_us = float("nan")
            try:
                side = torch.cuda.Stream()
                side.wait_stream(torch.cuda.current_stream())
                with torch.cuda.stream(side):
                    g = torch.cuda.CUDAGraph()
                    with torch.cuda.graph(g, stream=side):
                        fn(s1, s2, p1, p2, n, req, out, stream=torch.cuda.current_stream().cuda_stream)
                    for _ in range(5):
                        g.replay()
                    torch.cuda.synchronize()
                    t1 = time.perf_counter()
                    for _ in range(reps):
                        g.replay()
                    torch.cuda.synchronize()
                    graph_us = 1e6 * (time.perf_counter() - t1) / reps
                torch.cuda.current_stream().wait_stream(side)
            except Exception as e:  # capture not supported by this runtime
                graph_us = float("nan")
                print("   (graph capture failed: %s)" % str(e).splitlines()[0][:100])
            # the host-buffer entry point (what hpp::fcl::collide() through the shim calls): copies in, kernels, records out
            host_us = float("nan")
            if not f32 and n <= 1024:
                hfn = lib.distance if b.kind == "distance" else lib.collide
                for _ in range(5):
                    hfn(b.s1, b.s2, b.tf1, b.tf2, req)
                hl = []
                for _ in range(50):
                    t1 = time.perf_counter()
                    hfn(b.s1, b.s2, b.tf1, b.tf2, req)
                    hl.append(time.perf_counter() - t1)
                host_us = 1e6 * float(np.median(hl))
            print("%-26s n=%6d  back-to-back %7.1f us/call (host issue %6.1f us)   single call issue->done %7.1f us   "
                  "HIP-graph replay %7.1f us   host-buffer call %7.1f us" % (name, n, 1e6 * t_all, 1e6 * t_issue, 1e6 * float(np.median(lat)), graph_us, host_us))
            lib.close()


if __name__ == "__main__":
    main()
