#!/usr/bin/env python
"""Throughput of the host-buffer entry points (the drop-in boundary: H2D + kernels + D2H inside the call).

usage (GPU box): tools/host_boundary.py [--workload cfg2] [--pairs 2000000] [--reps 5]
Prints queries/s through hfcl_collide_batch / hfcl_distance_batch (12-double poses) and the *_qt forms (7-double poses),
the bytes per pair that cross the link and the resulting GB/s in + out, next to the device-resident rate."""
import argparse
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from __graft_entry__ import load_pkg  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--workload", default="cfg2")
    ap.add_argument("--pairs", type=int, default=2_000_000)
    ap.add_argument("--reps", type=int, default=5)
    ap.add_argument("--chunks", default="0")
    a = ap.parse_args()
    import torch
    pkg = load_pkg()
    abi, wl = pkg.abi, pkg.workloads
    mk = {"cfg2": wl.cfg2_box_capsule, "cfg3": wl.cfg3_convex_convex, "cfg5": wl.cfg5_mixed, "cfg1": wl.cfg1_sphere_sphere}[a.workload]
    b = mk(n=a.pairs)
    req = wl.make_request(b, abi)
    lib = pkg.Library(b.lib)
    n = len(b)
    s1, s2 = b.s1.astype(np.uint32), b.s2.astype(np.uint32)
    tf1, tf2 = np.ascontiguousarray(b.tf1), np.ascontiguousarray(b.tf2)
    q1, q2 = np.ascontiguousarray(b.pose1_qt), np.ascontiguousarray(b.pose2_qt)
    host = lib.distance if b.kind == "distance" else lib.collide
    host_qt = lib.distance_qt if b.kind == "distance" else lib.collide_qt
    # device-resident reference rate
    dev = torch.device("cuda:0")
    d = [torch.from_numpy(x).to(dev) for x in (s1.astype(np.int32), s2.astype(np.int32), tf1, tf2)]
    d_out = torch.zeros(n * 24, dtype=torch.int32, device=dev)
    fdev = lib.distance_device if b.kind == "distance" else lib.collide_device
    st = torch.cuda.current_stream().cuda_stream
    for _ in range(2):
        fdev(*d, n, req, d_out, stream=st)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(a.reps):
        fdev(*d, n, req, d_out, stream=st)
    torch.cuda.synchronize()
    t_dev = (time.perf_counter() - t0) / a.reps
    print("%s n=%d  device-resident: %.3f ms  %.1f M q/s" % (b.name, n, 1e3 * t_dev, n / t_dev / 1e6))
    import ctypes as C
    dll = pkg.engine.dll()
    cfn = {False: dll.hfcl_distance_batch if b.kind == "distance" else dll.hfcl_collide_batch,
           True: dll.hfcl_distance_batch_qt if b.kind == "distance" else dll.hfcl_collide_batch_qt}

    def pinned(x):  # the same array in page-locked memory (what a caller who pins its buffers passes)
        t = torch.from_numpy(np.ascontiguousarray(x).view(np.uint8).reshape(-1)).pin_memory()
        return t.numpy().view(x.dtype).reshape(x.shape), t

    out_pageable = np.zeros(n, dtype=abi.RESULT_DTYPE)
    out_pageable[:] = out_pageable  # touched
    out_pinned, _keep_out = pinned(out_pageable)
    arrays = {"pageable": (s1, s2, tf1, tf2, q1, q2, out_pageable)}
    pins = [pinned(x) for x in (s1, s2, tf1, tf2, q1, q2)]
    arrays["pinned"] = tuple(p[0] for p in pins) + (out_pinned,)
    for chunk in [int(c) for c in a.chunks.split(",")]:
        lib.set_host_chunk(chunk)
        for mem in ("pageable", "pinned"):
            A = arrays[mem]
            for qt in (False, True):
                p1, p2 = (A[4], A[5]) if qt else (A[2], A[3])
                bpp_in = 8 + (112 if qt else 192)

                def call():
                    rc = cfn[qt](lib._h, abi.ptr(A[0]), abi.ptr(A[1]), abi.ptr(p1), abi.ptr(p2), C.c_size_t(n), C.byref(req),
                                 abi.ptr(A[6]), None, None)
                    assert rc == 0, pkg.engine.last_error()
                call()
                ts = []
                for _ in range(a.reps):
                    t0 = time.perf_counter()
                    call()
                    ts.append(time.perf_counter() - t0)
                t = min(ts)
                print("  host %-8s %-15s chunk %-7s: %7.3f ms  %6.1f M q/s   in %d B/pair = %5.1f GB/s, out 96 B/pair = %5.1f GB/s  (median %.3f ms)" % (
                    mem, "7-double poses" if qt else "12-double poses", chunk or "auto", 1e3 * t, n / t / 1e6, bpp_in, n * bpp_in / t / 1e9,
                    n * 96 / t / 1e9, 1e3 * float(np.median(ts))))
    lib.close()


if __name__ == "__main__":
    main()
