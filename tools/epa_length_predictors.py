#!/usr/bin/env python
"""Which polytopes of the fp32 convex x convex EPA queue will run long (>= 17 iterations: the ones the loop kernel hands over)?  What
k_epa_prepare knows about a polytope before its loop -- the face distances of the first tetrahedron, its volume and extent, GJK's iteration count
-- against the iteration count of the loop, on the CPU (tests/hostsim: the device headers built for the host).  No GPU needed.
usage: python tools/epa_length_predictors.py [n]          (DESIGN.md "what comes next": a consumer beside the loop kernel would want the long
polytopes at the front of the queue)"""
import ctypes as C
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
from __graft_entry__ import load_pkg  # noqa: E402
import hostsim_binding as hs  # noqa: E402  (test infrastructure)

n = int(sys.argv[1]) if len(sys.argv) > 1 else 300000
pkg = load_pkg()
abi, wl = pkg.abi, pkg.workloads
L = hs.lib()
b = wl.cfg3_convex_convex(n=n, seed=1)
req = wl.make_request(b, abi)
shapes = np.ascontiguousarray(b.shapes)
verts = np.ascontiguousarray(b.verts, dtype=np.float64)
s1, s2 = np.ascontiguousarray(b.s1, dtype=np.uint32), np.ascontiguousarray(b.s2, dtype=np.uint32)
p1 = np.ascontiguousarray(b.pose1_f32, dtype=np.float32).reshape(-1, 7)
p2 = np.ascontiguousarray(b.pose2_f32, dtype=np.float32).reshape(-1, 7)
mm, fb = C.c_long(0), C.c_long(0)
feat = np.zeros((n, 6), dtype=np.float32)
L.sim_epa_prepare_selftest.restype = C.c_long
L.sim_epa_prepare_selftest(abi.ptr(shapes), C.c_size_t(len(shapes)), abi.ptr(verts), C.c_size_t(len(verts)), abi.ptr(s1), abi.ptr(s2), abi.ptr(p1),
                           abi.ptr(p2), C.c_size_t(n), C.byref(req), C.byref(mm), C.byref(fb), abi.ptr(feat))
out = hs.batch_f32(abi, b.shapes, b.verts, b.s1, b.s2, b.pose1_f32, b.pose2_f32, req)
ok = ~np.isnan(feat[:, 0])
ei = abi.status_epa_iters(out["status"]).astype(int)[ok]
F, depth = feat[ok], -out["distance"][ok]
long_ = ei >= 17
print("polytopes %d, %d of them (%.2f %%) with >= 17 iterations; histogram of the iteration counts: mean %.2f max %d" % (
    ok.sum(), long_.sum(), 100 * long_.mean(), ei.mean(), ei.max()))
cols = {"smallest face distance of the first tetrahedron": F[:, 0], "largest face distance": F[:, 1], "volume": F[:, 2], "GJK iterations": F[:, 3],
        "extent (longest edge)": F[:, 4], "ignored faces": F[:, 5], "final depth (not known before the loop)": depth}
print("%-52s %6s   share of the long polytopes in the top 10 %% / 25 %% / 50 %% by the predictor (larger first | smaller first)" % ("predictor", "corr"))
for name, c in cols.items():
    c = np.nan_to_num(c, nan=0, posinf=1e9, neginf=-1e9)
    res = []
    for sign in (1, -1):
        order = np.argsort(-sign * c)
        res.append(" / ".join("%.2f" % (long_[order[:int(q * len(c))]].sum() / long_.sum()) for q in (0.1, 0.25, 0.5)))
    print("%-52s %6.3f   %s | %s" % (name, np.corrcoef(c, ei)[0, 1], res[0], res[1]))

# What is left for a LAST continuation launch if the polytopes the predictor puts in front run (and are continued) first: the iteration counts
# of the long polytopes the predictor misses.  (round 6: the tail of k_epa_resume_cc is as long as its longest chain beyond the 17-iteration block)
c = np.nan_to_num(F[:, 1], nan=0, posinf=1e9, neginf=-1e9)  # largest face distance of the first tetrahedron
order = np.argsort(-c)
print("\nlongest chain beyond the block (iterations - 17) among the polytopes BEHIND the top q by 'largest face distance':")
for qf in (0.0, 0.1, 0.25, 0.5):
    rest = order[int(qf * len(c)):]
    e = ei[rest]
    over = np.sort(e[e >= 17] - 17)[::-1]
    print("  q = %.2f: %6d handed over, longest %2d, 99th percentile of those %2d, 90th %2d" % (
        qf, len(over), over[0] if len(over) else 0, over[int(0.01 * len(over))] if len(over) else 0, over[int(0.1 * len(over))] if len(over) else 0))

# ... and with the rule k_epa_prepare can apply without a threshold: a polytope goes in front if its predictor ranks among the largest 16 / 24 / 32 of the
# 64 polytopes of its wave (consecutive items of the queue)
print("\nwave-local rule (rank among the 64 consecutive polytopes of a wave):")
m = (len(c) // 64) * 64
cw, ew = c[:m].reshape(-1, 64), ei[:m].reshape(-1, 64)
rank = (cw[:, :, None] < cw[:, None, :]).sum(axis=2)  # how many of the wave's polytopes have a larger predictor
for top in (8, 16, 24, 32):
    front = rank < top
    e = ew[~front]
    over = np.sort(e[e >= 17] - 17)[::-1]
    caught = (ew[front] >= 17).sum() / max((ew >= 17).sum(), 1)
    print("  front = top %2d of 64 (%.0f %% of the polytopes, %.0f %% of their iterations): %.2f of the long ones in front; behind: %5d handed over, longest %2d, 99th percentile %2d" % (
        top, 100 * front.mean(), 100 * ew[front].sum() / ew.sum(), caught, len(over), over[0] if len(over) else 0, over[int(0.01 * len(over))] if len(over) else 0))
