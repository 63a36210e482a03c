import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
from __graft_entry__ import load_pkg
import oracle_binding as ob
import torch
pkg = load_pkg(); wl, abi = pkg.workloads, pkg.abi
n = 100000
b = wl.cfg4_mesh_mesh_distance(n=n, seed=1)
ML = pkg.bvh_builder.MeshLibrary(b.meshes)
ref, st = ob.bvh_distance_batch(ML, b.s1, b.s2, b.tf1, b.tf2, n_threads=os.cpu_count(), want_stats=True)
nbv = st[:, 0].astype(np.int64)
dev = torch.device("cuda:0")
req = wl.make_request(b, abi)
def run(order, label):
    lib = wl.make_library(pkg, b)
    s1 = torch.from_numpy(b.s1[order].astype(np.int32)).to(dev); s2 = torch.from_numpy(b.s2[order].astype(np.int32)).to(dev)
    p1 = torch.from_numpy(b.tf1[order]).to(dev); p2 = torch.from_numpy(b.tf2[order]).to(dev)
    out = torch.zeros(n * 24, dtype=torch.int32, device=dev)
    ts = []
    for _ in range(4):
        torch.cuda.synchronize(); t = time.perf_counter()
        lib.distance_device(s1, s2, p1, p2, n, req, out, stream=torch.cuda.current_stream().cuda_stream)
        torch.cuda.synchronize(); ts.append(time.perf_counter() - t)
    print("%-34s %.2f ms" % (label, 1e3 * min(ts[1:])))
    lib.close()
run(np.arange(n), "input order")
run(np.argsort(-nbv), "longest walks first (oracle)")
run(np.argsort(nbv), "shortest walks first")
run(np.argsort(-ref["distance"]), "largest distance first")
