"""sha1 of the fp32 device path's records on cfg3 / cfg3u batches (compare two builds: HFCL_LIB_PATH).  usage (GPU box): tools/dbg/cfg3_checksum.py [n]"""
import hashlib, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
from __graft_entry__ import load_pkg
import torch
pkg = load_pkg(); wl, abi = pkg.workloads, pkg.abi
n = int(sys.argv[1]) if len(sys.argv) > 1 else 1_000_000
dev = torch.device("cuda:0")
for seed in (1, 2):
    b = wl.cfg3_convex_convex(n=n, seed=seed)
    d = [torch.from_numpy(x).to(dev) for x in (b.s1.astype(np.int32), b.s2.astype(np.int32), b.pose1_f32, b.pose2_f32)]
    lib = pkg.Library(b.lib)
    out = torch.zeros(len(b) * 11, dtype=torch.int32, device=dev)
    req = wl.make_request(b, abi)
    lib.distance_device_f32(*d, len(b), req, out); torch.cuda.synchronize()
    print("cfg3 seed %d: %s" % (seed, hashlib.sha1(out.cpu().numpy().tobytes()).hexdigest()))
    lib.close()
