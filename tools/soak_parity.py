#!/usr/bin/env python
"""Large randomized parity sweep on the GPU box: the HIP path (C ABI, host entry points) against the fp64 oracle on
millions of pairs per workload, several seeds and request variants.  Prints one line per run with the mismatch
statistics of tests/compare.py:check_parity (contact flags / statuses outside the decision band, distances, witness
separation vectors); exits non-zero on any violation.  The oracle runs on the host cores (test infrastructure)."""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
from __graft_entry__ import load_pkg  # noqa: E402


def main():
    import oracle_binding as ob
    from compare import check_parity, check_properties
    pkg = load_pkg()
    abi, wl = pkg.abi, pkg.workloads
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 1_000_000
    threads = min(os.cpu_count() or 8, 128)
    runs = []
    for seed in (11, 12):
        runs += [("cfg2_box_capsule", {}, seed), ("cfg3_convex_convex", {}, seed), ("cfg5_mixed", {}, seed),
                 ("all_primitives", {}, seed)]
    runs += [("cfg5_mixed", {"security_margin": 0.05, "distance_upper_bound": 0.2}, 13),
             ("cfg5_mixed", {"security_margin": -0.03}, 14), ("cfg5_mixed", {"enable_contact": 0}, 15),
             ("all_primitives", {"kind": "distance"}, 16), ("cfg3_convex_convex", {"gjk_variant": abi.PolyakAcceleration}, 17),
             ("cfg3_convex_convex", {"gjk_variant": abi.DefaultGJK, "gjk_convergence_criterion": abi.Hybrid}, 18),
             ("flat_pairs", {}, 19), ("triangle_pairs", {}, 20), ("large_convex", {}, 21),
             ("large_convex", {"support": "climb"}, 22), ("cfg5_mixed", {"entry": "qt"}, 23)]
    total, t_all = 0, time.time()
    for name, over, seed in runs:
        kw = {"kind": over.pop("kind")} if "kind" in over else {}
        climb = over.pop("support", None) == "climb"  # hulls climb their registered adjacency (oracle: getShapeSupportLog)
        qt = over.pop("entry", None) == "qt"          # compact host poses (unit quaternion + translation)
        label = dict(over, **kw, **({"support": "climb"} if climb else {}), **({"entry": "qt"} if qt else {}))
        nn = n if name not in ("flat_pairs", "triangle_pairs", "large_convex") else max(n // 5, 1000)
        b = getattr(wl, name)(n=nn, seed=seed, **kw)
        req = wl.make_request(b, abi, **over)
        t0 = time.time()
        fn_o = ob.distance_batch if b.kind == "distance" else ob.collide_batch
        if climb:
            ob.register_hull_neighbors(b.shapes, b.verts)
        try:
            ref = fn_o(b.shapes, b.verts, b.s1, b.s2, b.tf1, b.tf2, req, n_threads=threads)
        finally:
            if climb:
                ob.lib().orc_clear_neighbors()
        t_cpu = time.time() - t0
        if climb:
            os.environ["HFCL_CLIMB_MIN"] = "33"
        lib = pkg.Library(b.lib, device=0)
        os.environ.pop("HFCL_CLIMB_MIN", None)
        if climb:
            wl.register_adjacency(lib, b.shapes, b.verts)
        t0 = time.time()
        if qt:
            got = (lib.distance_qt if b.kind == "distance" else lib.collide_qt)(b.s1, b.s2, b.pose1_qt, b.pose2_qt, req)
        else:
            got = (lib.distance if b.kind == "distance" else lib.collide)(b.s1, b.s2, b.tf1, b.tf2, req)
        t_gpu = time.time() - t0
        lib.close()
        smooth = name in ("all_primitives", "flat_pairs", "triangle_pairs", "large_convex")
        st = check_parity(abi, got, ref, dist_tol=4e-6 if smooth else 1e-6, point_tol=2e-3 if smooth else 1e-5, flag_band=1e-9,
                          name=name, allow_bad_frac=1e-5 if smooth else 2e-6)  # smooth shapes: EPA stops on its tolerance, a few per
        # million end on another of two near-equidistant faces (normal off by ~sqrt(tolerance))
        total += len(b)
        print("%-22s %-58s n=%8d contacts %.3f  flag/gjk/epa/dist/sep mismatches %d/%d/%d/%d/%d  max|dd| %.2e p99.9 %.1e  "
              "(oracle %d thr %.1fs, engine %.2fs incl. copies)" %
              (name, str(label or ""), len(b), st["contact_frac"], st["flag_mismatch"], st["gjk_status_mismatch"],
               st["epa_status_mismatch"], st["dist_bad"], st["sep_bad"], st["max_dd"], st["p999_dd"], threads, t_cpu, t_gpu), flush=True)
    # ---- meshes (cfg4): collide (first contact in DFS order) and distance, 5 000-triangle models
    bb = pkg.bvh_builder
    nm = max(n // 10, 1000)
    # (the third leg is small enough for the automatic task split of batches that do not fill the chip's lanes)
    for seed, hw, nq in ((31, 1.25, nm), (32, 1.0, nm), (33, 1.1, min(nm, 100000))):
        b = wl.cfg4_mesh_mesh(n=nq, seed=seed, half_width=hw)
        ML = bb.MeshLibrary(b.meshes)
        req = wl.make_request(b, abi)
        lib = wl.make_library(pkg, b)
        got = lib.collide(b.s1, b.s2, b.tf1, b.tf2, req)
        gd = lib.distance(b.s1, b.s2, b.tf1, b.tf2)
        lib.close()
        ref = ob.bvh_collide_batch(ML, b.s1, b.s2, b.tf1, b.tf2, req, n_threads=threads)
        rd = ob.bvh_distance_batch(ML, b.s1, b.s2, b.tf1, b.tf2, n_threads=threads)
        assert not np.any((got["status"] >> 30) & 1) and not np.any((gd["status"] >> 30) & 1), "traversal stack overflow"
        near = np.abs(ref["distance"]) < 1e-9
        same = got["num_contacts"] == ref["num_contacts"]
        ok = same & ~near
        ids = (got["b1"][ok] == ref["b1"][ok]) & (got["b2"][ok] == ref["b2"][ok])
        fin = ok & (np.abs(ref["distance"]) < 1e300)
        dd = np.abs(got["distance"][fin] - ref["distance"][fin]).max()
        ddist = np.abs(gd["distance"] - rd["distance"]).max()
        assert np.all(same | near) and ids.all() and dd < 1e-6 and ddist < 1e-9, (same.mean(), ids.mean(), dd, ddist)
        total += 2 * len(b)
        print("cfg4_mesh_mesh seed %d      n=%8d  collide: contact counts / first-contact ids identical (%d near-zero skipped), "
              "max|dd| %.1e, contacts %.3f;  distance: max|dd| %.1e" %
              (seed, len(b), int(near.sum()), dd, float((ref["num_contacts"] > 0).mean()), ddist), flush=True)
    # ---- meshes against solids (cfg4s: six solid kinds mixed, first-contact collide; the waves' long walks cut into chunks, DESIGN.md
    # section 3 item 6e) and their distance(): contact counts / first-contact triangles identical, depths and distances to 4e-6
    for seed in (51, 52, 53):
        b = wl.mesh_vs_solid("mixed", n=nm, seed=seed)
        ML = bb.MeshLibrary(b.meshes)
        req = abi.default_collision_request()
        lib = wl.make_library(pkg, b)
        got = lib.collide(b.s1, b.s2, b.tf1, b.tf2, req)
        lib.close()
        ref, _ = ob.mixed_collide_batch(b.shapes, b.verts, ML, b.s1, b.s2, b.tf1, b.tf2, req, max_contacts=10 ** 5, n_threads=threads)
        assert not np.any((got["status"] >> 30) & 1), "traversal stack overflow"
        near = np.abs(ref["distance"]) < 1e-9
        same = got["num_contacts"] == ref["num_contacts"]
        ok = same & ~near
        ids = (got["b1"][ok] == ref["b1"][ok]) & (got["b2"][ok] == ref["b2"][ok])
        hit = ok & (ref["num_contacts"] > 0)
        dd = np.abs(got["distance"][hit] - ref["distance"][hit]).max()
        assert np.all(same | near) and ids.all() and dd < 4e-6, (same.mean(), ids.mean(), dd)
        total += len(b)
        print("mesh_x_solid (mixed) seed %d n=%8d  collide: contact counts / first-contact triangles identical (%d near-zero skipped), "
              "max|dd| of the depths %.1e, contacts %.3f" % (seed, len(b), int(near.sum()), dd, float((ref["num_contacts"] > 0).mean())), flush=True)
    # ---- fp32 device-resident path (cfg3, the bench configuration): envelope of DESIGN.md "fp32 path"
    import torch
    dev = torch.device("cuda:0")
    for seed in (41, 42):
        b = wl.cfg3_convex_convex(n=n, seed=seed)
        req = wl.make_request(b, abi)
        tf1, tf2 = b.tf_from_f32()
        ref = ob.distance_batch(b.shapes, b.verts, b.s1, b.s2, tf1, tf2, req, n_threads=threads)
        lib = pkg.Library(b.lib)
        d = [torch.from_numpy(x).to(dev) for x in (b.s1.astype(np.int32), b.s2.astype(np.int32), b.pose1_f32, b.pose2_f32)]
        o = torch.zeros(len(b) * 11, dtype=torch.int32, device=dev)
        lib.distance_device_f32(*d, len(b), req, o, stream=torch.cuda.current_stream().cuda_stream)
        torch.cuda.synchronize()
        got = o.cpu().numpy().view(abi.RESULT_F32_DTYPE)
        lib.close()
        st = check_parity(abi, got, ref, dist_tol=1e-4, point_tol=5e-4, flag_band=1e-4, name="cfg3-f32", fp32=True,
                          allow_bad_frac=2e-5)
        total += len(b)
        print("cfg3 fp32 device path seed %d n=%8d contacts %.3f  flag/dist/sep outside the fp32 envelope %d/%d/%d  max|dd| %.2e" %
              (seed, len(b), st["contact_frac"], st["flag_mismatch"], st["dist_bad"], st["sep_bad"], st["max_dd"]), flush=True)
    print("soak: %d pairs compared in %.0f s" % (total, time.time() - t_all))


if __name__ == "__main__":
    main()
