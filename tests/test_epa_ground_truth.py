"""EPA on convex x convex pairs against an independent ground truth (tests/golden/epa_convex_convex.npz, made by
tools/make_epa_ground_truth.py): the exact penetration depth of two polytopes is the distance from the origin to the
closest facet of the convex hull of their pairwise vertex differences (qhull) -- no GJK / EPA involved.  The reference's own
tests hold no known-answer vector for this path; this pins the oracle's restatement of EPA::evaluate
(src/narrowphase/gjk.cpp:1156-1316) and, on the GPU, the HIP kernels, on the headline workload's shape class."""
import os

import numpy as np
import pytest

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "epa_convex_convex.npz")


def _batch(pkg):
    z = np.load(GOLD)
    g = pkg.geometry
    k = len(z["depth"])
    lib = g.ShapeLibrary()
    hulls = np.empty((2 * k, 32, 3))
    hulls[0::2], hulls[1::2] = z["hull1"], z["hull2"]
    lib.add_convex_many(hulls)
    s1 = 2 * np.arange(k, dtype=np.uint32)
    return z, lib, s1, s1 + 1


def _check(z, got, depth_tol, name, normal_tol=1e-4):
    depth, n_ref, gap = z["depth"], z["normal"], z["facet_gap"]
    d = got["distance"]
    assert np.all(d < 0), name + ": every golden pair penetrates"
    err = np.abs(-d - depth)
    bad = err > depth_tol * (1 + depth)
    assert not bad.any(), "%s: %d depths off, worst %.3g (depth %.3g)" % (name, bad.sum(), err.max(), depth[np.argmax(err)])
    # EPA never reports less than the true depth minus its tolerance: its face is a face of an inner polytope
    assert np.all(-d >= depth - depth_tol * (1 + depth))
    # where the closest facet is unique by a margin, the normal is that facet's (sign: hpp-fcl's normal points from
    # shape 1 to shape 2, the facet normal of {a - b} points the way shape 1 has to move out)
    uniq = gap > 1e-3
    cosang = np.einsum("ij,ij->i", got["normal"][uniq], n_ref[uniq])
    assert uniq.sum() > 400
    assert np.all(np.abs(np.abs(cosang) - 1) < normal_tol), name + ": normal off the closest facet's"
    assert np.all(cosang < 0) or np.all(cosang > 0), name + ": inconsistent normal orientation"
    # witness points: p2 - p1 = distance * normal (narrowphase.h:658-711)
    sep = got["p2"] - got["p1"]
    assert np.abs(sep - d[:, None] * got["normal"]).max() < 1e-6


@pytest.mark.parametrize("variant", ["DefaultGJK", "NesterovAcceleration"])
def test_oracle_epa_vs_qhull(pkg, oracle, variant):
    abi = pkg.abi
    z, lib, s1, s2 = _batch(pkg)
    req = abi.default_distance_request()
    req.q.gjk_variant = getattr(abi, variant)
    got = oracle.distance_batch(lib.shapes_array(), lib.vertices_array(), s1, s2, z["tf1"], z["tf2"], req)
    assert np.all(abi.status_epa(got["status"]) != 15), "EPA must have run on every pair"
    _check(z, got, 2e-6, "oracle-" + variant)


@pytest.mark.gpu
def test_gpu_epa_vs_qhull(pkg, torch_cuda):
    """fp64 kernels through the host boundary and the fp32 device path, both against the qhull depths."""
    torch = torch_cuda
    abi = pkg.abi
    z, lib_shapes, s1, s2 = _batch(pkg)
    req = abi.default_distance_request()
    req.q.gjk_variant = abi.NesterovAcceleration
    lib = pkg.Library(lib_shapes)
    got = lib.distance(s1, s2, z["tf1"], z["tf2"], req)
    _check(z, got, 2e-6, "gpu-fp64")
    # fp32: poses as quaternion + translation; envelope of the fp32 path: 1e-4 * (1 + |d|)
    dev = torch.device("cuda:0")
    k = len(s1)
    d = [torch.from_numpy(x).to(dev) for x in (s1.astype(np.int32), s2.astype(np.int32), z["pose1_qt"].astype(np.float32),
                                                z["pose2_qt"].astype(np.float32))]
    o = torch.zeros(k * 11, dtype=torch.int32, device=dev)
    lib.distance_device_f32(*d, k, req, o, stream=torch.cuda.current_stream().cuda_stream)
    torch.cuda.synchronize()
    got32 = o.cpu().numpy().view(abi.RESULT_F32_DTYPE)
    got32 = {f: got32[f].astype(np.float64) for f in ("distance", "normal", "p1", "p2")}
    depth = z["depth"]
    deep = depth > 1e-3  # (pairs shallower than the fp32 envelope may legitimately come out as touching)
    err = np.abs(-got32["distance"] - depth)
    assert np.all(err[deep] < 1e-4 * (1 + depth[deep])), err[deep].max()
    lib.close()
