"""Host BVHModel<OBBRSS> construction: the native builder behind `hfcl_bvh_build`
against the oracle restatement of src/BVH/BVH_model.cpp:858-960 + BV_fitter/BV_splitter/BVH_utility,
plus the structural invariants any hpp-fcl tree satisfies.  CPU only (the builder is host code)."""
import os

import numpy as np
import pytest

import oracle_binding as ob


def _soup(rng, nt, spread=1.0):
    c = rng.uniform(-spread, spread, size=(nt, 1, 3))
    v = (c + rng.normal(scale=0.05, size=(nt, 3, 3))).reshape(-1, 3)
    return v, np.arange(3 * nt, dtype=np.uint32).reshape(-1, 3)


def _cases(bb):
    rng = np.random.default_rng(7)
    yield "bumpy", bb.bumpy_sphere(20, 20)
    yield "uv_sphere", bb.uv_sphere(16, 16, 2.0)
    yield "soup", _soup(rng, 500)
    yield "one_triangle", (np.array([[0, 0, 0], [1, 0, 0], [0, 1, 0.]]), np.array([[0, 1, 2]], dtype=np.uint32))
    yield "two_triangles", (np.array([[0, 0, 0], [1, 0, 0], [0, 1, 0.], [1, 1, 0]]), np.array([[0, 1, 2], [1, 3, 2]], dtype=np.uint32))
    # degenerate: all triangles identical (mean split cannot separate -> median fallback, BVH_model.cpp:950)
    yield "duplicates", (np.array([[0, 0, 0], [1, 0, 0], [0, 1, 0.]]), np.tile(np.array([[0, 1, 2]], dtype=np.uint32), (9, 1)))
    # coplanar grid (zero eigenvalue, zero RSS radius)
    g = np.stack(np.meshgrid(np.arange(6.), np.arange(6.), indexing="ij"), -1).reshape(-1, 2)
    gv = np.concatenate([g, np.zeros((36, 1))], axis=1)
    gt = [(i * 6 + j, i * 6 + j + 1, (i + 1) * 6 + j) for i in range(5) for j in range(5)]
    yield "coplanar", (gv, np.array(gt, dtype=np.uint32))


def test_native_builder_matches_oracle_bit_for_bit(pkg):
    bb = pkg.bvh_builder
    for name, (v, t) in _cases(bb):
        nodes, prim = bb.build_obbrss(v, t, n_threads=1)
        onodes, oprim = ob.bvh_build(v, t)
        assert np.array_equal(prim, oprim), name
        assert nodes.tobytes() == onodes.tobytes(), name


def test_threaded_build_is_identical(pkg):
    bb = pkg.bvh_builder
    v, t = bb.bumpy_sphere(110, 110)  # 24200 triangles: above the threading threshold
    n1, p1 = bb.build_obbrss(v, t, n_threads=1)
    n8, p8 = bb.build_obbrss(v, t, n_threads=8)
    assert np.array_equal(p1, p8) and n1.tobytes() == n8.tobytes()
    on, op = ob.bvh_build(v, t)
    assert np.array_equal(p1, op) and n1.tobytes() == on.tobytes()


def test_tree_invariants(pkg):
    bb = pkg.bvh_builder
    for name, (v, t) in _cases(bb):
        nodes, prim = bb.build_obbrss(v, t)
        T = len(t)
        assert len(nodes) == 2 * T - 1
        assert sorted(prim.tolist()) == list(range(T)), name
        leaves = nodes["first_child"] < 0
        assert leaves.sum() == T
        assert sorted((-(nodes["first_child"][leaves] + 1)).tolist()) == list(range(T)), name
        seen = np.zeros(len(nodes), dtype=int)
        seen[0] = 1
        for i, nd in enumerate(nodes):
            fp, npim = int(nd["first_primitive"]), int(nd["num_primitives"])
            if nd["first_child"] > 0:
                l, r = nodes[nd["first_child"]], nodes[nd["first_child"] + 1]
                seen[nd["first_child"]] += 1
                seen[nd["first_child"] + 1] += 1
                assert l["first_primitive"] == fp and l["num_primitives"] + r["num_primitives"] == npim
                assert r["first_primitive"] == fp + l["num_primitives"]
            else:
                assert npim == 1 and prim[fp] == -(nd["first_child"] + 1)
            # BVs enclose the corners of their triangles (BV_fitter.cpp:501-531)
            P = v[t[prim[fp:fp + npim]].reshape(-1)]
            A = nd["obb_axes"].reshape(3, 3).T  # columns = axes
            q = (P - nd["obb_To"]) @ A
            assert (np.abs(q) <= nd["obb_extent"] + 1e-9).all(), (name, i)
            assert abs(np.linalg.det(A) - 1) < 1e-9 and np.allclose(A.T @ A, np.eye(3), atol=1e-9), (name, i)
            # RSS: distance of every corner to the rectangle <= radius
            q = (P - nd["rss_Tr"]) @ A
            cx = np.clip(q[:, 0], 0, nd["rss_length"][0])
            cy = np.clip(q[:, 1], 0, nd["rss_length"][1])
            d = np.sqrt((q[:, 0] - cx) ** 2 + (q[:, 1] - cy) ** 2 + q[:, 2] ** 2)
            assert (d <= nd["rss_radius"] + 1e-9).all(), (name, i, d.max(), nd["rss_radius"])
        assert (seen == 1).all(), name


def test_load_obj_reader_quirks(pkg, tmp_path):
    bb = pkg.bvh_builder
    p = tmp_path / "a.obj"
    p.write_text("# c\nv 0 0 0\nv 1 0 0\nv 1 1 0\nv 0 1 0\nf 1 2 3 4\n")
    v, t = bb.load_obj(str(p))
    assert v.shape == (4, 3)
    # no vn/vt: both fan triangles repeat the first three indices (test/utility.cpp:139-143)
    assert t.tolist() == [[0, 1, 2], [0, 1, 2]]
    p.write_text("v 0 0 0\nv 1 0 0\nv 1 1 0\nv 0 1 0\nvn 0 0 1\nf 1//1 2//1 3//1 4//1\n")
    v, t = bb.load_obj(str(p))
    assert t.tolist() == [[0, 1, 2], [0, 2, 3]]


def test_invalid_arguments(pkg):
    with pytest.raises(pkg.engine.EngineError):
        pkg.engine.bvh_build(np.zeros((3, 3)), np.array([[0, 1, 5]], dtype=np.uint32))
