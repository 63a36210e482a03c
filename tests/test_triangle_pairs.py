"""Top-level TriangleP pairs (collision_func_matrix.cpp:295-469, GJKSolver::shapeDistance's TriangleP overloads
narrowphase.h:320-348, triangle_sphere.cpp, triangle_triangle.cpp).

CPU: the oracle is pinned on the reference's Sphere x TriangleP known-answer test
(test/geometric_shapes.cpp:844-975: collide flag + contact normal) and cross-checked against an independent
route through the oracle (the triangle handed over as a 3-vertex ConvexBase: generic GJK/EPA without the
pre-transform / operand swap of the TriangleP overloads); the device header (host build) against the oracle.
GPU: k_triangle against the oracle."""
import numpy as np
import pytest

from compare import check_parity, check_properties


def _coll(oracle, L, a, b, tf1, tf2):
    return oracle.collide_batch(L.shapes_array(), L.vertices_array(), [a], [b], [tf1], [tf2], None)[0]


@pytest.fixture()
def frame(pkg):
    g = pkg.geometry
    rng = np.random.default_rng(5)
    q = rng.normal(size=4)
    tr = g.make_pose(quat=q / np.linalg.norm(q), T=rng.uniform(-10, 10, 3))
    return g, g.make_pose(), tr, g.pose_R(tr)


KAT = [  # (triangle vertices, translation of the triangle, expected normal): geometric_shapes.cpp:855-975
    ([[20, 0, 0], [-20, 0, 0], [0, 20, 0]], [0, 0, 0.001], [0, 0, 1]),
    ([[20, 0, 0], [-20, 0, 0], [0, 20, 0]], [0, 0, -0.001], [0, 0, -1]),
    ([[30, 0, 0], [9.9, -20, 0], [9.9, 20, 0]], [0, 0, 0.001], [9.9, 0, 0.001]),
    ([[30, 0, 0], [9.9, -20, 0], [9.9, 20, 0]], [0, 0, -0.001], [9.9, 0, -0.001]),
    ([[30, 0, 0], [-20, 0, 0], [0, 0, 20]], [0, 0.001, 0], [0, 1, 0]),
    ([[30, 0, 0], [-20, 0, 0], [0, 0, 20]], [0, -0.001, 0], [0, -1, 0]),
    ([[0, 30, 0], [0, -10, 0], [0, 0, 20]], [0.001, 0, 0], [1, 0, 0]),
    ([[0, 30, 0], [0, -10, 0], [0, 0, 20]], [-0.001, 0, 0], [-1, 0, 0]),
]


@pytest.mark.parametrize("verts,shift,normal", KAT)
def test_collide_spheretriangle_kat(oracle, pkg, frame, verts, shift, normal):
    g, I, tr, R = frame
    L = g.ShapeLibrary()
    s, t = L.add_sphere(10), L.add_triangle(*np.array(verts, dtype=float))
    n = np.array(normal, dtype=float)
    n /= np.linalg.norm(n)
    tf_tri = g.make_pose(T=shift)
    r = _coll(oracle, L, s, t, I, tf_tri)
    assert r["num_contacts"] == 1 and np.allclose(r["normal"], n, atol=1e-9)
    r = _coll(oracle, L, s, t, tr, g.compose(tr, tf_tri))
    assert r["num_contacts"] == 1 and np.allclose(r["normal"], R @ n, atol=1e-9)
    # operand swap (triangle_sphere.cpp:45-56): points exchanged, normal negated
    r2 = _coll(oracle, L, t, s, g.compose(tr, tf_tri), tr)
    assert r2["num_contacts"] == 1 and np.allclose(r2["normal"], -(R @ n), atol=1e-9)
    assert np.allclose(r2["p1"], r["p2"], atol=1e-9) and np.allclose(r2["p2"], r["p1"], atol=1e-9)


def _as_convex3(pkg, b):
    """The same batch with every TriangleP replaced by a ConvexBase of its 3 vertices."""
    shapes = b.shapes.copy()
    tri = shapes["type"] == pkg.abi.GEOM_TRIANGLE
    shapes["type"][tri] = pkg.abi.GEOM_CONVEX
    return shapes, tri


def test_oracle_triangle_overloads_vs_generic_route(oracle, pkg):
    """TriangleP x solid through the TriangleP overloads == 3-vertex hull x solid through plain GJK/EPA, to the
    solver tolerance (different frames and operand order: not bit-identical), with the witness on the triangle's
    side reported first or second according to the operand order."""
    abi, wl = pkg.abi, pkg.workloads
    b = wl.triangle_pairs(n=20000, seed=2)
    req = wl.make_request(b, abi)
    ref = oracle.collide_batch(b.shapes, b.verts, b.s1, b.s2, b.tf1, b.tf2, req, n_threads=4)
    shapes3, tri = _as_convex3(pkg, b)
    alt = oracle.collide_batch(shapes3, b.verts, b.s1, b.s2, b.tf1, b.tf2, req, n_threads=4)
    k1, k2 = b.shapes["type"][b.s1], b.shapes["type"][b.s2]
    # tri x tri has its own routine (no EPA, computePenetration) and sphere has a closed form: compare the GJK solids,
    # away from smooth shapes whose EPA stops on its tolerance
    poly = np.isin(k1, [abi.GEOM_BOX, abi.GEOM_CONVEX, abi.GEOM_TRIANGLE, abi.GEOM_CAPSULE]) & \
        np.isin(k2, [abi.GEOM_BOX, abi.GEOM_CONVEX, abi.GEOM_TRIANGLE, abi.GEOM_CAPSULE]) & \
        ~((k1 == abi.GEOM_TRIANGLE) & (k2 == abi.GEOM_TRIANGLE))
    assert poly.sum() > 4000
    dd = np.abs(ref["distance"][poly] - alt["distance"][poly])
    assert dd.max() < 5e-6, dd.max()
    assert np.array_equal(abi.status_contact(ref["status"][poly]), abi.status_contact(alt["status"][poly]) |
                          (np.abs(ref["distance"][poly]) < 1e-6) & abi.status_contact(ref["status"][poly]))
    sep_r = ref["p2"][poly] - ref["p1"][poly]
    sep_a = alt["p2"][poly] - alt["p1"][poly]
    assert np.abs(sep_r - sep_a).max() < 2e-3
    # sphere x triangle closed form vs GJK on (sphere, hull3)
    sph = ((k1 == abi.GEOM_SPHERE) | (k2 == abi.GEOM_SPHERE)) & ((k1 == abi.GEOM_TRIANGLE) | (k2 == abi.GEOM_TRIANGLE))
    sep = ref["distance"][sph] > 1e-3
    assert np.abs(ref["distance"][sph][sep] - alt["distance"][sph][sep]).max() < 5e-6


@pytest.mark.parametrize("margin", [0.0, 0.02])
def test_device_header_triangle_pairs(pkg, oracle, hostsim, margin):
    abi, wl = pkg.abi, pkg.workloads
    b = wl.triangle_pairs(n=20000, seed=3)
    req = wl.make_request(b, abi)
    req.security_margin = margin
    ref, gref = oracle.collide_batch(b.shapes, b.verts, b.s1, b.s2, b.tf1, b.tf2, req, n_threads=4, want_guess=True)
    got, ggot = hostsim.batch_f64(abi, b.shapes, b.verts, b.s1, b.s2, b.tf1, b.tf2, req, want_guess=True)
    assert np.array_equal(got["status"], ref["status"])
    assert np.array_equal(got["num_contacts"], ref["num_contacts"])
    fin = np.isfinite(ref["distance"]) & (np.abs(ref["distance"]) < 1e300)
    assert np.abs(got["distance"][fin] - ref["distance"][fin]).max() < 1e-12
    assert np.nanmax(np.abs(ggot["gjk_guess"] - gref["gjk_guess"])) < 1e-12


def test_distance_on_a_triangle_is_unsupported_as_in_the_reference(pkg, oracle, hostsim):
    """src/distance_func_matrix.cpp has no GEOM_TRIANGLE row or column: hpp::fcl::distance() throws for it."""
    abi, wl = pkg.abi, pkg.workloads
    b = wl.triangle_pairs(n=200, seed=5, kind="distance")
    req = wl.make_request(b, abi)
    with pytest.raises(Exception):
        oracle.distance_batch(b.shapes, b.verts, b.s1, b.s2, b.tf1, b.tf2, req)
    got = hostsim.batch_f64(abi, b.shapes, b.verts, b.s1, b.s2, b.tf1, b.tf2, req)
    assert abi.status_skipped(got["status"]).all()
    f = pkg.engine.dll().hfcl_pair_supported
    assert f(abi.GEOM_TRIANGLE, abi.GEOM_BOX, 0) == 1 and f(abi.GEOM_TRIANGLE, abi.GEOM_BOX, 1) == 0
    assert f(abi.GEOM_PLANE, abi.GEOM_TRIANGLE, 0) == 1 and f(abi.GEOM_PLANE, abi.GEOM_TRIANGLE, 1) == 0


@pytest.mark.gpu
def test_triangle_pairs_gpu(pkg, oracle):
    abi, wl = pkg.abi, pkg.workloads
    kind = "collide"
    b = wl.triangle_pairs(n=50000, seed=4)
    req = wl.make_request(b, abi)
    ref = oracle.collide_batch(b.shapes, b.verts, b.s1, b.s2, b.tf1, b.tf2, req, n_threads=8)
    lib = pkg.Library(b.lib, device=0)
    try:
        got = lib.collide(b.s1, b.s2, b.tf1, b.tf2, req)
        buckets = lib.last_bucket_counts()
        bd = wl.triangle_pairs(n=100, seed=4, kind="distance")
        with pytest.raises(pkg.EngineError) as e:  # distance(): no TriangleP in the reference's distance matrix
            lib.distance(bd.s1, bd.s2, bd.tf1, bd.tf2, wl.make_request(bd, abi))
        assert e.value.code == abi.ERR_UNSUPPORTED_PAIR
    finally:
        lib.close()
    assert buckets["tri"] == len(b) and buckets["unsupported"] == 0
    # cone / cylinder / ellipsoid: EPA stops on its tolerance, FMA contraction moves the depth by ~tolerance
    st = check_parity(abi, got, ref, dist_tol=4e-6, point_tol=2e-3, flag_band=1e-9, name="triangle_pairs-" + kind)
    assert st["p999_dd"] < 1e-6, st
    k1, k2 = b.shapes["type"][b.s1], b.shapes["type"][b.s2]
    keep = ~((k1 == abi.GEOM_TRIANGLE) & (k2 == abi.GEOM_TRIANGLE) & (ref["distance"] <= 0))
    check_properties(abi, got[keep], tol=1e-6, name="triangle_pairs-" + kind)
