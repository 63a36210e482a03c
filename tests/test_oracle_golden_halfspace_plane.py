"""Pins the oracle's Plane / Halfspace rows (details.h:347-428,509-691) on the reference's known-answer
tests test/geometric_shapes.cpp:1275-1658 (halfspace/plane x sphere/box) and :3214-3566 (plane-plane,
halfspace-halfspace, halfspace-plane): collision flag, contact position (= midpoint of the witness
points), penetration depth (= signed distance) and normal, each in the identity frame and under a
common rigid transform.  CPU only."""
import numpy as np
import pytest

from kat_solver import oracle  # noqa: F401 -- every test below runs on the oracle AND (-m gpu) on the HIP path


def _coll(oracle, L, a, b, tf1, tf2):
    return oracle.collide_batch(L.shapes_array(), L.vertices_array(), [a], [b], [tf1], [tf2], None)[0]


@pytest.fixture()
def frame(pkg):
    g = pkg.geometry
    rng = np.random.default_rng(17)
    q = rng.normal(size=4)
    tr = g.make_pose(quat=q / np.linalg.norm(q), T=rng.uniform(-10, 10, 3))
    return g, g.make_pose(), tr, g.pose_R(tr), np.asarray(tr)[9:]


def _check(oracle, L, s1, s2, tf1, tf2, expect, contact=None, depth=None, normal=None, opposite_ok=False, tol=1e-9):
    r = _coll(oracle, L, s1, s2, tf1, tf2)
    assert bool(r["num_contacts"]) == expect
    if not expect:
        return
    if contact is not None:
        assert np.allclose((r["p1"] + r["p2"]) / 2, contact, atol=max(tol, 1e-9)), ((r["p1"] + r["p2"]) / 2, contact)
    if depth is not None:
        assert abs(r["distance"] - depth) < max(tol, 1e-9), (r["distance"], depth)
    if normal is not None:
        ok = np.allclose(r["normal"], normal, atol=tol)
        if opposite_ok:
            ok = ok or np.allclose(r["normal"], -np.asarray(normal), atol=tol)
        assert ok, (r["normal"], normal)


def _both_frames(oracle, frame, L, s1, s2, cases, flat_second=True):
    """cases: (translation of the second object, expect, contact, depth, normal [, opposite_ok])."""
    g, I, tr, R, T = frame
    for c in cases:
        t2, expect, contact, depth, normal = c[:5]
        opp = c[5] if len(c) > 5 else False
        t1 = c[6] if len(c) > 6 else [0, 0, 0]
        a, b = g.make_pose(T=t1), g.make_pose(T=t2)
        _check(oracle, L, s1, s2, a, b, expect, contact, depth, normal, opp)
        wc = None if contact is None else R @ np.asarray(contact, dtype=float) + T
        wn = None if normal is None else R @ np.asarray(normal, dtype=float)
        _check(oracle, L, s1, s2, g.compose(tr, a), g.compose(tr, b), expect, wc, depth, wn, True if opp else False, tol=1e-8)


def test_collide_halfspacesphere(oracle, pkg, frame):  # :1275-1362
    L = pkg.geometry.ShapeLibrary()
    s, hs = L.add_sphere(10), L.add_halfspace([1, 0, 0], 0)
    n = [-1, 0, 0]
    _both_frames(oracle, frame, L, s, hs, [
        ([0, 0, 0], True, [-5, 0, 0], -10, n), ([5, 0, 0], True, [-2.5, 0, 0], -15, n),
        ([-5, 0, 0], True, [-7.5, 0, 0], -5, n), ([-10.1, 0, 0], False, None, None, None),
        ([10.1, 0, 0], True, [0.05, 0, 0], -20.1, n)])


def test_collide_planesphere(oracle, pkg, frame):  # :1364-1471
    L = pkg.geometry.ShapeLibrary()
    s, pl = L.add_sphere(10), L.add_plane([1, 0, 0], 0)
    eps = 1e-6
    _both_frames(oracle, frame, L, s, pl, [
        ([0, 0, 0], True, [(-10 + eps) / 2, 0, 0], -10 + eps, [-1, 0, 0], True, [eps, 0, 0]),
        ([0, 0, 0], True, [(10 - eps) / 2, 0, 0], -10 + eps, [1, 0, 0], True, [-eps, 0, 0]),
        ([5, 0, 0], True, [7.5, 0, 0], -5, [1, 0, 0]), ([-5, 0, 0], True, [-7.5, 0, 0], -5, [-1, 0, 0]),
        ([-10.1, 0, 0], False, None, None, None), ([10.1, 0, 0], False, None, None, None)])


def test_collide_halfspacebox(oracle, pkg, frame):  # :1473-1565
    L = pkg.geometry.ShapeLibrary()
    s, hs = L.add_box(5, 10, 20), L.add_halfspace([1, 0, 0], 0)
    n = [-1, 0, 0]
    _both_frames(oracle, frame, L, s, hs, [
        ([0, 0, 0], True, [-1.25, 0, 0], -2.5, n), ([1.25, 0, 0], True, [-0.625, 0, 0], -3.75, n),
        ([-1.25, 0, 0], True, [-1.875, 0, 0], -1.25, n), ([2.51, 0, 0], True, [0.005, 0, 0], -5.01, n),
        ([-2.51, 0, 0], False, None, None, None)])


def test_collide_planebox(oracle, pkg, frame):  # :1567-1658
    L = pkg.geometry.ShapeLibrary()
    s, pl = L.add_box(5, 10, 20), L.add_plane([1, 0, 0], 0)
    _both_frames(oracle, frame, L, s, pl, [
        ([0, 0, 0], True, [1.25, 0, 0], -2.5, [1, 0, 0], True),
        ([1.25, 0, 0], True, [(2.5 + 1.25) / 2, 0, 0], -1.25, [1, 0, 0]),
        ([-1.25, 0, 0], True, [(-2.5 - 1.25) / 2, 0, 0], -1.25, [-1, 0, 0]),
        ([2.51, 0, 0], False, None, None, None), ([-2.51, 0, 0], False, None, None, None)])


def _flat_cases(rng):
    n = rng.normal(size=3)
    n /= np.linalg.norm(n)
    return n


@pytest.mark.parametrize("kinds", ["plane-plane", "halfspace-halfspace", "halfspace-plane"])
def test_collide_flat_flat(oracle, pkg, frame, kinds):  # :3214-3566
    g, I, tr, R, T = frame
    rng = np.random.default_rng(23)
    k1, k2 = kinds.split("-")

    def mk(L, kind, n, d):
        return L.add_plane(n, d) if kind == "plane" else L.add_halfspace(n, d)

    def run(n1, d1, n2, d2, expect, contact=None, depth=None, normal=None, tol=1e-9):
        L = g.ShapeLibrary()
        a, b = mk(L, k1, n1, d1), mk(L, k2, n2, d2)
        _check(oracle, L, a, b, I, I, expect, contact, depth, normal, tol=tol)
        wn = None if normal is None else R @ np.asarray(normal, dtype=float)
        wc = None
        if contact is not None:  # the reference recomputes the contact of the transformed plane (:3240-3244)
            rn = R @ np.asarray(n1, dtype=float)
            wc = rn * (d1 + rn @ T)
        _check(oracle, L, a, b, tr, tr, expect, wc, depth, wn, tol=max(tol, 1e-8))

    n = _flat_cases(rng)
    off = 3.14
    if kinds == "plane-plane":
        run(n, off, n, off, True, contact=n * off, depth=0.0, normal=n)
        run(n, off, n, off + 1.19841, False)
        run(n, off, n, off - 1.19841, False)
    elif kinds == "halfspace-halfspace":
        run(n, off, n, off, True, normal=n)
        run(n, off, n, off + 1.19841, True, normal=n)
        off2 = off - 1.19841
        run(n, off, -n, -off2, True, depth=off2 - off, normal=n)
    else:
        run(n, off, n, off, True, depth=0.0, normal=n)
        run(n, off, n, off + 1.19841, False)
        off2 = off - 1.19841
        run(n, off, n, off2, True, depth=off2 - off, normal=n)
    # crossing flats: infinite penetration, normal = direction of the intersection line (not normalised)
    run([1, 0, 0], 3.14, [0, 0, 1], -2.13, True, normal=[0, -1, 0])
    run([1, 0, 0], 3.14, [1, 1, 1], -2.13, True, normal=[0, -0.5774, 0.5774], tol=1e-3)
    if kinds == "plane-plane":  # contact = origin of the intersection line (:3283-3284)
        L = g.ShapeLibrary()
        a, b = L.add_plane([1, 0, 0], 3.14), L.add_plane([0, 0, 1], -2.13)
        _check(oracle, L, a, b, I, I, True, contact=[3.14, 0, -2.13])


def test_operand_order_swaps_points_and_normal(oracle, pkg, frame):
    """src/distance/*_halfspace.cpp: (shape, flat) = (flat, shape) with p1/p2 swapped and the normal flipped."""
    g, I, tr, R, T = frame
    rng = np.random.default_rng(3)
    L = g.ShapeLibrary()
    solids = [L.add_box(1, 2, 3), L.add_sphere(0.7), L.add_capsule(0.4, 1.2), L.add_cone(0.5, 1.0), L.add_cylinder(0.5, 1.0),
              L.add_ellipsoid(0.3, 0.6, 0.9), L.add_convex(rng.normal(size=(20, 3))), L.add_box(1, 1, 1, swept_sphere_radius=0.2)]
    flats = [L.add_halfspace([0.2, -0.5, 1.0], 0.3), L.add_plane([1, 2, -0.5], -0.2), L.add_halfspace([0, 0, 1], 0.1, swept_sphere_radius=0.05)]
    S, V = L.shapes_array(), L.vertices_array()
    for s in solids:
        for f in flats:
            for _ in range(5):
                q = rng.normal(size=4)
                tf1 = g.make_pose(quat=q / np.linalg.norm(q), T=rng.uniform(-1, 1, 3))
                q = rng.normal(size=4)
                tf2 = g.make_pose(quat=q / np.linalg.norm(q), T=rng.uniform(-1, 1, 3))
                a = oracle.distance_batch(S, V, [s], [f], [tf1], [tf2], None)[0]
                b = oracle.distance_batch(S, V, [f], [s], [tf2], [tf1], None)[0]
                assert a["distance"] == b["distance"]
                assert np.array_equal(a["normal"], -b["normal"]) and np.array_equal(a["p1"], b["p2"])
                # the witness on the flat lies on its (inflated) boundary; the one on the solid realises the distance
                assert abs(np.linalg.norm(a["p2"] - a["p1"]) - abs(a["distance"])) < 1e-9
