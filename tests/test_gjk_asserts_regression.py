"""test/gjk_asserts.cpp:19-91 restated: two BVHModel<OBBRSS> UV spheres (radius 1 and 2, 32 x 32, built with the
float trigonometry of the reference's CreateSphereMesh, degenerate pole triangles included) under the 8 (i, j)
rotations x 6 unit translations that used to trip asserts in GJK / EPA (gjk.cpp:331, :1263).  The reference only asks
for "does not throw"; here: no stack overflow, no unsupported pair, and the host build of the device headers as
well as the kernels agree with the oracle."""
import numpy as np
import pytest

CASES = [(5, 48), (64, 151), (98, 47), (355, 48), (86, 52), (89, 17), (89, 58), (89, 145)]
DIRS = [(0, 0, 1), (0, 0, -1), (0, 1, 0), (0, -1, 0), (1, 0, 0), (-1, 0, 0)]


def _sphere_mesh(radius, polar=32, azimuth=32):
    f = np.float32
    pstep = f(np.pi) / f(polar - 1)
    astep = f(2.0) * f(np.pi) / f(azimuth - 1)
    v = []
    for p in range(polar):
        for a in range(azimuth):
            x = np.sin(f(p) * pstep, dtype=f) * np.cos(f(a) * astep, dtype=f)
            y = np.sin(f(p) * pstep, dtype=f) * np.sin(f(a) * astep, dtype=f)
            z = np.cos(f(p) * pstep, dtype=f)
            v.append((radius * float(x), radius * float(y), radius * float(z)))
    t = []
    for p in range(polar - 1):
        for a in range(azimuth - 1):
            p0, p1 = p * azimuth + a, p * azimuth + a + 1
            p2, p3 = (p + 1) * azimuth + a + 1, (p + 1) * azimuth + a
            t += [(p0, p2, p1), (p0, p3, p2)]
    return np.array(v, dtype=np.float64), np.array(t, dtype=np.uint32)


def _rot(i, j):  # AngleAxis(i deg, UnitZ) * AngleAxis(j deg, UnitY)
    a, b = np.deg2rad(i), np.deg2rad(j)
    Rz = np.array([[np.cos(a), -np.sin(a), 0], [np.sin(a), np.cos(a), 0], [0, 0, 1]])
    Ry = np.array([[np.cos(b), 0, np.sin(b)], [0, 1, 0], [-np.sin(b), 0, np.cos(b)]])
    return Rz @ Ry


@pytest.fixture(scope="module")
def scene(pkg):
    bb, g = pkg.bvh_builder, pkg.geometry
    m1, m2 = bb.Mesh(*_sphere_mesh(1.0)), bb.Mesh(*_sphere_mesh(2.0))
    R = np.array([_rot(i, j) for i, j in CASES for _ in DIRS])
    T = np.array([d for _ in CASES for d in DIRS], dtype=np.float64)
    tf_big = g.make_pose(R=R, T=T)       # compute(sphere2Tf, sphere1Tf, ...) with ComputeCollision(&sphere2, &sphere1)
    tf_small = g.make_pose(R=np.tile(np.eye(3), (len(R), 1, 1)), T=np.zeros((len(R), 3)))
    return m1, m2, tf_big, tf_small


def _request(abi):
    req = abi.default_collision_request()  # CollisionRequest(CONTACT | DISTANCE_LOWER_BOUND, 1)
    req.num_max_contacts, req.enable_contact = 1, 1
    return req


def test_oracle_and_device_headers(pkg, oracle, hostsim, scene):
    abi = pkg.abi
    m1, m2, tf_big, tf_small = scene
    ML = pkg.bvh_builder.MeshLibrary([m1, m2])
    n = len(tf_big)
    big, small = np.ones(n, dtype=np.uint32), np.zeros(n, dtype=np.uint32)
    ref = oracle.bvh_collide_batch(ML, big, small, tf_big, tf_small, _request(abi), n_threads=4)
    assert ref["num_contacts"].sum() > 0 and np.isfinite(ref["distance"][ref["num_contacts"] > 0]).all()
    got, _ = hostsim.bvh_collide_f64(abi, ML, big, small, tf_big, tf_small, _request(abi), max_contacts=10 ** 5)
    assert not np.any((got["status"] >> 30) & 1)
    assert np.array_equal(got["num_contacts"], ref["num_contacts"])
    assert np.array_equal(got["b1"], ref["b1"]) and np.array_equal(got["b2"], ref["b2"])
    fin = np.abs(ref["distance"]) < 1e300
    assert np.abs(got["distance"][fin] - ref["distance"][fin]).max() < 1e-12


@pytest.mark.gpu
def test_kernels(pkg, oracle, scene):
    abi, g = pkg.abi, pkg.geometry
    m1, m2, tf_big, tf_small = scene
    ML = pkg.bvh_builder.MeshLibrary([m1, m2])
    n = len(tf_big)
    L = g.ShapeLibrary()
    s_small, s_big = L.add_bvh(0, len(m1.vertices)), L.add_bvh(1, len(m2.vertices))
    lib = pkg.Library(L, device=0)
    try:
        lib.add_bvh(m1)
        lib.add_bvh(m2)
        got = lib.collide(np.full(n, s_big), np.full(n, s_small), tf_big, tf_small, _request(abi))
    finally:
        lib.close()
    ref = oracle.bvh_collide_batch(ML, np.ones(n, dtype=np.uint32), np.zeros(n, dtype=np.uint32), tf_big, tf_small,
                                   _request(abi), n_threads=4)
    assert not np.any((got["status"] >> 30) & 1) and not np.any(abi.status_skipped(got["status"]))
    near = np.abs(ref["distance"]) < 1e-9
    same = got["num_contacts"] == ref["num_contacts"]
    assert np.all(same | near)
    ok = same & ~near
    assert np.array_equal(got["b1"][ok], ref["b1"][ok]) and np.array_equal(got["b2"][ok], ref["b2"][ok])
    fin = ok & (np.abs(ref["distance"]) < 1e300)
    assert np.abs(got["distance"][fin] - ref["distance"][fin]).max() < 1e-6
