"""Swept-sphere radius (test/swept_sphere_radius.cpp:40-240): sweeping the shapes of a pair by spheres of radii r1, r2
lowers the distance by r1 + r2, leaves the normal alone and moves the witness points by r1 n / -r2 n -- for the GJK/EPA
pairs and for the specialised (closed-form) ones alike.  Checked on the oracle (CPU), the host build of the device
headers (CPU) and the kernels (GPU)."""
import numpy as np
import pytest

KINDS = ["box", "sphere", "capsule", "ellipsoid", "convex", "cone", "cylinder"]


def _library(pkg, rng, nper, ssr):
    """nper shapes of each kind; ssr = array of swept-sphere radii (one per shape, same order)."""
    g, wl = pkg.geometry, pkg.workloads
    L = g.ShapeLibrary()
    base = wl.fibonacci_sphere(20)
    k = 0
    for s in rng.uniform(0.2, 1.0, (nper, 3)):
        L.add_box(*map(float, s), swept_sphere_radius=float(ssr[k])); k += 1
    for r in rng.uniform(0.2, 1.0, nper):
        L.add_sphere(float(r), float(ssr[k])); k += 1
    for r, lz in zip(rng.uniform(0.1, 0.6, nper), rng.uniform(0.2, 1.0, nper)):
        L.add_capsule(float(r), float(lz), float(ssr[k])); k += 1
    for r in rng.uniform(0.2, 1.0, (nper, 3)):
        L.add_ellipsoid(*map(float, r), swept_sphere_radius=float(ssr[k])); k += 1
    for r in rng.uniform(0.2, 1.0, (nper, 3)):
        L.add_convex(base * r, float(ssr[k])); k += 1
    for r, lz in zip(rng.uniform(0.1, 0.6, nper), rng.uniform(0.2, 1.0, nper)):
        L.add_cone(float(r), float(lz), float(ssr[k])); k += 1
    for r, lz in zip(rng.uniform(0.1, 0.6, nper), rng.uniform(0.2, 1.0, nper)):
        L.add_cylinder(float(r), float(lz), float(ssr[k])); k += 1
    return L


def _scene(pkg, n=6000, nper=12, seed=3):
    rng = np.random.default_rng(seed)
    ns = nper * len(KINDS)
    ssr = rng.choice([0.0, 0.1, 1.0], ns)  # the reference's radii (its 10.0 only with a looser tolerance)
    L0 = _library(pkg, np.random.default_rng(seed + 1), nper, np.zeros(ns))
    L1 = _library(pkg, np.random.default_rng(seed + 1), nper, ssr)
    s1, s2 = rng.integers(0, ns, n), rng.integers(0, ns, n)
    q = rng.normal(size=(2, n, 4))
    q /= np.linalg.norm(q, axis=2, keepdims=True)
    g = pkg.geometry
    tf1 = g.make_pose(quat=q[0], T=rng.uniform(-2, 2, (n, 3)))  # extents of the reference's test
    tf2 = g.make_pose(quat=q[1], T=rng.uniform(-2, 2, (n, 3)))
    return L0, L1, ssr, s1, s2, tf1, tf2


def _check(abi, r0, r1, ra, rb, name):
    """r0: records without sweeping, r1: with; ra / rb: swept radii of shape 1 / 2 per pair."""
    fin = np.isfinite(r0["p1"]).all(1) & np.isfinite(r1["p1"]).all(1) & np.isfinite(r0["distance"]) & (np.abs(r0["distance"]) < 1e300)
    assert fin.mean() > 0.95
    d0, d1 = r0["distance"][fin], r1["distance"][fin]
    ra, rb = ra[fin], rb[fin]
    # smooth shapes (ellipsoid, cone, cylinder): EPA stops on its tolerance -> 3 sqrt(tol) as in the reference
    tol = 3e-3
    assert np.abs(d1 - (d0 - ra - rb)).max() < tol, name
    n0, n1 = r0["normal"][fin], r1["normal"][fin]
    assert ((n0 * n1).sum(1) > 1 - tol).all(), name
    assert np.abs(r1["p1"][fin] - (r0["p1"][fin] + ra[:, None] * n0)).max() < 2 * tol, name
    assert np.abs(r1["p2"][fin] - (r0["p2"][fin] - rb[:, None] * n0)).max() < 2 * tol, name
    # the bulk agrees to round-off: the radii are added after GJK / EPA, never inside the iterations
    assert np.quantile(np.abs(d1 - (d0 - ra - rb)), 0.99) < 1e-9, name
    assert (d0 <= 0).mean() > 0.03 and (d0 > 0).mean() > 0.3


def test_oracle_and_device_headers(pkg, oracle, hostsim):
    abi = pkg.abi
    L0, L1, ssr, s1, s2, tf1, tf2 = _scene(pkg)
    req = abi.default_distance_request()
    r0 = oracle.distance_batch(L0.shapes_array(), L0.vertices_array(), s1, s2, tf1, tf2, req, n_threads=4)
    r1 = oracle.distance_batch(L1.shapes_array(), L1.vertices_array(), s1, s2, tf1, tf2, req, n_threads=4)
    _check(abi, r0, r1, ssr[s1], ssr[s2], "oracle")
    h1 = hostsim.batch_f64(abi, L1.shapes_array(), L1.vertices_array(), s1, s2, tf1, tf2, req)
    assert np.array_equal(h1["status"], r1["status"])
    fin = np.isfinite(r1["distance"]) & (np.abs(r1["distance"]) < 1e300)
    assert np.abs(h1["distance"][fin] - r1["distance"][fin]).max() < 1e-12


@pytest.mark.gpu
def test_kernels(pkg, oracle):
    abi = pkg.abi
    L0, L1, ssr, s1, s2, tf1, tf2 = _scene(pkg, n=40000, seed=4)
    req = abi.default_distance_request()
    out = []
    for L in (L0, L1):
        lib = pkg.Library(L, device=0)
        try:
            out.append(lib.distance(s1, s2, tf1, tf2, req))
        finally:
            lib.close()
    _check(abi, out[0], out[1], ssr[s1], ssr[s2], "gpu")
    ref = oracle.distance_batch(L1.shapes_array(), L1.vertices_array(), s1, s2, tf1, tf2, req, n_threads=8)
    assert (abi.status_contact(ref["status"]) == abi.status_contact(out[1]["status"])).mean() > 0.9999
