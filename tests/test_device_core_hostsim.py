"""CPU tests of the *device* per-pair code (hpp-fcl_amd/csrc/*.hpp compiled for the host by
tests/hostsim) against the fp64 oracle.  This validates the GJK/EPA core that ships inside the
HIP kernels in a container without a GPU; the GPU tests then only have to validate the
lane-group plumbing on top of it."""
import numpy as np
import pytest

from compare import check_parity, check_properties


CASES = ["cfg1_sphere_sphere", "cfg2_box_capsule", "cfg3_convex_convex", "cfg5_mixed", "all_primitives", "triangle_pairs"]


def _oracle(oracle, b, req, tf1, tf2):
    fn = oracle.distance_batch if b.kind == "distance" else oracle.collide_batch
    return fn(b.shapes, b.verts, b.s1, b.s2, tf1, tf2, req, n_threads=4)


@pytest.mark.parametrize("case", CASES)
def test_fp64_core_matches_oracle(pkg, oracle, hostsim, case):
    abi, wl = pkg.abi, pkg.workloads
    b = getattr(wl, case)(n=20000 if case != "cfg1_sphere_sphere" else 1000)
    req = wl.make_request(b, abi)
    ref = _oracle(oracle, b, req, b.tf1, b.tf2)
    got = hostsim.batch_f64(abi, b.shapes, b.verts, b.s1, b.s2, b.tf1, b.tf2, req)
    st = check_parity(abi, got, ref, dist_tol=1e-9, point_tol=1e-7, flag_band=1e-9, name=case)
    # both are fp64 without FMA contraction: in practice they agree to the last bits
    assert st["max_dd"] < 1e-12
    assert np.array_equal(got["status"], ref["status"])
    keep = np.ones(len(b), dtype=bool)
    if case == "triangle_pairs":
        # penetrating TriangleP x TriangleP: the reference reports -computePenetration along triangle 1's normal
        # next to GJK's (coinciding) witness points (triangle_triangle.cpp:88-92): p2 = p1 + d n does not apply
        k1, k2 = b.shapes["type"][b.s1], b.shapes["type"][b.s2]
        keep = ~((k1 == abi.GEOM_TRIANGLE) & (k2 == abi.GEOM_TRIANGLE) & (ref["distance"] <= 0))
        assert 0.15 < abi.status_contact(ref["status"]).mean() < 0.6
    check_properties(abi, got[keep], tol=1e-6, name=case)


@pytest.mark.parametrize("variant", [0, 1, 2])
@pytest.mark.parametrize("crit", [(0, 0), (1, 0), (1, 1), (2, 0), (2, 1)])
def test_fp64_core_variants(pkg, oracle, hostsim, variant, crit):
    """All GJK variants x convergence criteria (gjk.cpp:246-278, 372-425) agree with the oracle."""
    abi, wl = pkg.abi, pkg.workloads
    b = wl.cfg5_mixed(n=4000, seed=3)
    b.kind = "distance"
    req = abi.default_distance_request()
    req.q.gjk_variant = variant
    req.q.gjk_convergence_criterion, req.q.gjk_convergence_criterion_type = crit
    ref = _oracle(oracle, b, req, b.tf1, b.tf2)
    got = hostsim.batch_f64(abi, b.shapes, b.verts, b.s1, b.s2, b.tf1, b.tf2, req)
    assert np.array_equal(got["status"], ref["status"])
    assert np.nanmax(np.abs(got["distance"] - ref["distance"])) < 1e-12


@pytest.mark.parametrize("case", ["cfg5_mixed", "cfg3_convex_convex"])
def test_fp64_core_bounding_volume_guess(pkg, oracle, hostsim, case):
    """GJKInitialGuess::BoundingVolumeGuess (narrowphase.h:366-378): centre difference of the local AABBs."""
    abi, wl = pkg.abi, pkg.workloads
    b = getattr(wl, case)(n=6000, seed=5)
    req = wl.make_request(b, abi)
    req.q.gjk_initial_guess = abi.BoundingVolumeGuess
    ref = _oracle(oracle, b, req, b.tf1, b.tf2)
    got = hostsim.batch_f64(abi, b.shapes, b.verts, b.s1, b.s2, b.tf1, b.tf2, req)
    assert np.array_equal(got["status"], ref["status"])
    fin = np.isfinite(ref["distance"]) & (np.abs(ref["distance"]) < 1e300)
    assert np.abs(got["distance"][fin] - ref["distance"][fin]).max() < 1e-12
    # it is a different start than the default guess: the iteration counts differ somewhere
    req0 = wl.make_request(b, abi)
    ref0 = _oracle(oracle, b, req0, b.tf1, b.tf2)
    assert (abi.status_gjk_iters(ref0["status"]) != abi.status_gjk_iters(ref["status"])).mean() > 0.2
    assert np.abs(ref0["distance"][fin] - ref["distance"][fin]).max() < 1e-5


def test_fp64_core_collide_options(pkg, oracle, hostsim):
    """security margin, early stop (distance_upper_bound), enable_contact=false, cached guesses."""
    abi, wl = pkg.abi, pkg.workloads
    b = wl.cfg5_mixed(n=6000, seed=4)
    for margin, dub, contact in [(0.05, 0.1, 1), (-0.02, 1e300, 1), (0.0, 0.0, 0), (0.0, 0.3, 1)]:
        req = abi.default_collision_request()
        req.security_margin, req.distance_upper_bound, req.enable_contact = margin, dub, contact
        ref = _oracle(oracle, b, req, b.tf1, b.tf2)
        got = hostsim.batch_f64(abi, b.shapes, b.verts, b.s1, b.s2, b.tf1, b.tf2, req)
        assert np.array_equal(got["status"], ref["status"]), (margin, dub, contact)
        assert np.array_equal(got["num_contacts"], ref["num_contacts"])
        fin = np.isfinite(ref["distance"]) & (np.abs(ref["distance"]) < 1e300)
        assert np.abs(got["distance"][fin] - ref["distance"][fin]).max() < 1e-12
        assert np.array_equal(np.isnan(got["p1"]), np.isnan(ref["p1"]))
    # warm start: second call seeded with the first call's cached guess
    req = abi.default_distance_request()
    ref, g_ref = oracle.distance_batch(b.shapes, b.verts, b.s1, b.s2, b.tf1, b.tf2, req, want_guess=True)
    got, g_got = hostsim.batch_f64(abi, b.shapes, b.verts, b.s1, b.s2, b.tf1, b.tf2, req, want_guess=True)
    assert np.allclose(g_ref["gjk_guess"], g_got["gjk_guess"], atol=1e-12, equal_nan=True)
    req.q.gjk_initial_guess = abi.CachedGuess
    ref2 = oracle.distance_batch(b.shapes, b.verts, b.s1, b.s2, b.tf1, b.tf2, req, guess_in=g_ref)
    got2 = hostsim.batch_f64(abi, b.shapes, b.verts, b.s1, b.s2, b.tf1, b.tf2, req, guess_in=g_got)
    assert np.array_equal(got2["status"], ref2["status"])
    assert np.nanmax(np.abs(got2["distance"] - ref2["distance"])) < 1e-12
    # warm-started GJK needs fewer iterations on separated pairs
    sep = ref["distance"] > 1e-3
    assert abi.status_gjk_iters(ref2["status"])[sep].mean() < abi.status_gjk_iters(ref["status"])[sep].mean()


@pytest.mark.parametrize("case,dist_tol", [("cfg2_box_capsule", 1e-4), ("cfg3_convex_convex", 1e-4)])
def test_fp32_core_within_envelope(pkg, oracle, hostsim, case, dist_tol):
    """fp32 instantiation vs the fp64 oracle fed with the same (fp32-rounded) poses.  Envelope:
    |dd| <= 1e-4*(1+|d|) -- what the reference tolerates between its own GJK variants
    (test/accelerated_gjk.cpp:162); flags may only differ when |d_oracle| <= 1e-4."""
    abi, wl = pkg.abi, pkg.workloads
    b = getattr(wl, case)(n=30000)
    req = wl.make_request(b, abi)
    tf1, tf2 = b.tf_from_f32()
    ref = _oracle(oracle, b, req, tf1, tf2)
    got = hostsim.batch_f32(abi, b.shapes, b.verts, b.s1, b.s2, b.pose1_f32, b.pose2_f32, req)
    st = check_parity(abi, got, ref, dist_tol=dist_tol, point_tol=5e-4, flag_band=1e-4, name=case, fp32=True,
                      allow_bad_frac=2e-5)
    assert st["contact_frac"] > 0.2
    check_properties(abi, got, tol=2e-4, name=case)


def test_tetra_region_table_matches_oracle_tree(pkg, oracle):
    """The 4096-entry region table used by the kernels is exercised indirectly above; here the
    projection of random tetrahedra through raw GJK (rank-4 simplices) is compared."""
    abi, wl = pkg.abi, pkg.workloads
    b = wl.cfg3_convex_convex(n=3000, half_width=0.3)  # deep penetrations: many rank-4 projections
    req = wl.make_request(b, abi)
    import hostsim_binding as hs
    ref = oracle.distance_batch(b.shapes, b.verts, b.s1, b.s2, b.tf1, b.tf2, req)
    got = hs.batch_f64(abi, b.shapes, b.verts, b.s1, b.s2, b.tf1, b.tf2, req)
    assert (ref["distance"] < 0).mean() > 0.8
    assert np.array_equal(got["status"], ref["status"])
    assert np.abs(got["distance"] - ref["distance"]).max() < 1e-12


# ------------------------------------------------------------------------------------- BVH (cfg4)
@pytest.fixture(scope="module")
def small_meshes(pkg):
    bb = pkg.bvh_builder
    return bb.MeshLibrary([bb.Mesh(*bb.bumpy_sphere(12, 12)), bb.Mesh(*bb.bumpy_sphere(10, 14, phase=1.0)),
                           bb.Mesh(*bb.bumpy_sphere(16, 16, phase=2.0))])


def _mesh_queries(pkg, n, seed, nm=3, hw=1.2):
    g = pkg.geometry
    rng = np.random.default_rng(seed)

    def rq():
        q = rng.normal(size=(n, 4))
        return q / np.linalg.norm(q, axis=1, keepdims=True)

    return (rng.integers(0, nm, n), rng.integers(0, nm, n), g.make_pose(quat=rq(), T=rng.uniform(-hw, hw, (n, 3))),
            g.make_pose(quat=rq(), T=rng.uniform(-hw, hw, (n, 3))))


def test_bvh_builder_tree_is_well_formed(pkg, small_meshes):
    for m in small_meshes.meshes:
        nodes = m.nodes
        assert len(nodes) == 2 * m.num_tris - 1  # BVH_model.cpp:821-825
        leaves = nodes["first_child"][nodes["first_child"] < 0]
        assert sorted((-(leaves + 1)).tolist()) == list(range(m.num_tris))
        inner = nodes["first_child"][nodes["first_child"] > 0]
        assert sorted(np.concatenate([inner, inner + 1]).tolist()) == list(range(1, len(nodes)))
        # every node, leaves included: its OBB and its RSS contain the vertices of all its triangles (what culling on a
        # node relies on; fitted boxes of children may poke out of the parent's, the primitives never do), and a parent's
        # primitive range is the concatenation of its children's
        for i, nd in enumerate(nodes):
            prims = m.primitive_indices[nd["first_primitive"]:nd["first_primitive"] + nd["num_primitives"]]
            assert len(prims) == nd["num_primitives"] >= 1
            P = m.vertices[m.triangles[prims].reshape(-1)]
            axes = nd["obb_axes"].reshape(3, 3).T
            loc = (P - nd["obb_To"]) @ axes
            assert np.all(np.abs(loc) <= nd["obb_extent"] + 1e-9), i
            raxes = nd["rss_axes"].reshape(3, 3).T  # columns = axes; Tr = the rectangle's origin corner (BV/RSS.h)
            rl = (P - nd["rss_Tr"]) @ raxes
            dx = rl[:, 0] - np.clip(rl[:, 0], 0.0, nd["rss_length"][0])
            dy = rl[:, 1] - np.clip(rl[:, 1], 0.0, nd["rss_length"][1])
            assert np.all(np.sqrt(dx * dx + dy * dy + rl[:, 2] ** 2) <= nd["rss_radius"] + 1e-9), i
            fc = nd["first_child"]
            if fc > 0:
                l, r = nodes[fc], nodes[fc + 1]
                assert l["first_primitive"] == nd["first_primitive"]
                assert r["first_primitive"] == l["first_primitive"] + l["num_primitives"]
                assert l["num_primitives"] + r["num_primitives"] == nd["num_primitives"]
            else:
                assert nd["num_primitives"] == 1 and -(fc + 1) == prims[0]


def test_oracle_bvh_contact_set_equals_brute_force(pkg, oracle, small_meshes):
    """test/collision.cpp:625-654 style: with num_max_contacts = inf the sorted contact set (b1,b2) is
    traversal independent -- compare the BVH traversal against all triangle pairs."""
    abi, g = pkg.abi, pkg.geometry
    m1, m2 = small_meshes.meshes[0], small_meshes.meshes[1]
    i1, i2, tf1, tf2 = _mesh_queries(pkg, 12, 1)
    i1[:], i2[:] = 0, 1
    req = abi.default_collision_request()
    req.num_max_contacts = 10 ** 6
    out, contacts = oracle.bvh_collide_batch(small_meshes, i1, i2, tf1, tf2, req, max_contacts=10 ** 6)
    L = g.ShapeLibrary()
    for mm in (m1, m2):
        for t3 in mm.triangles:
            L.add_triangle(*mm.vertices[t3])
    a = np.repeat(np.arange(m1.num_tris), m2.num_tris)
    b = m1.num_tris + np.tile(np.arange(m2.num_tris), m1.num_tris)
    for k in range(len(i1)):
        r = oracle.collide_batch(L.shapes_array(), L.vertices_array(), a, b, np.tile(tf1[k], (len(a), 1)),
                                 np.tile(tf2[k], (len(a), 1)), req)
        hit = r["num_contacts"] > 0
        brute = set(zip(a[hit].tolist(), (b[hit] - m1.num_tris).tolist()))
        ck = contacts[contacts["pair"] == k]
        assert brute == set(zip(ck["b1"].tolist(), ck["b2"].tolist()))
        assert out["num_contacts"][k] == len(brute)


@pytest.mark.parametrize("nmax,margin", [(1, 0.0), (10 ** 6, 0.0), (3, 0.02), (1, -0.01)])
def test_bvh_device_code_matches_oracle(pkg, oracle, hostsim, small_meshes, nmax, margin):
    """OBB SAT + triangle-triangle leaf + DFS order of the device headers (host build) vs the oracle:
    identical contact counts, first-contact primitive ids, contact lists (order included)."""
    abi = pkg.abi
    i1, i2, tf1, tf2 = _mesh_queries(pkg, 1500, 2)
    req = abi.default_collision_request()
    req.num_max_contacts, req.security_margin = nmax, margin
    ref, cref = oracle.bvh_collide_batch(small_meshes, i1, i2, tf1, tf2, req, max_contacts=2 * 10 ** 6, n_threads=1)
    got, cgot = hostsim.bvh_collide_f64(abi, small_meshes, i1, i2, tf1, tf2, req, max_contacts=2 * 10 ** 6)
    assert (ref["num_contacts"] > 0).mean() > 0.3
    assert np.array_equal(ref["num_contacts"], got["num_contacts"])
    assert np.array_equal(ref["b1"], got["b1"]) and np.array_equal(ref["b2"], got["b2"])
    fin = np.abs(ref["distance"]) < 1e300
    assert np.abs(ref["distance"][fin] - got["distance"][fin]).max() < 1e-12
    assert np.array_equal(np.isnan(ref["p1"]), np.isnan(got["p1"]))
    assert len(cref) == len(cgot)
    assert np.array_equal(cref["pair"], cgot["pair"]) and np.array_equal(cref["b1"], cgot["b1"])
    assert np.allclose(cref["penetration_depth"], cgot["penetration_depth"], atol=1e-12)


def test_rect_distance_two_formulations_and_brute_force(pkg, oracle, hostsim):
    """rectDistance (RSS.cpp:121-713): the oracle's block-by-block transcription, the device's
    edge-pair-parametric formulation and a bounded QP agree."""
    from scipy.optimize import minimize
    g, abi = pkg.geometry, pkg.abi
    rng = np.random.default_rng(3)
    for k in range(3000):
        q = rng.normal(size=4)
        R = g.quat_to_matrix(q / np.linalg.norm(q))
        T, a, b = rng.uniform(-3, 3, 3), rng.uniform(0.05, 1.5, 2), rng.uniform(0.05, 1.5, 2)
        d_or = oracle.rect_distance(R, T, a, b)
        assert abs(d_or - hostsim.rect_distance(abi, R, T, a, b)) < 1e-12
        if k < 60:
            def f(x):
                d = T + R[:, 0] * x[2] + R[:, 1] * x[3] - np.array([x[0], x[1], 0.0])
                return d @ d
            best = min(minimize(f, x0, bounds=[(0, a[0]), (0, a[1]), (0, b[0]), (0, b[1])], method="L-BFGS-B",
                                options=dict(ftol=1e-15, gtol=1e-12)).fun
                       for x0 in [(0, 0, 0, 0), (a[0], a[1], b[0], b[1]), (a[0] / 2, a[1] / 2, b[0] / 2, b[1] / 2)])
            assert d_or <= np.sqrt(best) + 1e-6  # a valid lower bound (exact in practice)
            assert abs(d_or - np.sqrt(best)) < 1e-4


def test_sqr_tri_distance_vs_gjk(pkg, oracle, hostsim):
    """sqrTriDistance (intersect.cpp:156-368) against the (independently pinned) GJK triangle distance."""
    g, abi = pkg.geometry, pkg.abi
    rng = np.random.default_rng(4)
    n = 1500
    S = rng.uniform(-1, 1, (n, 3, 3))
    T = rng.uniform(-1, 1, (n, 3, 3)) + rng.uniform(-1.5, 1.5, (n, 1, 3))
    L = g.ShapeLibrary()
    for i in range(n):
        L.add_triangle(*S[i])
    for i in range(n):
        L.add_triangle(*T[i])
    I = np.tile(g.make_pose(), (n, 1))
    r = oracle.collide_batch(L.shapes_array(), L.vertices_array(), np.arange(n), n + np.arange(n), I, I)  # TriangleP: collide() only
    for i in range(n):
        d2, P, Q = oracle.sqr_tri_distance(S[i], T[i])
        d2s, Ps, Qs = hostsim.sqr_tri_distance(abi, S[i], T[i])
        assert d2 == d2s
        if abi.status_gjk(r["status"][i]) == abi.GJK_Collision:
            assert d2 < 1e-10
        else:
            assert abs(np.sqrt(d2) - r["distance"][i]) < 1e-9
            assert np.allclose(P, Ps, atol=1e-12) and np.allclose(Q, Qs, atol=1e-12)


def test_bvh_distance_device_code_matches_oracle_and_brute_force(pkg, oracle, hostsim, small_meshes):
    abi, g = pkg.abi, pkg.geometry
    i1, i2, tf1, tf2 = _mesh_queries(pkg, 600, 5, hw=2.5)
    ref = oracle.bvh_distance_batch(small_meshes, i1, i2, tf1, tf2)
    got = hostsim.bvh_distance_f64(abi, small_meshes, i1, i2, tf1, tf2)
    assert 0.05 < (ref["distance"] == 0).mean() < 0.6
    assert np.abs(ref["distance"] - got["distance"]).max() < 1e-12
    assert np.array_equal(ref["b1"], got["b1"]) and np.array_equal(ref["b2"], got["b2"])
    assert np.nanmax(np.abs(ref["p1"] - got["p1"])) < 1e-12 and np.isnan(got["normal"]).all()
    for k in range(4):  # all triangle pairs
        A, B = small_meshes.meshes[i1[k]], small_meshes.meshes[i2[k]]
        VA = A.vertices @ g.pose_R(tf1[k]).T + g.pose_T(tf1[k])
        VB = B.vertices @ g.pose_R(tf2[k]).T + g.pose_T(tf2[k])
        best = min(oracle.sqr_tri_distance(VA[ta], VB[tb])[0] for ta in A.triangles[::1] for tb in B.triangles[::1])
        assert abs(np.sqrt(best) - ref["distance"][k]) < 1e-9


# ------------------------------------------------------------------ Plane / Halfspace rows
def _same(a, b, tol=0.0):
    a, b = np.asarray(a, dtype=np.float64), np.asarray(b, dtype=np.float64)
    both_nan = np.isnan(a) & np.isnan(b)
    if tol == 0.0:
        return bool(np.all(both_nan | (a == b)))
    with np.errstate(invalid="ignore", over="ignore"):
        return bool(np.all(both_nan | (a == b) | (np.abs(a - b) <= tol * (1 + np.abs(b)))))


@pytest.mark.parametrize("kind", ["collide", "distance"])
def test_flat_rows_match_oracle(pkg, oracle, hostsim, kind):
    abi, wl = pkg.abi, pkg.workloads
    b = wl.flat_pairs(n=20000, kind=kind)
    req = wl.make_request(b, abi)
    ref = _oracle(oracle, b, req, b.tf1, b.tf2)
    got = hostsim.batch_f64(abi, b.shapes, b.verts, b.s1, b.s2, b.tf1, b.tf2, req)
    assert np.array_equal(got["status"], ref["status"])
    assert not abi.status_skipped(ref["status"]).any()
    for f in ("distance", "normal", "p1", "p2"):
        assert _same(got[f], ref[f]), f
    inf = ref["distance"] == -np.finfo(np.float64).max
    assert 0.02 < inf.mean() < 0.2  # crossing / nested flats: infinite penetration (details.h:520-560)


# ------------------------------------------------------------------ hulls above 32 vertices
@pytest.mark.parametrize("kind", ["distance", "collide"])
def test_large_hulls_scan_vs_reference_hill_climbing(pkg, oracle, hostsim, kind):
    """The device scans all vertices of a large hull (first maximum); the reference climbs the neighbour graph
    from a hint (support_functions.cpp:323-397).  On hulls in general position both reach the same vertex, so
    GJK/EPA follow the same path: statuses equal, distances to round-off."""
    abi, wl = pkg.abi, pkg.workloads
    b = wl.large_convex(n=6000, kind=kind)
    assert oracle.register_hull_neighbors(b.shapes, b.verts) >= 60
    req = wl.make_request(b, abi)
    ref = _oracle(oracle, b, req, b.tf1, b.tf2)
    got = hostsim.batch_f64(abi, b.shapes, b.verts, b.s1, b.s2, b.tf1, b.tf2, req)
    st = check_parity(abi, got, ref, dist_tol=1e-9, point_tol=1e-7, flag_band=1e-9, name="large-" + kind)
    assert st["max_dd"] < 1e-10
    assert (abi.status_gjk(got["status"]) == abi.status_gjk(ref["status"])).all()
    # (a climb can stop on a vertex whose rounded dot product ties with the scan's winner: ~1 % of the paths differ)
    assert (abi.status_gjk_iters(got["status"]) == abi.status_gjk_iters(ref["status"])).mean() > 0.98
    frac = float(abi.status_contact(ref["status"]).mean())
    assert 0.15 < frac < 0.85, frac
    oracle.lib().orc_clear_neighbors()


@pytest.mark.parametrize("margin", [0.0, 0.05])
def test_bvh_fp32_filter_is_exact(pkg, hostsim, margin):
    """The fp32 separating-axis filter of k_bvh_collide<double, ., FILT> (hfcl_bvh.hpp: obb_filter) in front of the fp64 test:
    every verdict it commits to is the fp64 test's (no unsafe decision in ~600k node pairs of real traversals), the rank
    comparison is the fp64 size comparison, and the walk built on it -- candidates for the minimum of the bound resolved in
    fp64 only when something is compared with them -- produces byte-identical records to the plain fp64 walk."""
    abi, wl = pkg.abi, pkg.workloads
    for kw in (dict(n=1500), dict(n=3000, seg=12, ring=12, n_variants=4)):
        b = wl.cfg4_mesh_mesh(**kw)
        ML = pkg.bvh_builder.MeshLibrary(b.meshes)
        req = wl.make_request(b, abi, security_margin=margin)
        plain = hostsim.bvh_collide_f64(abi, ML, b.s1, b.s2, b.tf1, b.tf2, req)
        filt, st = hostsim.bvh_collide_filtered_f64(abi, ML, b.s1, b.s2, b.tf1, b.tf2, req)
        assert st["unsafe"] == 0 and st["rank_mismatch"] == 0, st
        assert plain.tobytes() == filt.tobytes()
        assert st["bv_tests"] > 100 * len(b) * 0.5
        # the filter decides nearly everything: fp64 tests per query stay a handful
        assert (st["disjoint_value_needed"] + st["unsure"]) < 4 * len(b), st
        assert st["unsure"] < 0.01 * st["bv_tests"], st
