"""BVHModel<OBBRSS> x convex shape collide(): oracle restatement of
BVHShapeCollider / MeshShapeCollisionTraversalNode (collision_func_matrix.cpp:102-155,
traversal_node_bvh_shape.h:98-188) checked against brute force over all triangles, then the device."""
import numpy as np
import pytest


def _scene(pkg, n=400, seed=1, **kw):
    return pkg.workloads.mesh_vs_shapes(n=n, seed=seed, **kw)


def test_oracle_mesh_shape_contacts_equal_brute_force(pkg, oracle):
    """With num_max_contacts = inf the contact set is every triangle whose distance to the shape is <= 0:
    compare with the shape-vs-triangle narrow phase run on ALL triangles (no BVH)."""
    abi, g, bb = pkg.abi, pkg.geometry, pkg.bvh_builder
    b = _scene(pkg, n=120, seed=3)
    ML = bb.MeshLibrary(b.meshes)
    req = abi.default_collision_request()
    req.num_max_contacts = 10 ** 6
    kinds = b.shapes["type"]
    is_mesh1, is_mesh2 = kinds[b.s1] == abi.BV_OBBRSS, kinds[b.s2] == abi.BV_OBBRSS
    sel = np.nonzero(is_mesh1 != is_mesh2)[0]
    out, cs = oracle.mixed_collide_batch(b.shapes, b.verts, ML, b.s1[sel], b.s2[sel], b.tf1[sel], b.tf2[sel], req,
                                         max_contacts=10 ** 6)
    assert 0.2 < (out["num_contacts"] > 0).mean() < 0.95
    checked = 0
    for k, i in enumerate(sel[:60]):
        mesh_first = bool(is_mesh1[i])
        ms, ss = (b.s1[i], b.s2[i]) if mesh_first else (b.s2[i], b.s1[i])
        tfm, tfs = (b.tf1[i], b.tf2[i]) if mesh_first else (b.tf2[i], b.tf1[i])
        m = b.meshes[int(b.shapes[ms]["bvh_index"])]
        L = g.ShapeLibrary()
        for t3 in m.triangles:
            L.add_triangle(*m.vertices[t3])
        # the solid shape, copied into the brute-force library
        src = b.shapes[ss]
        base = len(L)
        if src["type"] == abi.GEOM_CONVEX:
            L.add_convex(b.verts[src["vertex_offset"]:src["vertex_offset"] + src["num_points"]])
        else:
            L._add(int(src["type"]), tuple(src["params"]), float(src["swept_sphere_radius"]))
        nt = m.num_tris
        r = oracle.collide_batch(L.shapes_array(), L.vertices_array(), np.arange(nt), np.full(nt, base), np.tile(tfm, (nt, 1)),
                                 np.tile(tfs, (nt, 1)), req)
        brute = set(np.nonzero(r["num_contacts"] > 0)[0].tolist())
        ck = cs[cs["pair"] == k]
        got = set((ck["b1"] if mesh_first else ck["b2"]).tolist())
        assert got == brute, (i, len(got), len(brute))
        assert out["num_contacts"][k] == len(brute)
        # the other primitive id is Contact::NONE; normals point from the caller's first object to the second
        assert ((ck["b2"] if mesh_first else ck["b1"]) == -1).all()
        checked += len(brute)
    assert checked > 100


def test_oracle_operand_swap(pkg, oracle):
    """collide(shape, mesh) = collide(mesh, shape) with contacts mirrored (src/collision.cpp:93-108)."""
    abi, bb = pkg.abi, pkg.bvh_builder
    b = _scene(pkg, n=300, seed=4)
    ML = bb.MeshLibrary(b.meshes)
    req = abi.default_collision_request()
    a = oracle.mixed_collide_batch(b.shapes, b.verts, ML, b.s1, b.s2, b.tf1, b.tf2, req)
    s = oracle.mixed_collide_batch(b.shapes, b.verts, ML, b.s2, b.s1, b.tf2, b.tf1, req)
    kinds = b.shapes["type"]
    mixed = (kinds[b.s1] == abi.BV_OBBRSS) != (kinds[b.s2] == abi.BV_OBBRSS)
    assert mixed.sum() > 200
    assert np.array_equal(a["num_contacts"][mixed], s["num_contacts"][mixed])
    assert np.array_equal(a["b1"][mixed], s["b2"][mixed]) and np.array_equal(a["b2"][mixed], s["b1"][mixed])
    same = lambda x, y: np.all((x == y) | (np.isnan(x) & np.isnan(y)))
    assert same(a["distance"][mixed], s["distance"][mixed])
    assert same(a["normal"][mixed], -s["normal"][mixed]) and same(a["p1"][mixed], s["p2"][mixed])


def test_oracle_shape_bv_encloses_shape(pkg, oracle):
    """computeBV<OBBRSS,S>: the OBB fitted to the bound vertices contains them (and the shape's support points
    along its axes stay within the polyhedral bound the reference uses)."""
    # exercised through the traversal: a shape far away from the mesh never reports a contact and the lower
    # bound it reports is a true lower bound of the brute-force distance
    abi, g, bb = pkg.abi, pkg.geometry, pkg.bvh_builder
    b = _scene(pkg, n=200, seed=5, half_width=3.0)
    ML = bb.MeshLibrary(b.meshes)
    req = abi.default_collision_request()
    out = oracle.mixed_collide_batch(b.shapes, b.verts, ML, b.s1, b.s2, b.tf1, b.tf2, req)
    kinds = b.shapes["type"]
    mixed = (kinds[b.s1] == abi.BV_OBBRSS) != (kinds[b.s2] == abi.BV_OBBRSS)
    free = mixed & (out["num_contacts"] == 0)
    assert free.sum() > 50
    assert (out["distance"][free] > 0).all()


def _mixed_only(pkg, b):
    kinds = b.shapes["type"]
    return np.nonzero((kinds[b.s1] == pkg.abi.BV_OBBRSS) != (kinds[b.s2] == pkg.abi.BV_OBBRSS))[0]


def _same(a, b, tol=0.0):
    a, b = np.asarray(a, dtype=np.float64), np.asarray(b, dtype=np.float64)
    with np.errstate(invalid="ignore"):
        return bool(np.all((a == b) | (np.isnan(a) & np.isnan(b)) | (np.abs(a - b) <= tol)))


@pytest.mark.parametrize("nmax,margin,cached", [(1, 0.0, False), (10 ** 6, 0.0, False), (2, 0.03, False), (1, 0.0, True)])
def test_device_headers_match_oracle(pkg, oracle, hostsim, nmax, margin, cached):
    """Host build of hfcl_bvh_shape.hpp (one lane per query) vs the oracle: contact counts, primitive ids,
    contact order and distances are identical; normals / witness points agree to the last bits (the two
    code bases normalise vectors with differently ordered divisions: ~1 ulp)."""
    abi, bb = pkg.abi, pkg.bvh_builder
    b = _scene(pkg, n=3000, seed=6)
    ML = bb.MeshLibrary(b.meshes)
    sel = _mixed_only(pkg, b)
    req = abi.default_collision_request()
    req.num_max_contacts, req.security_margin = nmax, margin
    if cached:
        req.q.gjk_initial_guess = abi.CachedGuess
        req.q.cached_gjk_guess[:] = [0.3, -0.2, 0.9]
    a = (b.shapes, b.verts, ML, b.s1[sel], b.s2[sel], b.tf1[sel], b.tf2[sel], req)
    ref, cref, gref = oracle.mixed_collide_batch(*a, max_contacts=10 ** 6, want_guess=True)
    got, cgot, ggot = hostsim.mesh_shape_collide_f64(abi, *a, max_contacts=10 ** 6, want_guess=True)
    assert 0.1 < (ref["num_contacts"] > 0).mean() < 0.9
    assert np.array_equal(ref["num_contacts"], got["num_contacts"])
    assert np.array_equal(ref["b1"], got["b1"]) and np.array_equal(ref["b2"], got["b2"])
    assert _same(got["distance"], ref["distance"], 1e-15)
    for f in ("normal", "p1", "p2"):
        assert _same(got[f], ref[f], 1e-11), f
    assert len(cref) == len(cgot) and np.array_equal(cref["pair"], cgot["pair"])
    assert np.array_equal(cref["b1"], cgot["b1"]) and np.array_equal(cref["b2"], cgot["b2"])
    for f in ("penetration_depth", "normal", "p1", "p2"):
        assert _same(cgot[f], cref[f], 1e-11), f  # normals of near-touching pairs amplify the last-bit differences
    assert _same(ggot["gjk_guess"], gref["gjk_guess"], 1e-13)


@pytest.mark.parametrize("margin,cached", [(0.0, False), (0.03, False), (0.0, True)])
def test_lane_form_equals_group_form(pkg, hostsim, margin, cached):
    """The one-query-per-lane form of the traversal (k_bvh_collide<SOLID>: lane-local walk and GJK, EPA leaves finished by
    k_bvh_shape_finish from a queue) against the group form, host build of the same headers: every record, contact and
    cached guess byte for byte (first-contact requests; the others take the group form)."""
    abi, bb = pkg.abi, pkg.bvh_builder
    b = _scene(pkg, n=3000, seed=16)
    ML = bb.MeshLibrary(b.meshes)
    sel = _mixed_only(pkg, b)
    req = abi.default_collision_request()
    req.security_margin = margin
    if cached:
        req.q.gjk_initial_guess = abi.CachedGuess
        req.q.cached_gjk_guess[:] = [0.3, -0.2, 0.9]
    a = (b.shapes, b.verts, ML, b.s1[sel], b.s2[sel], b.tf1[sel], b.tf2[sel], req)
    ref, cref, gref = hostsim.mesh_shape_collide_f64(abi, *a, max_contacts=10 ** 6, want_guess=True)
    hostsim.set_shape_lane(True)
    try:
        got, cgot, ggot = hostsim.mesh_shape_collide_f64(abi, *a, max_contacts=10 ** 6, want_guess=True)
    finally:
        hostsim.set_shape_lane(False)
    assert 0.1 < (ref["num_contacts"] > 0).mean() < 0.9
    assert ref.tobytes() == got.tobytes()
    assert cref.tobytes() == cgot.tobytes()
    assert gref.tobytes() == ggot.tobytes()


@pytest.mark.gpu
@pytest.mark.parametrize("nmax,margin", [(1, 0.0), (10 ** 6, 0.0), (3, 0.02)])
def test_gpu_mesh_vs_shapes(pkg, oracle, nmax, margin):
    """k_bvh_shape (+ mesh x mesh and shape x shape pairs of the same batch) vs the oracle."""
    abi, wl, bb = pkg.abi, pkg.workloads, pkg.bvh_builder
    b = _scene(pkg, n=20000, seed=7)
    ML = bb.MeshLibrary(b.meshes)
    req = abi.default_collision_request()
    req.num_max_contacts, req.security_margin = nmax, margin
    ref, cref = oracle.mixed_collide_batch(b.shapes, b.verts, ML, b.s1, b.s2, b.tf1, b.tf2, req, max_contacts=4 * 10 ** 6,
                                           n_threads=16)
    lib = wl.make_library(pkg, b)
    try:
        got, cgot, produced = lib.collide_contacts(b.s1, b.s2, b.tf1, b.tf2, req, 4 * 10 ** 6)
        buckets = lib.last_bucket_counts()
    finally:
        lib.close()
    assert buckets["bvh_shape"] > 15000 and buckets["unsupported"] == 0
    assert not ((got["status"] >> 30) & 1).any()
    kinds = b.shapes["type"]
    mixed = (kinds[b.s1] == abi.BV_OBBRSS) != (kinds[b.s2] == abi.BV_OBBRSS)
    near = np.abs(ref["distance"]) < 1e-9  # decision boundary
    ok = (got["num_contacts"] == ref["num_contacts"]) | near
    assert ok[mixed].all(), int((~ok[mixed]).sum())
    m = mixed & ~near & (got["num_contacts"] == ref["num_contacts"])
    assert np.array_equal(got["b1"][m], ref["b1"][m]) and np.array_equal(got["b2"][m], ref["b2"][m])
    # distance field: exact semantics for contacts; for contact-free queries it is the running lower bound, which
    # mixes leaf distances with OBB-vs-OBB bounds.  The solid's OBB is a PCA fit of its bound vertices
    # (geometric_shapes_utility.h:73-82): for solids of revolution (sphere, capsule, cone, cylinder) two or three
    # eigenvalues coincide and the box orientation is decided by rounding noise -- in the reference as well --
    # so there only "a valid positive lower bound" can be asserted (checked against brute force below).
    solid = np.where(kinds[b.s1] == abi.BV_OBBRSS, kinds[b.s2], kinds[b.s1])
    round_solid = np.isin(solid, [abi.GEOM_SPHERE, abi.GEOM_CAPSULE, abi.GEOM_CONE, abi.GEOM_CYLINDER])
    fin = m & (np.abs(ref["distance"]) < 1e300)
    strict = fin & ((ref["num_contacts"] > 0) | ~round_solid)
    assert np.abs(got["distance"][strict] - ref["distance"][strict]).max() < 4e-6  # EPA on smooth solids: test_gpu_parity
    loose = fin & ~strict
    assert loose.sum() > 1000 and (got["distance"][loose] > 0).all()
    g = pkg.geometry
    for i in np.nonzero(loose)[0][:25]:  # true distance = min over all triangles (no BVH)
        mesh_first = kinds[b.s1[i]] == abi.BV_OBBRSS
        ms, ss = (b.s1[i], b.s2[i]) if mesh_first else (b.s2[i], b.s1[i])
        tfm, tfs = (b.tf1[i], b.tf2[i]) if mesh_first else (b.tf2[i], b.tf1[i])
        mesh = b.meshes[int(b.shapes[ms]["bvh_index"])]
        L = g.ShapeLibrary()
        for t3 in mesh.triangles:
            L.add_triangle(*mesh.vertices[t3])
        src = b.shapes[ss]
        base = L._add(int(src["type"]), tuple(src["params"]), float(src["swept_sphere_radius"]))
        nt = mesh.num_tris
        # per-triangle internal::ShapeShapeDistance<TriangleP, S> = what collide() runs for a top-level TriangleP
        # (distance() has no TriangleP rows, src/distance_func_matrix.cpp); default request: penetration computed
        r = oracle.collide_batch(L.shapes_array(), L.vertices_array(), np.arange(nt), np.full(nt, base), np.tile(tfm, (nt, 1)),
                                 np.tile(tfs, (nt, 1)), None, n_threads=8)
        assert got["distance"][i] <= r["distance"].min() + 1e-9 + margin, (i, got["distance"][i], r["distance"].min())
    assert np.array_equal(np.isnan(got["p1"][strict]), np.isnan(ref["p1"][strict]))
    # contact lists of the mixed pairs: same (pair, b1, b2) multiset
    pm = np.nonzero(m)[0]
    key = lambda c: sorted((int(p), int(x), int(y)) for p, x, y in zip(c["pair"], c["b1"], c["b2"]) if mixed[p] and not near[p])
    if nmax > 3:
        assert key(cref) == key(cgot)
    frac = (ref["num_contacts"][mixed] > 0).mean()
    assert 0.2 < frac < 0.9, frac


def _device_collide(pkg, b, req, env=None, f32=False):
    """Records of the device-resident entry point for batch b (library created with the options `env`)."""
    import torch
    abi, wl = pkg.abi, pkg.workloads
    lib = wl.make_library(pkg, b, options=env)  # (hfcl_lib_set_option takes the HFCL_ spelling of a key too)
    dev = torch.device("cuda:0")
    try:
        s1 = torch.from_numpy(b.s1.astype(np.int32)).to(dev)
        s2 = torch.from_numpy(b.s2.astype(np.int32)).to(dev)
        n = len(b)
        if f32:
            p1, p2 = torch.from_numpy(b.pose1_f32).to(dev), torch.from_numpy(b.pose2_f32).to(dev)
            out = torch.zeros(n * 11, dtype=torch.int32, device=dev)
            lib.collide_device_f32(s1, s2, p1, p2, n, req, out)
            torch.cuda.synchronize()
            return out.cpu().numpy().view(abi.RESULT_F32_DTYPE).copy()
        p1, p2 = torch.from_numpy(b.tf1).to(dev), torch.from_numpy(b.tf2).to(dev)
        out = torch.zeros(n * 24, dtype=torch.int32, device=dev)
        lib.collide_device(s1, s2, p1, p2, n, req, out)
        torch.cuda.synchronize()
        return out.cpu().numpy().view(abi.RESULT_DTYPE).copy()
    finally:
        lib.close()


@pytest.mark.gpu
def test_gpu_mesh_solid_ties_forms_agree(pkg, oracle):
    """Exact ties: a UV sphere (every ring of triangles is the same distance from the axis, mirrored rings from the centre) against
    boxes ON its axis, identity rotations (a box with three different sides: its fitted OBB is well defined, unlike a solid of
    revolution's) -- symmetric triangles tie for every bound.  Which of them is reported is decided by
    the order of the walk (the first that attains the minimum; the last that lowered the bound on its visit), so the wave
    continuation (budget 2: every query continues there), the task levels and the walk in one piece must report the same
    triangle and witness -- and the oracle's -- for collide() and distance()."""
    abi, bb, g, wl = pkg.abi, pkg.bvh_builder, pkg.geometry, pkg.workloads
    mesh = bb.Mesh(*bb.uv_sphere(24, 24, 1.0))
    L = g.ShapeLibrary()
    L.add_bvh(0, len(mesh.vertices))
    radii = [0.05, 0.2, 0.35, 0.5]
    for r in radii:
        L.add_box(r, 0.8 * r, 0.6 * r)
    zs = np.linspace(-0.25, 0.25, 11)
    n = len(radii) * len(zs)
    ident = np.tile(np.array([1.0, 0, 0, 0]), (n, 1))
    T1 = np.zeros((n, 3))
    T2 = np.zeros((n, 3))
    T2[:, 2] = np.repeat(zs, len(radii))
    s2 = 1 + np.tile(np.arange(len(radii)), len(zs))
    b = wl.Batch("ties", L, np.zeros(n, dtype=np.int64), s2, ident, T1, ident, T2, "collide")
    b.meshes = [mesh]
    ML = bb.MeshLibrary(b.meshes)
    big = wl.Batch("ties_x", L, np.tile(b.s1, 8), np.tile(b.s2, 8), np.tile(ident, (8, 1)), np.tile(T1, (8, 1)), np.tile(ident, (8, 1)), np.tile(T2, (8, 1)), "collide")
    big.meshes = [mesh]  # (352 queries: the walks are split from 256 on)
    creq = abi.default_collision_request()
    cref, _ = oracle.mixed_collide_batch(big.shapes, big.verts, ML, big.s1, big.s2, big.tf1, big.tf2, creq, max_contacts=10 ** 4, n_threads=8)
    dref = oracle.mixed_distance_batch(big.shapes, big.verts, ML, big.s1, big.s2, big.tf1, big.tf2, None, n_threads=8)
    assert (cref["num_contacts"] == 0).all() and (dref["distance"] > 0.01).all()
    envs = (dict(HFCL_SHAPE_BUDGET0="2", HFCL_SHAPE_DIST_BUDGET="2"), dict(HFCL_SHAPE_COOP="0", HFCL_SHAPE_BUDGET0="2", HFCL_SHAPE_BUDGET="2"),
            dict(HFCL_SHAPE_LEVELS="1", HFCL_SHAPE_DIST_BUDGET="0"))
    for env in envs:
        lib = wl.make_library(pkg, big, options=env)
        try:
            cg = lib.collide(big.s1, big.s2, big.tf1, big.tf2, creq)
            dg = lib.distance(big.s1, big.s2, big.tf1, big.tf2)
        finally:
            lib.close()
        assert np.array_equal(cg["num_contacts"], cref["num_contacts"]), env
        assert np.abs(cg["distance"] - cref["distance"]).max() < 1e-12, env
        for f in ("p1", "p2", "normal"):  # the witness of the bound: the LAST triangle that lowered it on its visit
            assert _same(cg[f], cref[f], 1e-9), (env, f)
        assert np.abs(dg["distance"] - dref["distance"]).max() < 1e-12, env
        assert np.array_equal(dg["b1"], dref["b1"]), env  # the FIRST triangle that attains the minimum
        assert np.abs(dg["p1"] - dref["p1"]).max() < 1e-9 and np.abs(dg["p2"] - dref["p2"]).max() < 1e-9, env


@pytest.mark.gpu
@pytest.mark.parametrize("cached", [False, True])
def test_gpu_mesh_solid_guesses(pkg, oracle, cached):
    """The solver's cached guess through the one-query-per-lane form: a request that reads the guess back (and one whose
    leaves hand it on: CachedGuess) keeps its walks in one piece; decisions and first-contact triangles are the oracle's,
    the returned guess is the oracle's where the walk ended on a leaf result that round-off does not move (a contact)."""
    abi, bb = pkg.abi, pkg.bvh_builder
    b = pkg.workloads.mesh_vs_solid("box,capsule,convex32", n=4000, seed=11)
    ML = bb.MeshLibrary(b.meshes)
    req = abi.default_collision_request()
    if cached:
        req.q.gjk_initial_guess = abi.CachedGuess
        req.q.cached_gjk_guess[:] = [0.3, -0.2, 0.9]
    ref, _, gref = oracle.mixed_collide_batch(b.shapes, b.verts, ML, b.s1, b.s2, b.tf1, b.tf2, req, max_contacts=10 ** 5,
                                              n_threads=16, want_guess=True)
    lib = pkg.workloads.make_library(pkg, b)
    try:
        got, ggot = lib.collide(b.s1, b.s2, b.tf1, b.tf2, req, want_guess=True)
    finally:
        lib.close()
    near = np.abs(ref["distance"]) < 1e-9
    assert ((got["num_contacts"] == ref["num_contacts"]) | near).all()
    m = ~near
    assert np.array_equal(got["b1"][m], ref["b1"][m]) and np.array_equal(got["b2"][m], ref["b2"][m])
    hit = m & (ref["num_contacts"] > 0)
    assert hit.sum() > 300
    assert np.isfinite(ggot["gjk_guess"]).all()
    # (the guess after EPA is -depth * normal: as well conditioned as the contact itself)
    err = np.abs(ggot["gjk_guess"][hit] - gref["gjk_guess"][hit]).max(axis=1)
    assert np.quantile(err, 0.99) < 1e-5, float(np.quantile(err, 0.99))


@pytest.mark.gpu
@pytest.mark.parametrize("kind", ["box", "capsule"])
def test_gpu_mesh_solid_margin_forms_agree(pkg, oracle, kind):
    """A security margin on the one-query-per-lane path (contacts without penetration: no EPA leaf; bounds shifted by the margin):
    wave continuation (budget 8), task levels and walks in one piece against the oracle."""
    abi, bb = pkg.abi, pkg.bvh_builder
    b = pkg.workloads.mesh_vs_solid(kind, n=4000, seed=8)
    ML = bb.MeshLibrary(b.meshes)
    req = abi.default_collision_request()
    req.security_margin = 0.04
    ref, _ = oracle.mixed_collide_batch(b.shapes, b.verts, ML, b.s1, b.s2, b.tf1, b.tf2, req, max_contacts=10 ** 5, n_threads=16)
    near = np.abs(ref["distance"] - 0.04) < 1e-9
    for env in (dict(HFCL_SHAPE_BUDGET0="8"), dict(HFCL_SHAPE_COOP="0", HFCL_SHAPE_BUDGET0="8", HFCL_SHAPE_BUDGET="8"), dict(HFCL_SHAPE_LEVELS="1")):
        got = _device_collide(pkg, b, req, env=env)
        assert ((got["num_contacts"] == ref["num_contacts"]) | near).all(), env
        m = ~near
        assert np.array_equal(got["b1"][m], ref["b1"][m]) and np.array_equal(got["b2"][m], ref["b2"][m]), env
        hit = m & (ref["num_contacts"] > 0)
        assert hit.sum() > 400 and np.abs(got["distance"][hit] - ref["distance"][hit]).max() < 4e-6, env
        if kind == "box":
            free = m & (ref["num_contacts"] == 0) & (np.abs(ref["distance"]) < 1e300)
            assert np.abs(got["distance"][free] - ref["distance"][free]).max() < 4e-6, env


@pytest.mark.gpu
@pytest.mark.parametrize("kind", ["sphere", "box", "capsule", "ellipsoid", "convex32"])
def test_gpu_mesh_solid_long_walks(pkg, oracle, kind):
    """Mesh x solid at the size of cfg4's models (5 000 triangles): walks of thousands of steps, which the one-query-per-lane
    form (k_bvh_collide<SOLID>) cuts into tasks.  (1) Against the oracle: contacts, first-contact triangles, depths.
    (2) The cut is invisible: tiny step budgets (every long query becomes a deep task tree, entries expanded on the way out,
    EPA leaves overtaken by earlier contacts) and no budget at all give the same records.  (3) The 16-lane
    group kernel agrees with both.  (4) The fp32 device path takes the same decisions away from the decision boundary."""
    abi, bb = pkg.abi, pkg.bvh_builder
    b = pkg.workloads.mesh_vs_solid(kind, n=6000, seed=3)
    ML = bb.MeshLibrary(b.meshes)
    req = abi.default_collision_request()
    ref, _ = oracle.mixed_collide_batch(b.shapes, b.verts, ML, b.s1, b.s2, b.tf1, b.tf2, req, max_contacts=10 ** 5, n_threads=16)
    got = _device_collide(pkg, b, req)
    assert not ((got["status"] >> 30) & 1).any()
    near = np.abs(ref["distance"]) < 1e-9
    ok = (got["num_contacts"] == ref["num_contacts"]) | near
    assert ok.all(), int((~ok).sum())
    m = ~near
    assert np.array_equal(got["b1"][m], ref["b1"][m]) and np.array_equal(got["b2"][m], ref["b2"][m])
    hit = m & (ref["num_contacts"] > 0)
    assert 0.1 < hit.mean() < 0.5
    assert np.abs(got["distance"][hit] - ref["distance"][hit]).max() < 4e-6
    free = m & (ref["num_contacts"] == 0) & (np.abs(ref["distance"]) < 1e300)
    if kind in ("box", "ellipsoid", "convex32"):  # (solids of revolution: the fitted OBB's orientation is rounding noise)
        assert np.abs(got["distance"][free] - ref["distance"][free]).max() < 4e-6
        for f in ("p1", "p2", "normal"):
            assert np.array_equal(np.isnan(got[f][free]), np.isnan(ref[f][free])), f
    else:
        assert (got["distance"][free] > 0).all()
    tiny = _device_collide(pkg, b, req, env=dict(HFCL_SHAPE_BUDGET0="8", HFCL_SHAPE_BUDGET="8", HFCL_SHAPE_LEAF_COST="8"))
    whole = _device_collide(pkg, b, req, env=dict(HFCL_SHAPE_LEVELS="1"))
    for other in (tiny, whole):  # (two inlined copies of the box test may contract their FMAs differently: last-bit room)
        assert np.array_equal(other["num_contacts"], got["num_contacts"]) and np.array_equal(other["status"], got["status"])
        assert np.array_equal(other["b1"], got["b1"]) and np.array_equal(other["b2"], got["b2"])
        for f in ("distance", "p1", "p2", "normal"):
            assert _same(other[f], got[f], 1e-12), f
    # (2b) the waves' own long walks cut into chunks for later launches (BvhSplit::cut_ticks; here after 15 000 clock ticks, with
    # the lanes handing over after 8 steps): the same kernel on the same entries -- every field of every record, bit for bit
    cut = _device_collide(pkg, b, req, env=dict(HFCL_SHAPE_CUT_TICKS="15000", HFCL_SHAPE_BUDGET0="8"))
    uncut = _device_collide(pkg, b, req, env=dict(HFCL_SHAPE_BUDGET0="8"))
    for f in cut.dtype.names:
        assert _same(cut[f], uncut[f], 0.0) if cut[f].dtype.kind == "f" else np.array_equal(cut[f], uncut[f]), f
    for f in ("num_contacts", "status", "b1", "b2"):
        assert np.array_equal(cut[f], got[f]), f
    group = _device_collide(pkg, b, req, env=dict(HFCL_BVH_SHAPE_LANE="0"))
    assert np.array_equal(group["num_contacts"][m], got["num_contacts"][m])
    assert np.array_equal(group["b1"], got["b1"]) and np.array_equal(group["b2"], got["b2"])
    fin = np.abs(got["distance"]) < 1e300
    assert np.array_equal(fin, np.abs(group["distance"]) < 1e300)
    cmp = fin if kind in ("box", "ellipsoid", "convex32") else fin & (got["num_contacts"] > 0)  # (round solids: see above)
    assert np.abs(group["distance"][cmp] - got["distance"][cmp]).max() < 4e-6
    g32 = _device_collide(pkg, b, req, f32=True)
    clear = np.abs(ref["distance"]) > 1e-3
    c32 = (g32["status"] >> 7) & 1
    assert np.array_equal(c32[clear] != 0, ref["num_contacts"][clear] > 0)
    assert np.abs(g32["distance"][hit & clear] - ref["distance"][hit & clear]).max() < 2e-4


def test_oracle_mesh_shape_distance_equals_brute_force(pkg, oracle):
    """distance(mesh, shape) = min over all triangles of distance(triangle, shape) (signed: GJK + EPA per
    triangle); the reported b1 is a triangle realising it."""
    abi, g, bb = pkg.abi, pkg.geometry, pkg.bvh_builder
    b = _scene(pkg, n=200, seed=8, half_width=1.0)
    ML = bb.MeshLibrary(b.meshes)
    sel = _mixed_only(pkg, b)[:80]
    req = abi.default_distance_request()
    out = oracle.mixed_distance_batch(b.shapes, b.verts, ML, b.s1[sel], b.s2[sel], b.tf1[sel], b.tf2[sel], req)
    kinds = b.shapes["type"]
    assert 0.1 < (out["distance"] <= 0).mean() < 0.9
    for k, i in enumerate(sel):
        mesh_first = kinds[b.s1[i]] == abi.BV_OBBRSS
        ms, ss = (b.s1[i], b.s2[i]) if mesh_first else (b.s2[i], b.s1[i])
        tfm, tfs = (b.tf1[i], b.tf2[i]) if mesh_first else (b.tf2[i], b.tf1[i])
        mesh = b.meshes[int(b.shapes[ms]["bvh_index"])]
        L = g.ShapeLibrary()
        for t3 in mesh.triangles:
            L.add_triangle(*mesh.vertices[t3])
        src = b.shapes[ss]
        if src["type"] == abi.GEOM_CONVEX:
            base = L.add_convex(b.verts[src["vertex_offset"]:src["vertex_offset"] + src["num_points"]])
        else:
            base = L._add(int(src["type"]), tuple(src["params"]), float(src["swept_sphere_radius"]))
        nt = mesh.num_tris
        # per-triangle signed distance through collide() (the distance matrix has no TriangleP rows)
        r = oracle.collide_batch(L.shapes_array(), L.vertices_array(), np.arange(nt), np.full(nt, base), np.tile(tfm, (nt, 1)),
                                 np.tile(tfs, (nt, 1)), None, n_threads=4)
        if r["distance"].min() > 0:
            assert abs(out["distance"][k] - r["distance"].min()) < 1e-9, (i, out["distance"][k], r["distance"].min())
        else:
            # RSS bounds are clamped at 0 (RSS.cpp:1003), so once a penetrating triangle is found every other
            # subtree "can stop": the reference reports the first penetration met, not the deepest one
            assert r["distance"].min() - 1e-9 <= out["distance"][k] <= 0
        assert abs(r["distance"][out["b1"][k]] - out["distance"][k]) < 1e-9 and out["b2"][k] == -1
        # witness points in the caller's order: |p2 - p1| = |d| and the normal points from o1 to o2
        sep = out["p2"][k] - out["p1"][k]
        assert abs(np.linalg.norm(sep) - abs(out["distance"][k])) < 1e-6


@pytest.mark.parametrize("signed", [True, False])
def test_device_headers_distance_match_oracle(pkg, oracle, hostsim, signed):
    abi, bb = pkg.abi, pkg.bvh_builder
    b = _scene(pkg, n=3000, seed=9, half_width=1.2)
    ML = bb.MeshLibrary(b.meshes)
    sel = _mixed_only(pkg, b)
    req = abi.default_distance_request()
    req.enable_signed_distance = 1 if signed else 0
    a = (b.shapes, b.verts, ML, b.s1[sel], b.s2[sel], b.tf1[sel], b.tf2[sel], req)
    ref, gref = oracle.mixed_distance_batch(*a, want_guess=True, n_threads=4)
    got, ggot = hostsim.mesh_shape_distance_f64(abi, *a, want_guess=True)
    assert 0.1 < (ref["distance"] <= 0).mean() < 0.9
    assert np.array_equal(ref["b1"], got["b1"]) and (got["b2"] == -1).all()
    assert np.array_equal(ref["status"], got["status"])
    assert _same(got["distance"], ref["distance"], 1e-15)
    for f in ("normal", "p1", "p2"):
        assert _same(got[f], ref[f], 1e-11), f
    assert _same(ggot["gjk_guess"], gref["gjk_guess"], 1e-11)


def _check_shape_distance_records(pkg, oracle, ML, b, got, ref, what, req=None, max_ties=0):
    """mesh x solid distance() records of the device against the oracle's.  The kernels are built without contraction
    (hfcl_k_bvhs.o), i.e. with the reference's arithmetic: statuses and distances EQUAL in every record, the triangle id EQUAL
    (max_ties = 0: the default kernels -- DistanceResult::update keeps the first triangle at the minimal distance,
    /root/reference/include/hpp/fcl/collision_data.h:1099-1125, and the pooled continuation hands every walk whose choice could hang
    on a rounding error back to the ordered walk), witness points and normal to 1e-12 (their last step is summed in another order
    than the oracle's).  max_ties > 0 (the wave-per-walk continuation HFCL_SHAPE_DIST_POOL=0 and the 16-lane group kernel only)
    allows that share of records, enumerated here, whose reported triangle is at the oracle's distance bit for bit as well."""
    abi = pkg.abi
    kinds = b.shapes["type"]
    mixed = (kinds[b.s1] == abi.BV_OBBRSS) != (kinds[b.s2] == abi.BV_OBBRSS)
    assert not ((got["status"] >> 30) & 1).any(), what + ": overflow flags"
    assert np.array_equal(got["status"][mixed], ref["status"][mixed]), what + ": statuses"
    assert np.array_equal(got["distance"][mixed], ref["distance"][mixed]), "%s: %d distances differ from the oracle's, max %g" % (
        what, int((got["distance"][mixed] != ref["distance"][mixed]).sum()), np.abs(got["distance"][mixed] - ref["distance"][mixed]).max())
    assert (got["b2"][mixed] == -1).all()
    same = got["b1"] == ref["b1"]
    ties = np.flatnonzero(mixed & ~same)
    allowed = max(1, int(max_ties * mixed.sum())) if max_ties else 0
    assert len(ties) <= allowed, "%s: %d records with another triangle id: %s" % (what, len(ties), ties[:8])
    for k in ties:  # enumerated: the reported triangle is at the oracle's distance, bit for bit
        d = oracle.mixed_leaf_distance(b.shapes, b.verts, ML, b.s1[k], b.s2[k], b.tf1[k], b.tf2[k], got["b1"][k], req)
        assert d == ref["distance"][k], "%s: record %d reports triangle %d at %.17g, the oracle triangle %d at %.17g" % (
            what, k, got["b1"][k], d, ref["b1"][k], ref["distance"][k])
    m = mixed & same
    for f in ("p1", "p2", "normal"):
        assert np.array_equal(np.isnan(got[f][m]), np.isnan(ref[f][m])), what + ": " + f
        assert np.nanmax(np.abs(got[f][m] - ref[f][m]), initial=0.0) < 1e-12, what + ": " + f
    return len(ties)


@pytest.mark.gpu
def test_gpu_mesh_vs_shapes_distance(pkg, oracle):
    """distance() on a scene of meshes and solids of every kind: the mesh x solid records against the oracle's -- statuses, distances
    and triangle ids equal (_check_shape_distance_records); the other pair kinds of the batch as in the parity suite."""
    abi, wl, bb = pkg.abi, pkg.workloads, pkg.bvh_builder
    b = _scene(pkg, n=20000, seed=10, half_width=1.3)
    ML = bb.MeshLibrary(b.meshes)
    req = abi.default_distance_request()
    ref = oracle.mixed_distance_batch(b.shapes, b.verts, ML, b.s1, b.s2, b.tf1, b.tf2, req, n_threads=16)
    lib = wl.make_library(pkg, b)
    try:
        got = lib.distance(b.s1, b.s2, b.tf1, b.tf2, req)
        buckets = lib.last_bucket_counts()
    finally:
        lib.close()
    assert buckets["bvh_shape"] > 15000 and buckets["unsupported"] == 0
    kinds = b.shapes["type"]
    mixed = (kinds[b.s1] == abi.BV_OBBRSS) != (kinds[b.s2] == abi.BV_OBBRSS)
    assert (mixed & (ref["distance"] > 1e-6)).sum() > 5000 and (mixed & (ref["distance"] <= 0)).sum() > 1000
    _check_shape_distance_records(pkg, oracle, ML, b, got, ref, "scene", req)


def _device_distance(pkg, b, req, env=None):
    lib = pkg.workloads.make_library(pkg, b, options=env)
    try:
        return lib.distance(b.s1, b.s2, b.tf1, b.tf2, req)
    finally:
        lib.close()


@pytest.mark.gpu
@pytest.mark.parametrize("kind", ["sphere", "box", "capsule", "ellipsoid", "convex32"])
def test_gpu_mesh_solid_distance_long_walks(pkg, oracle, kind):
    """distance() between cfg4-size models and a solid, every form of the walk against the oracle (equal statuses, distances and
    triangle ids) and against each other (every field of every record, but for the enumerated ties): one query per lane with the
    continuation of long walks pooled four to a wave (default), ordered a wave per walk (HFCL_SHAPE_DIST_POOL=0), with a budget
    of 16 steps (nearly every walk continues there) and of 0 (none does), and the 16-lane group kernel."""
    abi, wl, bb = pkg.abi, pkg.workloads, pkg.bvh_builder
    b = wl.mesh_vs_solid(kind, n=3000, seed=5, half_width=2.0)
    ML = bb.MeshLibrary(b.meshes)
    req = abi.default_distance_request()
    ref = oracle.mixed_distance_batch(b.shapes, b.verts, ML, b.s1, b.s2, b.tf1, b.tf2, req, n_threads=16)
    assert (ref["distance"] > 1e-6).sum() > 1000 and (ref["distance"] <= 0).sum() > 200
    forms = {"default": {}, "ordered": dict(HFCL_SHAPE_DIST_POOL="0"), "pool-16": dict(HFCL_SHAPE_DIST_BUDGET="16"),
             "ordered-16": dict(HFCL_SHAPE_DIST_POOL="0", HFCL_SHAPE_DIST_BUDGET="16"), "lanes-only": dict(HFCL_SHAPE_DIST_BUDGET="0"),
             "group": dict(HFCL_BVH_SHAPE_LANE="0")}
    res = {name: _device_distance(pkg, b, req, env) for name, env in forms.items()}
    ties = {name: _check_shape_distance_records(pkg, oracle, ML, b, r, ref, "%s/%s" % (kind, name), req,
                                                max_ties=0.002 if name in ("ordered", "ordered-16", "group") else 0) for name, r in res.items()}
    base = res["lanes-only"]  # the sequential walk of one lane: the oracle's order of visits
    assert ties["lanes-only"] == 0
    for name, r in res.items():
        differ = np.flatnonzero(r["b1"] != base["b1"])
        assert len(differ) <= ties[name], name
        same = r["b1"] == base["b1"]
        for f in r.dtype.names:
            if r[f].dtype.kind == "f":
                assert np.array_equal(np.nan_to_num(r[f][same], nan=-7.0), np.nan_to_num(base[f][same], nan=-7.0)), (name, f)
            elif f != "b1":
                assert np.array_equal(r[f], base[f]), (name, f)


@pytest.mark.gpu
def test_gpu_mesh_solid_distance_at_baseline_size(pkg, oracle):
    """100 000 distance() queries between 5 000-triangle models and solids of six kinds (the cfg4s scene): every record against the oracle."""
    abi, wl, bb = pkg.abi, pkg.workloads, pkg.bvh_builder
    b = wl.mesh_vs_solid("mixed", n=100000, seed=2)
    ML = bb.MeshLibrary(b.meshes)
    req = abi.default_distance_request()
    import os
    ref = oracle.mixed_distance_batch(b.shapes, b.verts, ML, b.s1, b.s2, b.tf1, b.tf2, req, n_threads=min(128, os.cpu_count() or 16))
    got = _device_distance(pkg, b, req)
    _check_shape_distance_records(pkg, oracle, ML, b, got, ref, "100k", req)


@pytest.mark.gpu
def test_gpu_mesh_solid_collide_at_baseline_size(pkg, oracle):
    """cfg4s as bench.py runs it -- 100 000 collide() queries, mixed solids, default request -- against the oracle: contact flags and
    first-contact triangle ids equal (outside a 1e-9 band around touching), penetration depths to the solver tolerance."""
    abi, wl, bb = pkg.abi, pkg.workloads, pkg.bvh_builder
    b = wl.mesh_vs_solid("mixed", n=100000, seed=1)
    ML = bb.MeshLibrary(b.meshes)
    req = wl.make_request(b, abi)
    import os
    ref, _ = oracle.mixed_collide_batch(b.shapes, b.verts, ML, b.s1, b.s2, b.tf1, b.tf2, req, max_contacts=10 ** 6,
                                        n_threads=min(128, os.cpu_count() or 16))
    got = _device_collide(pkg, b, req)
    assert not ((got["status"] >> 30) & 1).any()
    near = np.abs(ref["distance"]) < 1e-9
    assert ((got["num_contacts"] == ref["num_contacts"]) | near).all()
    m = ~near
    assert np.array_equal(got["b1"][m], ref["b1"][m]) and np.array_equal(got["b2"][m], ref["b2"][m])
    hit = m & (ref["num_contacts"] > 0)
    assert 0.05 < hit.mean() < 0.6
    assert np.abs(got["distance"][hit] - ref["distance"][hit]).max() < 4e-6
    # the cut form of the continuation (default for mesh x solid) against the uncut one: every field of every record
    uncut = _device_collide(pkg, b, req, env=dict(HFCL_SHAPE_CUT_TICKS="0"))
    for f in got.dtype.names:
        assert _same(got[f], uncut[f], 0.0) if got[f].dtype.kind == "f" else np.array_equal(got[f], uncut[f]), f
    # the EPA leaves (k_bvh_shape_finish): two tiers with the whole walks' items on the helper stream beside the chunk launches (default)
    # against one launch at full capacity behind the last launch of the walk, and against each switch alone
    for env in (dict(HFCL_SHAPE_FINISH_TIERS="0", HFCL_SHAPE_FINISH_ASIDE="0"), dict(HFCL_SHAPE_FINISH_TIERS="0"), dict(HFCL_SHAPE_FINISH_ASIDE="0")):
        other = _device_collide(pkg, b, req, env=env)
        for f in got.dtype.names:
            assert _same(got[f], other[f], 0.0) if got[f].dtype.kind == "f" else np.array_equal(got[f], other[f]), (f, env)
    # the queries' own phase: walk / leaves / resolve (default from 65 536 queries per batch) against k_bvh_collide's SOLID form and against the
    # listed leaves evaluated as listed; forced on a batch below its threshold as well
    for env in (dict(HFCL_SHAPE_WALK="0"), dict(HFCL_SHAPE_WALK_SORT="0")):
        other = _device_collide(pkg, b, req, env=env)
        assert got.tobytes() == other.tobytes(), env
    small = wl.mesh_vs_solid("mixed", n=3000, seed=5)
    assert _device_collide(pkg, small, req, env=dict(HFCL_SHAPE_WALK_MIN="0")).tobytes() == _device_collide(pkg, small, req).tobytes()


def test_mesh_vs_flats_headers_match_oracle(pkg, oracle, hostsim):
    """Meshes against Plane / Halfspace: the flats' BVs are unbounded (geometric_shapes_utility.cpp:545-581,
    803-850), so the reference tests every triangle; contacts = triangles reaching into the halfspace."""
    abi, bb = pkg.abi, pkg.bvh_builder
    b = _scene(pkg, n=400, seed=11, flats_only=True)
    ML = bb.MeshLibrary(b.meshes)
    sel = _mixed_only(pkg, b)
    req = abi.default_collision_request()
    req.num_max_contacts = 10 ** 6
    a = (b.shapes, b.verts, ML, b.s1[sel], b.s2[sel], b.tf1[sel], b.tf2[sel], req)
    ref, cref = oracle.mixed_collide_batch(*a, max_contacts=10 ** 6)
    got, cgot = hostsim.mesh_shape_collide_f64(abi, *a, max_contacts=10 ** 6)
    assert (ref["num_contacts"] > 0).mean() > 0.5 and np.array_equal(ref["num_contacts"], got["num_contacts"])
    assert np.array_equal(cref["b1"], cgot["b1"]) and np.array_equal(cref["b2"], cgot["b2"])
    assert _same(got["distance"], ref["distance"], 1e-15) and _same(cgot["penetration_depth"], cref["penetration_depth"], 1e-15)
    # brute force for halfspaces: a triangle is in contact iff its lowest vertex along the normal is below the plane
    kinds = b.shapes["type"]
    checked = 0
    for k, i in enumerate(sel[:40]):
        mesh_first = kinds[b.s1[i]] == abi.BV_OBBRSS
        fs = b.s2[i] if mesh_first else b.s1[i]
        if kinds[fs] != abi.GEOM_HALFSPACE:
            continue
        tfm, tff = (b.tf1[i], b.tf2[i]) if mesh_first else (b.tf2[i], b.tf1[i])
        mesh = b.meshes[int(b.shapes[b.s1[i] if mesh_first else b.s2[i]]["bvh_index"])]
        R, T = pkg.geometry.pose_R(tfm), np.asarray(tfm)[9:]
        W = mesh.vertices @ R.T + T
        Rf, Tf = pkg.geometry.pose_R(tff), np.asarray(tff)[9:]
        nw = Rf @ b.shapes[fs]["params"][:3]
        dw = b.shapes[fs]["params"][3] + nw @ Tf
        sd = W @ nw - dw
        brute = int((sd[mesh.triangles].min(axis=1) <= 0).sum())
        assert ref["num_contacts"][k] == brute
        checked += 1
    assert checked > 5
    d_ref = oracle.mixed_distance_batch(b.shapes, b.verts, ML, b.s1[sel], b.s2[sel], b.tf1[sel], b.tf2[sel], None)
    d_got = hostsim.mesh_shape_distance_f64(abi, b.shapes, b.verts, ML, b.s1[sel], b.s2[sel], b.tf1[sel], b.tf2[sel],
                                            abi.default_distance_request())
    assert _same(d_got["distance"], d_ref["distance"], 1e-15) and np.array_equal(d_got["b1"], d_ref["b1"])


@pytest.mark.gpu
def test_gpu_mesh_vs_flats(pkg, oracle):
    abi, wl, bb = pkg.abi, pkg.workloads, pkg.bvh_builder
    b = _scene(pkg, n=3000, seed=12, flats_only=True)
    ML = bb.MeshLibrary(b.meshes)
    req = abi.default_collision_request()
    ref = oracle.mixed_collide_batch(b.shapes, b.verts, ML, b.s1, b.s2, b.tf1, b.tf2, req, n_threads=16)
    dref = oracle.mixed_distance_batch(b.shapes, b.verts, ML, b.s1, b.s2, b.tf1, b.tf2, None, n_threads=16)
    lib = wl.make_library(pkg, b)
    try:
        got = lib.collide(b.s1, b.s2, b.tf1, b.tf2, req)
        dgot = lib.distance(b.s1, b.s2, b.tf1, b.tf2, abi.default_distance_request())
    finally:
        lib.close()
    kinds = b.shapes["type"]
    mixed = (kinds[b.s1] == abi.BV_OBBRSS) != (kinds[b.s2] == abi.BV_OBBRSS)
    near = np.abs(ref["distance"]) < 1e-9
    m = mixed & ~near
    assert np.array_equal(got["num_contacts"][m], ref["num_contacts"][m])
    assert np.array_equal(got["b1"][m], ref["b1"][m]) and np.array_equal(got["b2"][m], ref["b2"][m])
    assert np.abs(got["distance"][m] - ref["distance"][m]).max() < 1e-9
    # distance(): traversal order among equal RSS bounds (all 0 / NaN here) decides which penetrating triangle is met first
    sepd = mixed & (dref["distance"] > 1e-9)
    assert np.abs(dgot["distance"][sepd] - dref["distance"][sepd]).max() < 1e-9
    assert (dgot["distance"][mixed & (dref["distance"] <= 0)] <= 1e-9).all()


@pytest.mark.gpu
def test_gpu_mixed_scene_beside_equals_in_line(pkg, oracle):
    """A batch with solid x solid, mesh x solid (both operand orders) and mesh x mesh pairs on cfg4-size models (workloads.mixed_scene, bench.py's
    cfgmix row): the mesh walks on streams of their own beside the solids' kernels (option mesh_beside: 2, the default, also runs the mesh x mesh
    walks beside the mesh x solid walks on tables of their own and the solids' EPA section beside both -- when the library's batch before held
    both kinds; 4 whatever it held; 1 only forks the mesh walks) give byte for
    byte the records of the in-line order, twice, and all equal the oracle's decisions (contact flags, first-contact triangle ids); distance()
    over the same batch likewise."""
    abi, wl, bb = pkg.abi, pkg.workloads, pkg.bvh_builder
    b = wl.mixed_scene(n=40_000, seed=3)
    assert min(b.mix.values()) > 0.1
    req = abi.default_collision_request()
    dreq = abi.default_distance_request()
    recs, drecs = {}, {}
    for beside in (4, 2, 1, 0):
        lib = wl.make_library(pkg, b, options={"mesh_beside": beside})
        try:
            first = lib.collide(b.s1, b.s2, b.tf1, b.tf2, req)
            again = lib.collide(b.s1, b.s2, b.tf1, b.tf2, req)
            assert first.tobytes() == again.tobytes()
            recs[beside] = first
            drecs[beside] = lib.distance(b.s1[:8000], b.s2[:8000], b.tf1[:8000], b.tf2[:8000], dreq)
        finally:
            lib.close()
    assert all(recs[k].tobytes() == recs[0].tobytes() for k in (4, 2, 1))
    assert all(drecs[k].tobytes() == drecs[0].tobytes() for k in (4, 2, 1))
    ML = bb.MeshLibrary(b.meshes)
    ref = oracle.mixed_collide_batch(b.shapes, b.verts, ML, b.s1, b.s2, b.tf1, b.tf2, req, n_threads=32)
    got = recs[2]
    assert not np.any((got["status"] >> 31) & 1)
    clear = np.abs(ref["distance"]) > 1e-9
    assert np.array_equal((got["num_contacts"] > 0)[clear], (ref["num_contacts"] > 0)[clear])
    mesh_pair = (b.shapes["type"][b.s1] == abi.BV_OBBRSS) | (b.shapes["type"][b.s2] == abi.BV_OBBRSS)
    hit = clear & mesh_pair & (ref["num_contacts"] > 0)
    assert np.array_equal(got["b1"][hit], ref["b1"][hit]) and np.array_equal(got["b2"][hit], ref["b2"][hit])
    assert 0.1 < (ref["num_contacts"] > 0).mean() < 0.9
