"""GPU parity tests: the HIP path, called through the C ABI, against the fp64 CPU oracle on the
same seeded inputs; plus size-independent properties at BASELINE.json's full sizes."""
import os

import numpy as np
import pytest

from compare import check_parity, check_properties

pytestmark = pytest.mark.gpu


def _oracle(oracle, b, req, tf1=None, tf2=None):
    fn = oracle.distance_batch if b.kind == "distance" else oracle.collide_batch
    return fn(b.shapes, b.verts, b.s1, b.s2, b.tf1 if tf1 is None else tf1, b.tf2 if tf2 is None else tf2, req,
              n_threads=min(64, max(8, os.cpu_count() or 8)))


def _engine(pkg, b, req, options=None):
    lib = pkg.Library(b.lib, device=0, options=options)
    try:
        if b.kind == "distance":
            return lib.distance(b.s1, b.s2, b.tf1, b.tf2, req), lib.last_bucket_counts()
        return lib.collide(b.s1, b.s2, b.tf1, b.tf2, req), lib.last_bucket_counts()
    finally:
        lib.close()


def _check_exact(abi, got, ref, name):
    """The fp64 GJK / EPA kernels are built without contraction (hfcl_k_gjk64.o / hfcl_k_epa64.o): every integer output -- GJK and EPA
    status, both iteration counts, the contact flag, the contact count -- and every distance equal the oracle's, record for record."""
    assert np.array_equal(got["status"], ref["status"]), "%s: %d status words differ" % (name, int((got["status"] != ref["status"]).sum()))
    assert np.array_equal(got["num_contacts"], ref["num_contacts"]), name
    d_eq = (got["distance"] == ref["distance"]) | (np.isnan(got["distance"]) & np.isnan(ref["distance"]))
    assert d_eq.all(), "%s: %d distances differ in their last bits, max %g" % (
        name, int((~d_eq).sum()), np.nanmax(np.abs(got["distance"][~d_eq] - ref["distance"][~d_eq])))


def test_native_library_is_loaded(pkg):
    assert pkg.engine.device_count() >= 1
    assert pkg.engine.dll().hfcl_abi_version() == 5


@pytest.mark.parametrize("case,n", [("cfg1_sphere_sphere", 1000), ("cfg2_box_capsule", 100000),
                                    ("cfg3_convex_convex", 100000), ("cfg5_mixed", 100000),
                                    ("all_primitives", 100000)])
def test_fp64_parity(pkg, oracle, case, n):
    """fp64 kernels vs oracle: statuses, iteration counts, flags and distances EQUAL in every record (no decision band: the kernels
    do the reference's arithmetic, no contraction), witness points and normals to 1e-9 (their last step -- closest_points, the
    transform to the world frame -- is summed in another order than the oracle's)."""
    abi, wl = pkg.abi, pkg.workloads
    b = getattr(wl, case)(n=n)
    req = wl.make_request(b, abi)
    ref = _oracle(oracle, b, req)
    got, buckets = _engine(pkg, b, req)
    _check_exact(abi, got, ref, case)
    check_parity(abi, got, ref, dist_tol=1e-15, point_tol=1e-9, flag_band=0.0, name=case)
    check_properties(abi, got, tol=1e-6, name=case)
    assert buckets["unsupported"] == 0


def test_epa_hand_over_equals_restart(pkg):
    """A polytope that outgrows the 20-iteration block is continued by the full-capacity kernel from the
    saved block; when the save area is full it is redone from its seed.  Both must give the same record
    bit for bit (the first iterations are the reference's in either tier)."""
    abi, wl = pkg.abi, pkg.workloads
    b = wl.cfg5_mixed(n=60000, seed=11)
    req = wl.make_request(b, abi)
    got, buckets = _engine(pkg, b, req)
    assert buckets["epa_overflow"] > 500, buckets  # the workload does exercise the hand-over
    redo, buckets2 = _engine(pkg, b, req, options={"epa_resume_slots": 64})  # nearly every hand-over now falls back to the seed
    assert buckets2["epa_overflow"] == buckets["epa_overflow"]
    assert got.tobytes() == redo.tobytes()


@pytest.mark.parametrize("variant", [0, 1, 2])
def test_fp64_gjk_variants(pkg, oracle, variant):
    abi, wl = pkg.abi, pkg.workloads
    b = wl.cfg5_mixed(n=20000, seed=7)
    b.kind = "distance"
    req = abi.default_distance_request()
    req.q.gjk_variant = variant
    ref = _oracle(oracle, b, req)
    got, _ = _engine(pkg, b, req)
    _check_exact(abi, got, ref, "variant%d" % variant)
    check_parity(abi, got, ref, dist_tol=1e-15, point_tol=1e-9, flag_band=0.0, name="variant%d" % variant)


@pytest.mark.parametrize("crit", [(0, 0), (1, 0), (1, 1), (2, 0), (2, 1)])
@pytest.mark.parametrize("variant", [0, 2])
def test_fp64_convergence_criteria(pkg, oracle, variant, crit):
    """GJKConvergenceCriterion {Default, DualityGap, Hybrid} x {Relative, Absolute} (gjk.cpp:372-425) on the device,
    with and without Nesterov acceleration: statuses and iteration counts are the oracle's."""
    abi, wl = pkg.abi, pkg.workloads
    b = wl.cfg5_mixed(n=20000, seed=12)
    b.kind = "distance"
    req = abi.default_distance_request()
    req.q.gjk_variant = variant
    req.q.gjk_convergence_criterion, req.q.gjk_convergence_criterion_type = crit
    ref = _oracle(oracle, b, req)
    got, _ = _engine(pkg, b, req)
    name = "crit%d%d-v%d" % (crit + (variant,))
    # (The *relative* duality-gap and hybrid tests ask diff <= tol^2 = 1e-12 of a difference of nearly equal numbers, gjk.cpp:405-422: on
    # an Ellipsoid round-off decides GJK's last iterations.  With the kernels' arithmetic equal to the oracle's that is no exception any more.)
    _check_exact(abi, got, ref, name)
    check_parity(abi, got, ref, dist_tol=1e-15, point_tol=1e-9, flag_band=0.0, name=name)


@pytest.mark.parametrize("case", ["cfg5_mixed", "cfg3_convex_convex"])
def test_fp64_bounding_volume_guess(pkg, oracle, case):
    """GJKInitialGuess::BoundingVolumeGuess on the device: same statuses / iteration counts as the oracle."""
    abi, wl = pkg.abi, pkg.workloads
    b = getattr(wl, case)(n=50000, seed=6)
    req = wl.make_request(b, abi)
    req.q.gjk_initial_guess = abi.BoundingVolumeGuess
    ref = _oracle(oracle, b, req)
    got, _ = _engine(pkg, b, req)
    _check_exact(abi, got, ref, case + "-bvguess")
    check_parity(abi, got, ref, dist_tol=1e-15, point_tol=1e-9, flag_band=0.0, name=case + "-bvguess")


def test_fp64_collide_options(pkg, oracle):
    abi, wl = pkg.abi, pkg.workloads
    b = wl.cfg5_mixed(n=20000, seed=8)
    for margin, dub, contact in [(0.05, 0.1, 1), (-0.02, 1e300, 1), (0.0, 0.0, 0), (0.0, 0.3, 1)]:
        req = abi.default_collision_request()
        req.security_margin, req.distance_upper_bound, req.enable_contact = margin, dub, contact
        ref = _oracle(oracle, b, req)
        got, _ = _engine(pkg, b, req)
        _check_exact(abi, got, ref, "opts%s" % ((margin, dub),))
        check_parity(abi, got, ref, dist_tol=1e-15, point_tol=1e-9, flag_band=0.0, name="opts%s" % ((margin, dub),))
        assert np.array_equal(got["num_contacts"] > 0, abi.status_contact(got["status"]) > 0)
    req = abi.default_collision_request()
    req.security_margin = -np.inf
    got, _ = _engine(pkg, b, req)
    assert np.all(abi.status_skipped(got["status"]) == 1) and np.all(got["num_contacts"] == 0)
    req = abi.default_collision_request()
    req.num_max_contacts = 0
    with pytest.raises(pkg.EngineError) as e:
        _engine(pkg, b, req)
    assert e.value.code == abi.ERR_INVALID_ARGUMENT


def test_fp64_warm_start_guess(pkg, oracle):
    abi, wl = pkg.abi, pkg.workloads
    b = wl.cfg3_convex_convex(n=20000, seed=9)
    req = abi.default_distance_request()
    lib = pkg.Library(b.lib)
    got, g = lib.distance(b.s1, b.s2, b.tf1, b.tf2, req, want_guess=True)
    ref, g_ref = oracle.distance_batch(b.shapes, b.verts, b.s1, b.s2, b.tf1, b.tf2, req, want_guess=True)
    assert np.allclose(g["gjk_guess"], g_ref["gjk_guess"], atol=1e-6, equal_nan=True)
    req.q.gjk_initial_guess = abi.CachedGuess
    # same guess to both sides.  A warm-started GJK stops as soon as the distance is within
    # tolerance; the direction of the separation vector is then only accurate to ~sqrt(tol)
    # (the oracle itself moves by 2e-3 under a 1e-12 perturbation of the guess), hence point_tol.
    got2 = lib.distance(b.s1, b.s2, b.tf1, b.tf2, req, guess_in=g_ref)
    ref2 = oracle.distance_batch(b.shapes, b.verts, b.s1, b.s2, b.tf1, b.tf2, req, guess_in=g_ref)
    check_parity(abi, got2, ref2, dist_tol=4e-6, point_tol=5e-3, flag_band=1e-5, name="warm")
    sep = ref["distance"] > 1e-3
    assert abi.status_gjk_iters(got2["status"])[sep].mean() < abi.status_gjk_iters(got["status"])[sep].mean()
    lib.close()


def test_edge_cases(pkg, oracle):
    """empty batch, single pair, ragged hull sizes (4..32 vertices), identical poses, touching shapes."""
    abi, wl, g = pkg.abi, pkg.workloads, pkg.geometry
    L = pkg.ShapeLibrary()
    rng = np.random.default_rng(5)
    ids = []
    for nv in [4, 5, 7, 12, 13, 31, 32]:
        ids.append(L.add_convex(wl.fibonacci_sphere(nv) * rng.uniform(0.2, 1.0, 3)))
    box, sph, cap, ell = L.add_box(1, 1, 1), L.add_sphere(0.5), L.add_capsule(0.3, 1.0), L.add_ellipsoid(0.5, 0.3, 0.4)
    ids += [box, sph, cap, ell]
    lib = pkg.Library(L)
    out = lib.distance(np.zeros(0, np.uint32), np.zeros(0, np.uint32), np.zeros((0, 12)), np.zeros((0, 12)))
    assert len(out) == 0
    n = 5000
    s1, s2 = rng.choice(ids, n), rng.choice(ids, n)
    q = rng.normal(size=(n, 4))
    q /= np.linalg.norm(q, axis=1, keepdims=True)
    tf1 = g.make_pose(quat=q, T=rng.uniform(-1, 1, (n, 3)))
    tf2 = g.make_pose(quat=q[::-1], T=rng.uniform(-1, 1, (n, 3)))
    tf2[:50] = tf1[:50]  # coincident poses
    tf1[50:60] = g.make_pose()
    tf2[50:60] = g.make_pose(T=[1.0, 0, 0])  # axis-aligned, touching for box-box
    s1[50:60] = box
    s2[50:60] = box
    for kind in ("distance", "collide"):
        if kind == "distance":
            got = lib.distance(s1, s2, tf1, tf2)
            ref = oracle.distance_batch(L.shapes_array(), L.vertices_array(), s1, s2, tf1, tf2)
        else:
            got = lib.collide(s1, s2, tf1, tf2)
            ref = oracle.collide_batch(L.shapes_array(), L.vertices_array(), s1, s2, tf1, tf2)
        # Every record matches with no allowance, except one documented class: a hull against ITSELF at the same pose.
        # Its Minkowski difference is centrally symmetric, so the closest faces come in opposite pairs at exactly the
        # same distance and which of the two EPA ends on is decided by last-bit rounding in GJK's first steps (the GPU
        # contracts a*b+c, the oracle does not).  There the depth must agree and the separation vectors must be equal
        # or opposite; nothing else is excused.
        twin = (s1 == s2) & (np.abs(tf1 - tf2).max(axis=1) == 0) & (L.shapes_array()["type"][s1] == abi.GEOM_CONVEX)
        assert 0 < twin.sum() < 20
        check_parity(abi, got[~twin], ref[~twin], dist_tol=1e-6, point_tol=1e-5, flag_band=1e-6, name="edge-" + kind)
        gt, rt = got[twin], ref[twin]
        assert np.array_equal(abi.status_contact(gt["status"]), abi.status_contact(rt["status"]))
        assert np.abs(gt["distance"] - rt["distance"]).max() < 1e-9
        sg, sr = gt["p2"] - gt["p1"], rt["p2"] - rt["p1"]
        assert np.all(np.minimum(np.abs(sg - sr).max(axis=1), np.abs(sg + sr).max(axis=1)) < 1e-5)
        # a batch of one pair gives, byte for byte, the record that pair has in the big batch (first, a middle and the last pair)
        run1 = lib.distance if kind == "distance" else lib.collide
        for k in (0, 55, 2500, n - 1):
            one = run1(s1[k:k + 1], s2[k:k + 1], tf1[k:k + 1], tf2[k:k + 1])
            assert one.tobytes() == got[k:k + 1].tobytes(), (kind, k)
    # unsupported pair kinds are reported, not silently computed
    Lb = pkg.ShapeLibrary()
    t = Lb.add_triangle([0, 0, 0], [1, 0, 0], [0, 1, 0])
    b2 = Lb.add_bvh(0)  # (TriangleP, BVHModel) is not in the reference's function matrices either
    lib2 = pkg.Library(Lb)
    with pytest.raises(pkg.EngineError) as e:
        lib2.distance([t], [b2], [g.make_pose()], [g.make_pose()])
    assert e.value.code == abi.ERR_UNSUPPORTED_PAIR
    lib.close()
    lib2.close()


@pytest.mark.parametrize("case", ["cfg2_box_capsule", "cfg3_convex_convex", "cfg3_unique_hulls"])
def test_fp32_device_path(pkg, oracle, torch_cuda, case):
    """fp32 device-resident path (7-float poses, 44-byte records) vs the fp64 oracle fed with the
    fp32-rounded poses.  Envelope |dd| <= 1e-4*(1+|d|); flags may differ only if |d| <= 1e-4.  cfg3_convex_convex is the
    headline configuration (BASELINE.json configs[2]) at its size: 1 000 000 pairs, every record against the oracle."""
    torch = torch_cuda
    abi, wl = pkg.abi, pkg.workloads
    b = getattr(wl, case)(n={"cfg3_unique_hulls": 100000, "cfg3_convex_convex": 1000000}.get(case, 200000))
    req = wl.make_request(b, abi)
    tf1, tf2 = b.tf_from_f32()
    ref = _oracle(oracle, b, req, tf1, tf2)
    lib = pkg.Library(b.lib)
    dev = torch.device("cuda:0")
    d_s1 = torch.from_numpy(b.s1.astype(np.int32)).to(dev)
    d_s2 = torch.from_numpy(b.s2.astype(np.int32)).to(dev)
    d_p1 = torch.from_numpy(b.pose1_f32).to(dev)
    d_p2 = torch.from_numpy(b.pose2_f32).to(dev)
    d_out = torch.zeros(len(b) * 11, dtype=torch.int32, device=dev)
    fn = lib.distance_device_f32 if b.kind == "distance" else lib.collide_device_f32
    fn(d_s1, d_s2, d_p1, d_p2, len(b), req, d_out, stream=torch.cuda.current_stream().cuda_stream)
    torch.cuda.synchronize()
    got = d_out.cpu().numpy().view(abi.RESULT_F32_DTYPE)
    # No blanket allowance.  One class of records is excused, enumerated and checked for its stated reason: the separation
    # DIRECTION differs from the oracle's by more than the envelope while depth / distance and flags are inside it -- a polytope
    # with two near-equidistant closest faces where fp32 EPA ends on the other one, or (one separated pair per million, met at
    # BASELINE size) a GJK run whose fp32 iterates stop on another point of a nearly flat minimum: the solvers' stopping rules
    # bound the distance to `tol`, the direction only to ~sqrt(2 tol).  Such a record must still be RIGHT: along the GPU's
    # normal the two shapes, evaluated in fp64 from the shape table, overlap by the oracle's depth (penetration) or are apart by
    # the oracle's distance (separation), within the envelope -- i.e. the direction realises the same signed distance.
    # At most 4 such records in a batch (the 2M-pair soak finds 1 per million).
    bad = check_parity(abi, got, ref, dist_tol=1e-4, point_tol=5e-4, flag_band=1e-4, name=case + "-f32", fp32=True,
                       collect_only=True)
    excused = bad["sep_bad_mask"] & ~bad["flag_bad_mask"] & ~bad["dist_bad_mask"] & ~bad["nan_bad_mask"]
    assert excused.sum() <= 4, "%s: %d records outside the fp32 envelope: %s" % (case, excused.sum(), np.flatnonzero(excused)[:20])
    for k in np.flatnonzero(excused):
        nrm = got["normal"][k].astype(np.float64)
        ext = _overlap_extent(abi, b, k, tf1[k], tf2[k], nrm / np.linalg.norm(nrm))  # > 0: overlap along the normal, < 0: a gap
        d = float(ref["distance"][k])  # signed: -depth for a penetrating pair
        assert abs(ext + d) <= 1e-4 * (1 + abs(d)), "record %d: the GPU's normal does not realise the oracle's signed distance (%g vs %g)" % (k, -ext, d)
    check_parity(abi, got[~excused], ref[~excused], dist_tol=1e-4, point_tol=5e-4, flag_band=1e-4, name=case + "-f32", fp32=True)
    check_properties(abi, got, tol=2e-4, name=case + "-f32")
    lib.close()


@pytest.mark.parametrize("case,n", [("cfg3_convex_convex", 300001), ("cfg2_box_capsule", 70000), ("cfg5_mixed", 257), ("cfg3_convex_convex", 1)])
def test_fp32_host_buffers_equal_device_path(pkg, torch_cuda, case, n):
    """hfcl_collide_batch_f32 / hfcl_distance_batch_f32 (host arrays through the chunked pipeline; 300 001 pairs = several chunks of ramping
    size, 257 and 1 = one chunk) against the device-resident fp32 call on the same batch: every record byte for byte."""
    torch = torch_cuda
    abi, wl = pkg.abi, pkg.workloads
    b = getattr(wl, case)(n=n, seed=11)
    req = wl.make_request(b, abi)
    lib = pkg.Library(b.lib)
    host = (lib.distance_f32 if b.kind == "distance" else lib.collide_f32)(b.s1, b.s2, b.pose1_f32, b.pose2_f32, req)
    dev = torch.device("cuda:0")
    d_s1 = torch.from_numpy(b.s1.astype(np.int32)).to(dev)
    d_s2 = torch.from_numpy(b.s2.astype(np.int32)).to(dev)
    d_p1 = torch.from_numpy(b.pose1_f32).to(dev)
    d_p2 = torch.from_numpy(b.pose2_f32).to(dev)
    d_out = torch.zeros(len(b) * 11, dtype=torch.int32, device=dev)
    fn = lib.distance_device_f32 if b.kind == "distance" else lib.collide_device_f32
    fn(d_s1, d_s2, d_p1, d_p2, len(b), req, d_out, stream=torch.cuda.current_stream().cuda_stream)
    torch.cuda.synchronize()
    want = d_out.cpu().numpy().reshape(len(b), 11)
    got = host.view(np.int32).reshape(len(b), 11)
    differ = np.flatnonzero((got != want).any(axis=1))
    assert differ.size == 0, "%d records differ, first %s" % (differ.size, differ[:10])
    assert lib.last_bucket_counts()  # (the host call's populations are those of the whole batch)
    lib.close()


@pytest.mark.parametrize("case,n", [("cfg3_convex_convex", 300000), ("cfg3_unique_hulls", 60000), ("cfg5_mixed", 120000)])
def test_fp32_staged_epa_equals_one_kernel_form(pkg, torch_cuda, case, n):
    """The convex x convex EPA fast tier in three stages (k_epa_prepare: one lane per polytope builds the first tetrahedron;
    k_epa_loop: the expansion loop; k_epa_records: one lane per polytope writes the record; k_epa_resume_cc continues the
    polytopes that outgrow the block) against the one-kernel streaming form on the same batch: every record byte for byte
    (the same arithmetic in other kernels).  cfg5_mixed: convex x convex pairs beside the general queue of the same batch."""
    torch = torch_cuda
    abi, wl = pkg.abi, pkg.workloads
    b = getattr(wl, case)(n=n, seed=3)
    req = wl.make_request(b, abi)
    dev = torch.device("cuda:0")
    d_s1 = torch.from_numpy(b.s1.astype(np.int32)).to(dev)
    d_s2 = torch.from_numpy(b.s2.astype(np.int32)).to(dev)
    d_p1 = torch.from_numpy(b.pose1_f32).to(dev)
    d_p2 = torch.from_numpy(b.pose2_f32).to(dev)
    recs, queues = {}, {}
    for staged in ("0", "1"):
        lib = pkg.Library(b.lib, options={"epa_cc_staged_min": 0, "epa_cc_staged": staged})
        d_out = torch.zeros(len(b) * 11, dtype=torch.int32, device=dev)
        fn = lib.distance_device_f32 if b.kind == "distance" else lib.collide_device_f32
        for _ in range(2):  # (the second call runs on a warm workspace)
            fn(d_s1, d_s2, d_p1, d_p2, len(b), req, d_out, stream=torch.cuda.current_stream().cuda_stream)
            torch.cuda.synchronize()
        recs[staged] = d_out.cpu().numpy().reshape(len(b), 11).copy()
        queues[staged] = lib.last_bucket_counts()
        names = [k for k, _ in lib.last_kernel_breakdown()]
        assert ("k_epa_prepare" in names) == (staged == "1"), names
        lib.close()
    assert queues["0"]["epa_queue"] == queues["1"]["epa_queue"] > 0
    differ = np.flatnonzero((recs["0"] != recs["1"]).any(axis=1))
    assert differ.size == 0, "%d records differ, first %s" % (differ.size, differ[:10])


@pytest.mark.parametrize("case,n,precision", [("cfg5_mixed", 120000, "f64"), ("all_primitives", 60000, "f64"), ("cfg2_box_capsule", 120000, "f64"),
                                              ("cfg5_mixed", 120000, "f32"), ("cfg2_box_capsule", 120000, "f32")])
def test_general_staged_epa_equals_one_kernel_form(pkg, torch_cuda, case, n, precision):
    """HFCL_EPA_GENERAL_STAGED=1 (k_epa_prepare_general / k_epa_loop_general / k_epa_records_general for the general EPA queues; off by
    default, profiles/r05_e_general_staged.md) against the default one-kernel forms on the same batch: records and cached guesses byte for
    byte."""
    if not pkg.engine.has_ab_forms():
        pytest.skip("the general staged EPA lost its A/B (profiles/r05_e) and is not in the product build: tools/build_variant.sh ab host,k_bvh,k_epa "
                    "-DHFCL_KEEP_AB_FORMS=1, then HFCL_LIB_PATH=build/ab/lib_ab.so")
    torch = torch_cuda
    abi, wl = pkg.abi, pkg.workloads
    b = getattr(wl, case)(n=n, seed=5)
    req = wl.make_request(b, abi)
    dev = torch.device("cuda:0")
    d_s1 = torch.from_numpy(b.s1.astype(np.int32)).to(dev)
    d_s2 = torch.from_numpy(b.s2.astype(np.int32)).to(dev)
    f32 = precision == "f32"
    d_p1 = torch.from_numpy(b.pose1_f32 if f32 else b.tf1).to(dev)
    d_p2 = torch.from_numpy(b.pose2_f32 if f32 else b.tf2).to(dev)
    words = 11 if f32 else 24
    recs = {}
    for staged in ("0", "1"):
        lib = pkg.Library(b.lib, options={"epa_general_staged_min": 0, "epa_general_staged": staged})
        d_out = torch.zeros(len(b) * words, dtype=torch.int32, device=dev)
        name = ("distance" if b.kind == "distance" else "collide") + ("_device_f32" if f32 else "_device")
        for _ in range(2):
            getattr(lib, name)(d_s1, d_s2, d_p1, d_p2, len(b), req, d_out, stream=torch.cuda.current_stream().cuda_stream)
            torch.cuda.synchronize()
        recs[staged] = d_out.cpu().numpy().reshape(len(b), words).copy()
        names = [k for k, _ in lib.last_kernel_breakdown()]
        assert ("k_epa_prepare_general" in names) == (staged == "1"), names
        assert lib.last_bucket_counts()["epa_queue"] > 0
        lib.close()
    differ = np.flatnonzero((recs["0"] != recs["1"]).any(axis=1))
    assert differ.size == 0, "%d records differ, first %s" % (differ.size, differ[:10])


def _support(abi, b, shape_id, R, T, n):
    """max over the posed shape of x . n (fp64, from the shape table): Box / Capsule / Convex."""
    sh = b.shapes[shape_id]
    nl = R.T @ n
    p = sh["params"]
    if sh["type"] == abi.GEOM_BOX:
        v = np.sign(nl) * p[:3]
    elif sh["type"] == abi.GEOM_CAPSULE:
        v = np.array([0, 0, np.sign(nl[2]) * p[1]]) + p[0] * nl / np.linalg.norm(nl)
    else:
        assert sh["type"] == abi.GEOM_CONVEX
        V = b.verts[sh["vertex_offset"]:sh["vertex_offset"] + sh["num_points"]]
        v = V[np.argmax(V @ nl)]
    return float((R @ v + T) @ n)


def _overlap_extent(abi, b, k, tf1, tf2, n):
    """Length of the overlap of the two posed shapes of pair k along the unit direction n (from shape 1 to shape 2): the
    penetration depth is the minimum of this over all directions."""
    g = __import__("hppfcl_amd").geometry
    R1, T1, R2, T2 = g.pose_R(tf1), g.pose_T(tf1), g.pose_R(tf2), g.pose_T(tf2)
    return _support(abi, b, b.s1[k], R1, T1, n) + _support(abi, b, b.s2[k], R2, T2, -n)


@pytest.mark.parametrize("case", ["cfg2_box_capsule", "cfg3_convex_convex"])
def test_full_size_properties(pkg, oracle, torch_cuda, case):
    """BASELINE.json full size (1M pairs): EVERY record against the oracle (all host threads; statuses, iteration counts and distances
    equal, points to 1e-9), the invariants of the records (p2 = p1 + d n, |n| = 1) and determinism across two runs."""
    torch = torch_cuda
    abi, wl = pkg.abi, pkg.workloads
    b = getattr(wl, case)(n=1_000_000)
    req = wl.make_request(b, abi)
    lib = pkg.Library(b.lib)
    fn = lib.distance if b.kind == "distance" else lib.collide
    got = fn(b.s1, b.s2, b.tf1, b.tf2, req)
    again = fn(b.s1, b.s2, b.tf1, b.tf2, req)
    assert np.array_equal(got["status"], again["status"])
    assert np.array_equal(np.nan_to_num(got["distance"]), np.nan_to_num(again["distance"]))
    n_ok = check_properties(abi, got, tol=1e-6, name=case)
    assert n_ok > 0.9 * len(b)
    ofn = oracle.distance_batch if b.kind == "distance" else oracle.collide_batch
    ref = ofn(b.shapes, b.verts, b.s1, b.s2, b.tf1, b.tf2, req, n_threads=os.cpu_count() or 8)
    _check_exact(abi, got, ref, case + "-1M")
    check_parity(abi, got, ref, dist_tol=1e-15, point_tol=1e-9, flag_band=0.0, name=case + "-1M")
    frac = abi.status_contact(got["status"]).mean()
    assert 0.2 < frac < 0.45, frac
    lib.close()


@pytest.mark.parametrize("case,kw", [("cfg5_mixed", {}), ("all_primitives", {"kind": "distance"}), ("cfg3_convex_convex", {}), ("large_convex", {"kind": "collide"})])
@pytest.mark.parametrize("f32", [False, True])
def test_tiny_batch_epa_in_the_full_tier_alone(pkg, torch_cuda, case, kw, f32):
    """Option epa_direct_max: every EPA seed of a very small batch moved to the full-capacity tier's queue (k_epa_requeue), the fast tiers not launched --
    the records of the two-tier order byte for byte (fp64; the fp32 path keeps its two tiers, whose instantiations round differently)."""
    torch = torch_cuda
    abi, wl = pkg.abi, pkg.workloads
    dev = torch.device("cuda:0")
    for n in (300, 4000):
        b = getattr(wl, case)(n=n, seed=13, **kw)
        req = wl.make_request(b, abi)
        d_s1, d_s2 = (torch.from_numpy(x.astype(np.int32)).to(dev) for x in (b.s1, b.s2))
        d_p1, d_p2 = (torch.from_numpy(x).to(dev) for x in ((b.pose1_f32, b.pose2_f32) if f32 else (b.tf1, b.tf2)))
        words = 11 if f32 else 24
        name = ("distance" if b.kind == "distance" else "collide") + ("_device_f32" if f32 else "_device")
        recs = {}
        for direct in (4096, 0):
            lib = pkg.Library(b.lib, options={"epa_direct_max": direct})
            try:
                d_out = torch.zeros(len(b) * words, dtype=torch.int32, device=dev)
                for _ in range(2):
                    getattr(lib, name)(d_s1, d_s2, d_p1, d_p2, len(b), req, d_out)
                    torch.cuda.synchronize()
                recs[direct] = d_out.cpu().numpy().copy()
                names = [k for k, _ in lib.last_kernel_breakdown()]
                assert ("k_epa<fast>" in names) == (direct == 0 or f32), names  # (fp64 only: the fp32 tiers do not agree to the bit)
                c = lib.last_bucket_counts()
                assert c["epa_queue"] + c["epa_overflow"] > 0, c  # (large hulls queue for the full tier themselves)
            finally:
                lib.close()
        assert recs[4096].tobytes() == recs[0].tobytes(), (case, n, f32)


@pytest.mark.parametrize("case,kw", [("cfg5_mixed", {}), ("all_primitives", {"kind": "distance"}), ("triangle_pairs", {}), ("large_convex", {})])
def test_small_batch_kernels_beside_each_other(pkg, case, kw):
    """Option gjk_beside_max: the solids' kernels of a small batch (closed forms, k_gjk_prim, the three k_gjk_cvx, k_gjk_large, k_triangle) on four
    streams beside each other, joined in front of the EPA section -- every record and cached guess of the in-line order, twice per library."""
    abi, wl = pkg.abi, pkg.workloads
    for n in (2_000, 50_000):
        b = getattr(wl, case)(n=n, seed=9, **kw)
        req = wl.make_request(b, abi)
        out = {}
        for fan in (120_000, 0):
            lib = pkg.Library(b.lib, options={"gjk_beside_max": fan})
            try:
                fn = lib.distance if b.kind == "distance" else lib.collide
                first, g1 = fn(b.s1, b.s2, b.tf1, b.tf2, req, want_guess=True)
                again, g2 = fn(b.s1, b.s2, b.tf1, b.tf2, req, want_guess=True)
                assert first.tobytes() == again.tobytes() and g1.tobytes() == g2.tobytes()
                out[fan] = (first, g1)
            finally:
                lib.close()
        assert out[120_000][0].tobytes() == out[0][0].tobytes() and out[120_000][1].tobytes() == out[0][1].tobytes()


@pytest.mark.parametrize("case", ["cfg3_convex_convex", "cfg5_mixed"])
def test_split_batches_are_bit_identical(pkg, torch_cuda, case):
    """hfcl_lib_set_split(2): the two halves of a batch run on two streams; every record (and cached guess) is the
    one the unsplit call produces, and the bucket populations add up."""
    abi, wl = pkg.abi, pkg.workloads
    b = getattr(wl, case)(n=300_001)
    req = wl.make_request(b, abi)
    lib = pkg.Library(b.lib)
    lib.set_host_chunk(len(b))  # the host entry points pipeline chunks; here the batch is one chunk (chunking: its own test)
    fn = lib.distance if b.kind == "distance" else lib.collide
    lib.set_split(1)
    one, g1 = fn(b.s1, b.s2, b.tf1, b.tf2, req, want_guess=True)
    assert lib.last_split_parts() == 1
    c1 = lib.last_bucket_counts()
    lib.set_split(0)  # automatic: on for the mixed library only
    fn(b.s1, b.s2, b.tf1, b.tf2, req)
    assert lib.last_split_parts() == (2 if case == "cfg5_mixed" else 1)
    lib.set_split(2)
    assert lib.get_split() == 2
    for _ in range(2):  # twice: the helper's workspace is reused
        two, g2 = fn(b.s1, b.s2, b.tf1, b.tf2, req, want_guess=True)
        c2 = lib.last_bucket_counts()
        assert one.tobytes() == two.tobytes() and g1.tobytes() == g2.tobytes()
        assert c1 == c2
    # fp32 device-resident entry point on a caller-owned stream
    torch = torch_cuda
    dev = torch.device("cuda:0")
    d = [torch.from_numpy(x).to(dev) for x in (b.s1.astype(np.int32), b.s2.astype(np.int32), b.pose1_f32, b.pose2_f32)]
    f32 = lib.distance_device_f32 if b.kind == "distance" else lib.collide_device_f32
    outs = []
    for parts in (1, 2):
        lib.set_split(parts)
        o = torch.zeros(len(b) * 11, dtype=torch.int32, device=dev)
        st = torch.cuda.Stream()
        with torch.cuda.stream(st):
            f32(*d, len(b), req, o, stream=st.cuda_stream)
            o2 = o.clone()  # ordered after the call on the caller's stream: must see complete results
        torch.cuda.synchronize()
        outs.append((o.cpu().numpy(), o2.cpu().numpy()))
    assert np.array_equal(outs[0][0], outs[1][0]) and np.array_equal(outs[1][0], outs[1][1])
    lib.close()


@pytest.mark.parametrize("case", ["cfg2_box_capsule", "cfg5_mixed", "cfg3_convex_convex"])
def test_host_pipeline_equals_device_path(pkg, torch_cuda, case):
    """The host-buffer entry points (chunked H2D | kernels | D2H pipeline, what hpp::fcl::collide()/distance() callers
    get) return byte-identical records, warm-start guesses and bucket populations to one device-resident call, for
    any chunk size (ragged last chunk, more chunks than pipeline slots, split chunks)."""
    torch = torch_cuda
    abi, wl = pkg.abi, pkg.workloads
    b = getattr(wl, case)(n=300000)
    req = wl.make_request(b, abi)
    lib = pkg.Library(b.lib)
    dev = torch.device("cuda:0")
    n = len(b)
    d = [torch.from_numpy(x).to(dev) for x in (b.s1.astype(np.int32), b.s2.astype(np.int32), b.tf1, b.tf2)]
    d_out = torch.zeros(n * 24, dtype=torch.int32, device=dev)
    d_g = torch.zeros(n * 8, dtype=torch.int32, device=dev)
    fn_dev = lib.distance_device if b.kind == "distance" else lib.collide_device
    lib.set_split(1)
    fn_dev(*d, n, req, d_out, d_gout=d_g, stream=torch.cuda.current_stream().cuda_stream)
    torch.cuda.synchronize()
    ref = d_out.cpu().numpy().view(abi.RESULT_DTYPE)
    ref_g = d_g.cpu().numpy().view(abi.GUESS_DTYPE)
    ref_counts = lib.last_bucket_counts()
    fn_host = lib.distance if b.kind == "distance" else lib.collide
    for chunk, split in ((0, 1), (7777, 1), (70001, 1), (150000, 2), (n, 1)):
        lib.set_host_chunk(chunk)
        lib.set_split(split)
        got, g = fn_host(b.s1, b.s2, b.tf1, b.tf2, req, want_guess=True)
        assert got.tobytes() == ref.tobytes(), "records differ (chunk %d)" % chunk
        assert g.tobytes() == ref_g.tobytes(), "guesses differ (chunk %d)" % chunk
        assert lib.last_bucket_counts() == ref_counts, "bucket populations differ (chunk %d)" % chunk
    # compact poses (quaternion + translation): the rotation is rebuilt on the device, so only round-off may differ
    lib.set_host_chunk(0)
    fn_qt = lib.distance_qt if b.kind == "distance" else lib.collide_qt
    got = fn_qt(b.s1, b.s2, b.pose1_qt, b.pose2_qt, req)
    ok = np.isfinite(ref["distance"]) & (np.abs(ref["distance"]) < 1e300)
    assert np.array_equal(np.isfinite(got["distance"]), np.isfinite(ref["distance"]))
    dd = np.abs(got["distance"][ok] - ref["distance"][ok])
    assert dd.max() < 1e-9, dd.max()
    near = np.abs(ref["distance"]) < 1e-9
    assert np.all((abi.status_contact(got["status"]) == abi.status_contact(ref["status"])) | near)
    lib.close()


# ------------------------------------------------------------------------------------- BVH (cfg4)
def _run_bvh(pkg, oracle, b, req, max_contacts=0, options=None):
    bb = pkg.bvh_builder
    ML = bb.MeshLibrary(b.meshes)
    lib = pkg.workloads.make_library(pkg, b, options=options)
    try:
        if max_contacts:
            got, cgot, produced = lib.collide_contacts(b.s1, b.s2, b.tf1, b.tf2, req, max_contacts)
            ref, cref = oracle.bvh_collide_batch(ML, b.s1, b.s2, b.tf1, b.tf2, req, max_contacts=max_contacts, n_threads=8)
            return got, ref, cgot, cref, produced, lib.last_kernel_breakdown()
        got = lib.collide(b.s1, b.s2, b.tf1, b.tf2, req)
        ref = oracle.bvh_collide_batch(ML, b.s1, b.s2, b.tf1, b.tf2, req, n_threads=8)
        return got, ref, lib.last_kernel_breakdown()
    finally:
        lib.close()


def _check_bvh_records(abi, got, ref, name):
    assert not np.any((got["status"] >> 30) & 1), name + ": traversal stack overflow"
    near = np.abs(ref["distance"]) < 1e-9
    same_nc = got["num_contacts"] == ref["num_contacts"]
    assert np.all(same_nc | near), "%s: contact counts differ on %d queries" % (name, (~(same_nc | near)).sum())
    ok = same_nc & ~near
    assert np.array_equal(got["b1"][ok], ref["b1"][ok]) and np.array_equal(got["b2"][ok], ref["b2"][ok]), \
        name + ": first-contact primitive ids differ"
    fin = (np.abs(ref["distance"]) < 1e300) & ok
    assert np.abs(got["distance"][fin] - ref["distance"][fin]).max() < 1e-6
    assert np.array_equal(np.isnan(got["p1"][ok]), np.isnan(ref["p1"][ok]))
    sep_g, sep_r = got["p2"] - got["p1"], ref["p2"] - ref["p1"]
    m = fin & ~np.isnan(ref["p1"]).any(axis=1)
    assert np.abs(sep_g[m] - sep_r[m]).max() < 1e-5


@pytest.mark.parametrize("form", ["default", "filter"])
@pytest.mark.parametrize("seg,n", [(12, 20000), (50, 4000)])
def test_bvh_collide_first_contact(pkg, oracle, seg, n, form):
    """Default request (num_max_contacts = 1): collision flag and the first contact's (b1, b2) in the
    reference's DFS order are exact; depth / witness data to 1e-6.  filter: through the fp32 separating-axis filter in
    front of the fp64 test (HFCL_BVH_FILTER=1: the decisions of the default form, numbers to the last bits)."""
    abi, wl = pkg.abi, pkg.workloads
    if form == "filter" and not pkg.engine.has_ab_forms():
        pytest.skip("the fp32 filter form lost its A/B (profiles/r03_b) and is not in the product build: tools/build_variant.sh ab host,k_bvh,k_epa "
                    "-DHFCL_KEEP_AB_FORMS=1, then HFCL_LIB_PATH=build/ab/lib_ab.so")
    if form == "filter":
        b0 = wl.cfg4_mesh_mesh(n=n, seg=seg, ring=seg, n_variants=4)
        plain, _, _ = _run_bvh(pkg, oracle, b0, wl.make_request(b0, abi))
    b = wl.cfg4_mesh_mesh(n=n, seg=seg, ring=seg, n_variants=4)
    req = wl.make_request(b, abi)
    got, ref, kt = _run_bvh(pkg, oracle, b, req, options={"bvh_filter": 1} if form == "filter" else None)
    _check_bvh_records(abi, got, ref, "bvh-first-%d" % seg)
    if form == "filter":
        # the same decisions as the plain fp64 kernel; the two are different instantiations, so their fp64 arithmetic may be
        # contracted differently (last bits of the reported bound)
        for f in ("num_contacts", "b1", "b2", "status"):
            assert np.array_equal(got[f], plain[f]), f
        fin = np.abs(plain["distance"]) < 1e300
        assert np.array_equal(fin, np.abs(got["distance"]) < 1e300)
        assert np.abs(got["distance"][fin] - plain["distance"][fin]).max() < 1e-12
    frac = (ref["num_contacts"] > 0).mean()
    assert 0.2 < frac < 0.8, frac


@pytest.mark.parametrize("n", [100_000, 250_000])
def test_bvh_collide_baseline_size(pkg, oracle, n):
    """BASELINE.json configs[3] at its size: 100k mesh pairs of 5 000-triangle models, and a 250k-query batch (the batch form
    the host picks for them: one query per lane for 256 steps, then the suspended queries continued by waves that walk 64
    stack entries per trip, k_bvh_coop).  Every record against the oracle: contact counts, first-contact triangle ids in DFS
    order, depth and witness data."""
    abi, wl = pkg.abi, pkg.workloads
    b = wl.cfg4_mesh_mesh(n=n)
    assert len(b.meshes) == 8 and all(len(m.triangles) == 5000 for m in b.meshes)
    req = wl.make_request(b, abi)
    ML = pkg.bvh_builder.MeshLibrary(b.meshes)
    lib = wl.make_library(pkg, b)
    try:
        got = lib.collide(b.s1, b.s2, b.tf1, b.tf2, req)
        again = lib.collide(b.s1, b.s2, b.tf1, b.tf2, req)
    finally:
        lib.close()
    assert got.tobytes() == again.tobytes()  # no scheduling-dependent choice in either form
    ref = oracle.bvh_collide_batch(ML, b.s1, b.s2, b.tf1, b.tf2, req, n_threads=32)
    _check_bvh_records(abi, got, ref, "bvh-cfg4-%d" % n)
    assert 0.2 < (ref["num_contacts"] > 0).mean() < 0.8


@pytest.mark.parametrize("form", ["coop", "cut", "rounds", "inline", "levels", "whole"])
def test_bvh_collide_forms_agree(pkg, oracle, form):
    """The forms of a long mesh x mesh walk -- continued 64 entries wide by a wave (with a budget of 24 steps, so that
    nearly every query is), the same with the waves' long walks cut into chunks that later launches walk and k_bvh_combine folds
    back (BvhSplit::cut_ticks, here after 15 000 clock ticks: thousands of cuts, chunks cut again, chunks behind a contact), cut into
    task levels by the lanes, and walked in one piece by its lane -- give the oracle's records; so do the forms of the queries' own
    phase in front of the continuation: walk / leaves / resolve rounds (the default: here four rounds of 24-48 box tests and 2-4 listed
    leaves, every round's hand-overs continued on a stream of their own beside the next) and k_bvh_collide with its leaves inline
    (bvh_walk_rounds = 0); and the fp32 device path (its own
    instantiations of the same kernels) the same decisions away from the decision boundary."""
    import torch
    abi, wl = pkg.abi, pkg.workloads
    options = {"coop": {"bvh_budget0_coop": 24}, "cut": {"bvh_budget0_coop": 24, "bvh_cut_ticks": 15000},
               "rounds": {"bvh_budget0_coop": 24, "bvh_walk_rounds": 4, "bvh_walk_k": [2, 3, 4, 16], "bvh_walk_budget": [24, 24, 48]},
               "inline": {"bvh_budget0_coop": 24, "bvh_walk_rounds": 0},
               "levels": {"bvh_coop": 0}, "whole": {"bvh_coop": 0, "bvh_levels": 1}}[form]
    b = wl.cfg4_mesh_mesh(n=30_000, seed=21)
    req = wl.make_request(b, abi)
    ML = pkg.bvh_builder.MeshLibrary(b.meshes)
    ref = oracle.bvh_collide_batch(ML, b.s1, b.s2, b.tf1, b.tf2, req, n_threads=32)
    lib = wl.make_library(pkg, b, options=options)
    try:
        got = lib.collide(b.s1, b.s2, b.tf1, b.tf2, req)
        dev = torch.device("cuda:0")
        d = [torch.from_numpy(x).to(dev) for x in (b.s1.astype(np.int32), b.s2.astype(np.int32), b.pose1_f32, b.pose2_f32)]
        out = torch.zeros(len(b) * 11, dtype=torch.int32, device=dev)
        lib.collide_device_f32(*d, len(b), req, out)
        torch.cuda.synchronize()
        g32 = out.cpu().numpy().view(abi.RESULT_F32_DTYPE)
    finally:
        lib.close()
    _check_bvh_records(abi, got, ref, "bvh-forms-" + form)
    clear = np.abs(ref["distance"]) > 1e-3
    assert np.array_equal(abi.status_contact(g32["status"])[clear] != 0, ref["num_contacts"][clear] > 0)
    hit = clear & (ref["num_contacts"] > 0)
    # (fp32 has no triangle ids to compare; where a box test falls the other way in fp32 the walk ends on another contact of the
    # same pair of meshes, with its own depth: a handful of queries)
    err = np.abs(g32["distance"][hit] - ref["distance"][hit])
    assert np.quantile(err, 0.995) < 2e-4 and (err > 2e-4).sum() < 0.005 * hit.sum()


def test_bvh_collide_all_contacts(pkg, oracle):
    """num_max_contacts = inf: the sorted contact set per query equals the oracle's
    (test/collision.cpp:419-422,523-563 semantics)."""
    abi, wl = pkg.abi, pkg.workloads
    b = wl.cfg4_mesh_mesh(n=3000, seg=14, ring=14, n_variants=3)
    req = wl.make_request(b, abi, num_max_contacts=10 ** 6)
    got, ref, cgot, cref, produced, kt = _run_bvh(pkg, oracle, b, req, max_contacts=4 * 10 ** 6)
    assert produced == len(cref) == len(cgot)
    assert np.array_equal(got["num_contacts"], ref["num_contacts"])
    kg = np.lexsort((cgot["b2"], cgot["b1"], cgot["pair"]))
    kr = np.lexsort((cref["b2"], cref["b1"], cref["pair"]))
    for f in ("pair", "b1", "b2"):
        assert np.array_equal(cgot[f][kg], cref[f][kr])
    assert np.abs(cgot["penetration_depth"][kg] - cref["penetration_depth"][kr]).max() < 1e-6


def test_bvh_security_margin_and_mixed_batch(pkg, oracle):
    """margin > 0 widens contacts; a batch mixing mesh pairs and primitive pairs goes to the right kernels."""
    abi, wl, g = pkg.abi, pkg.workloads, pkg.geometry
    b = wl.cfg4_mesh_mesh(n=3000, seg=12, ring=12, n_variants=2)
    req = wl.make_request(b, abi, security_margin=0.05)
    got, ref, kt = _run_bvh(pkg, oracle, b, req)
    _check_bvh_records(abi, got, ref, "bvh-margin")
    req0 = wl.make_request(b, abi)
    got0, ref0, _ = _run_bvh(pkg, oracle, b, req0)
    assert (got["num_contacts"] > 0).sum() >= (got0["num_contacts"] > 0).sum()


def _check_distance_records(oracle, ML, b, got, ref, what, max_ties=0):
    """mesh x mesh distance() records against the oracle's: the distance to 0 ulp (the device unit is built without
    contraction, as the oracle and the reference's default build), the triangle ids EQUAL (max_ties = 0: the default kernels, whose
    pooled continuation hands every walk whose reported pair could hang on a rounding error back to the ordered walk), the witness
    points to 0 ulp as well.  max_ties > 0 -- the wave-per-walk continuation HFCL_BVHD_POOL=0 only, which splits box pairs ahead of
    their turn -- allows that share of records, enumerated here, where the pair the device reports has exactly the oracle's distance
    too (two triangle pairs at 0 ulp: which of them a walk reports then hangs on a bound that exceeds a distance below it by an ulp)."""
    assert not np.any((got["status"] >> 30) & 1), what + ": traversal stack overflow"
    assert np.array_equal(got["distance"], ref["distance"]), what + ": distances differ from the oracle's"
    same = (got["b1"] == ref["b1"]) & (got["b2"] == ref["b2"])
    ties = np.where(~same)[0]
    allowed = max(1, int(max_ties * len(ref))) if max_ties else 0
    assert len(ties) <= allowed, "%s: %d records with other triangle ids: %s" % (what, len(ties), ties[:8])
    for k in ties:  # enumerated: the reported pair is at the oracle's minimal distance, bit for bit
        d = oracle.bvh_leaf_distance(ML, b.s1[k], b.s2[k], b.tf1[k], b.tf2[k], got["b1"][k], got["b2"][k])
        assert d == ref["distance"][k], "%s: record %d reports pair (%d, %d) at %.17g, the minimum is %.17g" % (
            what, k, got["b1"][k], got["b2"][k], d, ref["distance"][k])
    pos = (ref["distance"] > 0) & same
    assert np.array_equal(got["p1"][pos], ref["p1"][pos]) and np.array_equal(got["p2"][pos], ref["p2"][pos]), what + ": witness points"
    assert np.isnan(got["normal"]).all()
    return len(ties)


@pytest.mark.parametrize("seg,n,hw", [(12, 20000, 2.5), (50, 3000, 2.2)])
def test_bvh_distance(pkg, oracle, seg, n, hw):
    """BVHModel<OBBRSS> distance(): min distance, nearest triangle ids and nearest points vs the oracle, bit for bit."""
    abi, wl, bb = pkg.abi, pkg.workloads, pkg.bvh_builder
    b = wl.cfg4_mesh_mesh(n=n, seg=seg, ring=seg, n_variants=4, half_width=hw)
    ML = bb.MeshLibrary(b.meshes)
    lib = wl.make_library(pkg, b)
    got = lib.distance(b.s1, b.s2, b.tf1, b.tf2)
    lib.close()
    ref = oracle.bvh_distance_batch(ML, b.s1, b.s2, b.tf1, b.tf2, n_threads=8)
    _check_distance_records(oracle, ML, b, got, ref, "bvh-distance")
    pos = ref["distance"] > 1e-9
    assert 0.05 < (ref["distance"] == 0).mean() < 0.9
    # |p2 - p1| = d for separated meshes
    assert np.abs(np.linalg.norm(got["p2"][pos] - got["p1"][pos], axis=1) - got["distance"][pos]).max() < 1e-7


def test_bvh_distance_continuations(pkg, oracle):
    """distance() walks past their step budget are continued by waves: k_bvh_distance_pool (default: several walks per wave,
    their box and triangle tests pooled, order kept by a marker) or k_bvh_distance_coop (HFCL_BVHD_POOL=0: a wave per walk, 64
    stack entries per trip, applied in order).  With a budget of 16 steps (every query continues there), the default, and
    "never" (the lane's sequential walk), in both forms: the oracle's records, and the same records between the forms, byte
    for byte but for enumerated 0-ulp ties."""
    abi, wl, bb = pkg.abi, pkg.workloads, pkg.bvh_builder
    b = wl.cfg4_mesh_mesh_distance(n=3000, seed=4)
    ML = bb.MeshLibrary(b.meshes)
    ref = oracle.bvh_distance_batch(ML, b.s1, b.s2, b.tf1, b.tf2, n_threads=32)
    assert 0.4 < (ref["distance"] > 1e-9).mean() < 0.95
    res = {}
    for pool, budget in (("1", "16"), ("1", ""), ("0", "16"), ("0", "1024"), ("1", "0")):
        options = {"bvhd_pool": pool}
        if budget:
            options["bvhd_budget"] = budget
        lib = wl.make_library(pkg, b, options=options)
        try:
            res[(pool, budget)] = lib.distance(b.s1, b.s2, b.tf1, b.tf2)
        finally:
            lib.close()
    n_ties = 0
    for key, got in res.items():  # the pooled continuation (the default) and the lanes alone: no exception at all
        n_ties += _check_distance_records(oracle, ML, b, got, ref, "continuation %s/%s" % key, max_ties=0.001 if key[0] == "0" else 0)
    base = res[("1", "0")]  # the lanes' sequential walk
    for key, got in res.items():
        same = (got["b1"] == base["b1"]) & (got["b2"] == base["b2"])
        for f in ("distance", "p1", "p2", "status", "num_contacts"):
            assert np.array_equal(got[f][same], base[f][same]), (key, f)
    assert n_ties <= 2


@pytest.mark.parametrize("n", [100_000])
def test_bvh_distance_at_baseline_size(pkg, oracle, n):
    """BASELINE.json configs[3]'s distance() variant at its size: 100 000 queries on 5 000-triangle models.  Every record
    against the oracle (all host threads), ids exact but for enumerated 0-ulp ties."""
    abi, wl, bb = pkg.abi, pkg.workloads, pkg.bvh_builder
    b = wl.cfg4_mesh_mesh_distance(n=n, seed=1)
    ML = bb.MeshLibrary(b.meshes)
    lib = wl.make_library(pkg, b)
    got = lib.distance(b.s1, b.s2, b.tf1, b.tf2)
    reruns = lib.last_ordered_reruns()
    again = lib.distance(b.s1, b.s2, b.tf1, b.tf2)
    lib.close()
    assert reruns["mesh_continued"] > 1000 and reruns["mesh_rerun"] <= reruns["mesh_continued"] // 20
    ref = oracle.bvh_distance_batch(ML, b.s1, b.s2, b.tf1, b.tf2, n_threads=os.cpu_count() or 8)
    _check_distance_records(oracle, ML, b, got, ref, "cfg4d")
    _check_distance_records(oracle, ML, b, again, ref, "cfg4d, second run")
    # which walks share a wave depends on the order the waves take their tickets, and with it the order in which a walk's pairs are
    # evaluated -- the records do not: a walk whose reported pair could depend on it is re-run in the reference's order
    assert np.array_equal(got.view(np.uint8), again.view(np.uint8)), "distance() is not deterministic"
    print("cfg4d %d queries: walks continued by waves / re-run in order: %s" % (n, reruns))


def _mesh_batch(pkg, meshes, n, seed, half_width):
    wl, g = pkg.workloads, pkg.geometry
    rng = np.random.default_rng(seed)
    lib = g.ShapeLibrary()
    for k, m in enumerate(meshes):
        lib.add_bvh(k, len(m.vertices))
    s1, s2 = rng.integers(0, len(meshes), n), rng.integers(0, len(meshes), n)
    q1, T1, q2, T2 = wl._poses(rng, n, half_width)
    b = wl.Batch("mesh_pairs", lib, s1, s2, q1, T1, q2, T2, "collide")
    b.meshes = meshes
    return b


def test_bvh_models_beyond_16_bit_node_ids(pkg, oracle):
    """A BVHModel of 72 200 triangles (144 399 nodes: node ids no longer fit 16 bits, BV_node.h:57 uses int): collide()
    and distance() against the oracle, no traversal flagged as overflowed."""
    abi, bb = pkg.abi, pkg.bvh_builder
    big = bb.Mesh(*bb.bumpy_sphere(190, 190, r=1.0, amp=0.1, freq=3, phase=0.3))
    small = bb.Mesh(*bb.bumpy_sphere(20, 20, r=0.7, amp=0.1, freq=2, phase=0.1))
    assert len(big.nodes) > 65535
    b = _mesh_batch(pkg, [big, small], 600, 3, 1.1)
    ML = bb.MeshLibrary(b.meshes)
    lib = pkg.workloads.make_library(pkg, b)
    req = pkg.workloads.make_request(b, abi)
    got = lib.collide(b.s1, b.s2, b.tf1, b.tf2, req)
    ref = oracle.bvh_collide_batch(ML, b.s1, b.s2, b.tf1, b.tf2, req, n_threads=16)
    _check_bvh_records(abi, got, ref, "bvh-wide")
    assert 0.1 < (ref["num_contacts"] > 0).mean() < 0.9
    gd = lib.distance(b.s1, b.s2, b.tf1, b.tf2)
    rd = oracle.bvh_distance_batch(ML, b.s1, b.s2, b.tf1, b.tf2, n_threads=16)
    assert not np.any((gd["status"] >> 30) & 1)
    assert np.abs(gd["distance"] - rd["distance"]).max() < 1e-9
    lib.close()
    # the same model against solids: the one-query-per-lane walk and its wave continuation carry 32-bit node ids in their entries
    g = pkg.geometry
    rng = np.random.default_rng(5)
    L = g.ShapeLibrary()
    L.add_bvh(0, len(big.vertices))
    for _ in range(8):
        L.add_box(*map(float, rng.uniform(0.2, 0.8, 3)))
        L.add_capsule(float(rng.uniform(0.1, 0.3)), float(rng.uniform(0.2, 0.8)))
    n = 1500
    q1, T1, q2, T2 = pkg.workloads._poses(rng, n, 1.4)
    bs = pkg.workloads.Batch("big_mesh_x_solid", L, np.zeros(n, dtype=np.int64), rng.integers(1, 17, n), q1, T1, q2, T2, "collide")
    bs.meshes = [big]
    MLs = bb.MeshLibrary(bs.meshes)
    creq = abi.default_collision_request()
    refs, _ = oracle.mixed_collide_batch(bs.shapes, bs.verts, MLs, bs.s1, bs.s2, bs.tf1, bs.tf2, creq, max_contacts=10 ** 4, n_threads=16)
    libs = pkg.workloads.make_library(pkg, bs)
    try:
        gots = libs.collide(bs.s1, bs.s2, bs.tf1, bs.tf2, creq)
        gds = libs.distance(bs.s1, bs.s2, bs.tf1, bs.tf2)
    finally:
        libs.close()
    rds = oracle.mixed_distance_batch(bs.shapes, bs.verts, MLs, bs.s1, bs.s2, bs.tf1, bs.tf2, None, n_threads=16)
    assert not np.any((gots["status"] >> 30) & 1) and not np.any((gds["status"] >> 30) & 1)
    near = np.abs(refs["distance"]) < 1e-9
    assert ((gots["num_contacts"] == refs["num_contacts"]) | near).all()
    assert np.array_equal(gots["b1"][~near], refs["b1"][~near])
    assert 0.1 < (refs["num_contacts"] > 0).mean() < 0.9
    sep = rds["distance"] > 1e-6
    assert sep.sum() > 300 and np.abs(gds["distance"][sep] - rds["distance"][sep]).max() < 1e-6


@pytest.mark.parametrize("force_wide", [False, True])
def test_bvh_degenerate_deep_tree(pkg, oracle, force_wide):
    """A strip of triangles whose positions grow geometrically: the mean split (BV_splitter) peels a few triangles off
    per level and the tree gets deep (133 levels here; a mean split can only stay this lopsided while the coordinates
    grow faster than geometrically, so the range of a double bounds the depth of any BVHModel to a few hundred levels).
    The reference's traversal stack is a growable vector (traversal_recurse.cpp:95).  Here a full 96-entry LDS stack
    suspends into tasks (default form), or continues in a per-lane global slab (wide form: models beyond 65535 nodes or
    deeper than the task levels hold; forced here with the option bvh_force_wide).  collide() and distance() vs the oracle."""
    abi, bb = pkg.abi, pkg.bvh_builder
    nt, r = 1200, 3.2
    x = r ** (np.arange(nt // 2 + 2) - float(nt // 2 + 1))  # largest coordinate 1
    v = np.zeros((2 * (nt // 2 + 2), 3))
    k = np.arange(len(v) // 2)
    v[0::2] = np.stack([x[k], 0 * x[k], 0.02 * x[k]], 1)
    v[1::2] = np.stack([x[k], 0.3 * x[k], -0.02 * x[k]], 1)
    t = np.array([[i, i + 1, i + 2] for i in range(nt)], dtype=np.uint32)
    m = bb.Mesh(v, t)
    depth = 1 + int(_depths(m.nodes).max())
    assert depth > 100, depth
    b = _mesh_batch(pkg, [m], 256, 4, 0.05)
    ML = bb.MeshLibrary(b.meshes)
    lib = pkg.workloads.make_library(pkg, b, options={"bvh_force_wide": 1} if force_wide else None)
    req = pkg.workloads.make_request(b, abi)
    got = lib.collide(b.s1, b.s2, b.tf1, b.tf2, req)
    ref = oracle.bvh_collide_batch(ML, b.s1, b.s2, b.tf1, b.tf2, req, n_threads=16)
    _check_bvh_records(abi, got, ref, "bvh-deep")
    gd = lib.distance(b.s1, b.s2, b.tf1, b.tf2)
    rd = oracle.bvh_distance_batch(ML, b.s1, b.s2, b.tf1, b.tf2, n_threads=16)
    assert not np.any((gd["status"] >> 30) & 1)
    assert np.abs(gd["distance"] - rd["distance"]).max() < 1e-9
    lib.close()


def _depths(nodes):
    d = np.zeros(len(nodes), dtype=np.int64)
    for i, fc in enumerate(nodes["first_child"]):  # children always follow their parent in the node array
        if fc > 0:
            d[fc] = d[fc + 1] = d[i] + 1
    return d


def test_cpp_shim_runs_reference_style_tests(pkg):
    """tests/cpp/test_compat.cpp: hpp-fcl-named C++ (collide/distance/CollisionRequest/...) on the GPU."""
    import os
    import subprocess
    d = os.path.join(os.path.dirname(os.path.abspath(__file__)), "cpp")
    subprocess.check_call(["make", "-s", "-C", d])
    r = subprocess.run([os.path.join(d, "test_compat")], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stdout + r.stderr


@pytest.mark.parametrize("kind", ["collide", "distance"])
def test_flat_rows_gpu(pkg, oracle, kind):
    """Plane / Halfspace rows (closed forms, details.h:347-428,509-691) on the device vs the oracle: statuses
    exact, numbers to 1e-9 relative (FMA contraction only), infinite penetrations (-DBL_MAX) exact."""
    abi, wl = pkg.abi, pkg.workloads
    b = wl.flat_pairs(n=60000, kind=kind)
    req = wl.make_request(b, abi)
    ref = _oracle(oracle, b, req)
    got, buckets = _engine(pkg, b, req)
    assert buckets["unsupported"] == 0 and buckets["closed"] == len(b)
    assert np.array_equal(got["status"], ref["status"])
    inf = ref["distance"] == -np.finfo(np.float64).max
    assert np.array_equal(got["distance"] == -np.finfo(np.float64).max, inf) and 0.02 < inf.mean() < 0.2
    with np.errstate(invalid="ignore", over="ignore"):
        for f in ("distance", "normal", "p1", "p2"):
            a, r = got[f].astype(np.float64), ref[f].astype(np.float64)
            fin = np.isfinite(r) & (np.abs(r) < 1e300)
            assert np.array_equal(np.isnan(a), np.isnan(r)), f
            assert np.all(np.abs(a[fin] - r[fin]) <= 1e-9 * (1 + np.abs(r[fin]))), f


@pytest.mark.parametrize("support", ["scan", "climb"])
@pytest.mark.parametrize("kind", ["distance", "collide"])
def test_large_hulls_gpu(pkg, oracle, kind, support):
    """Hulls of 33..256 vertices (k_gjk_large + full-capacity EPA tier) vs the oracle's neighbour hill-climbing support
    (support_functions.cpp:323-397): with the vertices scanned from memory, and with the registered vertex adjacency
    climbed from the previous answer (hfcl_lib_set_convex_neighbors; HFCL_CLIMB_MIN lowered so that these hulls use it)."""
    abi, wl = pkg.abi, pkg.workloads
    b = wl.large_convex(n=60000, kind=kind)
    oracle.register_hull_neighbors(b.shapes, b.verts)
    try:
        req = wl.make_request(b, abi)
        ref = _oracle(oracle, b, req)
    finally:
        oracle.lib().orc_clear_neighbors()
    if support == "climb":
        lib = pkg.Library(b.lib, device=0, options={"climb_min": 33})
        try:
            first = int(np.flatnonzero(b.shapes["num_points"] > 32)[0])
            n0 = int(b.shapes["num_points"][first])
            for offs, ids in ((np.zeros(n0 + 1, np.uint32), np.zeros(1, np.uint32)),                  # empty
                              (np.arange(n0 + 1, dtype=np.uint32), np.full(n0, n0, np.uint32)),     # index out of range
                              (np.arange(n0 + 1, dtype=np.uint32)[::-1].copy(), np.zeros(n0, np.uint32))):  # offsets decrease
                with pytest.raises(pkg.EngineError):
                    lib.set_convex_neighbors(first, offs, ids)
            with pytest.raises(pkg.EngineError):  # not a convex shape / not a shape
                lib.set_convex_neighbors(len(b.shapes), np.zeros(2, np.uint32), np.zeros(1, np.uint32))
            assert wl.register_adjacency(lib, b.shapes, b.verts) == b.n_large
            run = lib.distance if b.kind == "distance" else lib.collide
            got, buckets = run(b.s1, b.s2, b.tf1, b.tf2, req), lib.last_bucket_counts()
            again = run(b.s1, b.s2, b.tf1, b.tf2, req)  # deterministic: the walk has no scheduling-dependent choice
            assert got.tobytes() == again.tobytes()
        finally:
            lib.close()
    else:
        got, buckets = _engine(pkg, b, req)
    assert buckets["large"] == len(b) and buckets["unsupported"] == 0
    st = check_parity(abi, got, ref, dist_tol=1e-6, point_tol=1e-5, flag_band=1e-9, name="large-" + kind)
    assert st["p999_dd"] < 1e-9, st
    check_properties(abi, got, tol=1e-6, name="large-" + kind)


@pytest.mark.parametrize("kind", ["distance", "collide"])
def test_hill_climb_on_flat_triangulated_facets(pkg, oracle, kind):
    """A subdivided box (602 points, most of them inside a flat facet with all their neighbours in the facet's plane) climbed
    along axis-aligned directions: every neighbour of such a point ties, and along the facet's inward normal none improves
    although the facet is the hull's minimum.  The reference walks over equal neighbours (loose_check,
    support_functions.cpp:368-382); the device answers that plateau with the scan.  Axis-aligned placements (identity
    rotations: GJK's directions stay on the axes) and random poses, against the analytic box distance, the oracle's
    hill-climb and the device's own scan."""
    abi, wl, g = pkg.abi, pkg.workloads, pkg.geometry
    pts, offs, ids = wl.subdivided_box(11)
    assert len(pts) >= 512
    L = pkg.ShapeLibrary()
    hull, sph, box = L.add_convex(pts), L.add_sphere(0.5), L.add_box(1.0, 1.0, 1.0)
    rng = np.random.default_rng(17)
    axes = np.array([[1, 0, 0], [-1, 0, 0], [0, 1, 0], [0, -1, 0], [0, 0, 1], [0, 0, -1]], dtype=np.float64)
    offsets = np.concatenate([axes * d for d in (3.0, 2.0, 1.6, 1.2, 0.7)])  # separated ... deeply penetrating
    n_ax = len(offsets)
    n = n_ax * 4 + 4000
    s1 = np.full(n, hull)
    s2 = np.where(np.arange(n) % 2 == 0, sph, box)
    tf1, tf2 = g.make_pose(T=np.zeros((n, 3))), g.make_pose(T=np.zeros((n, 3)))
    for rep in range(4):  # hull first / second, sphere / box as the other shape
        sl = slice(rep * n_ax, (rep + 1) * n_ax)
        tf2[sl] = g.make_pose(T=offsets)
        s2[sl] = sph if rep < 2 else box
    sw = slice(n_ax, 2 * n_ax), slice(3 * n_ax, 4 * n_ax)
    for sl in sw:  # operands exchanged: the hull is shape 2
        s1[sl], s2[sl] = s2[sl].copy(), hull
        tf1[sl], tf2[sl] = tf2[sl].copy(), g.make_pose(T=np.zeros((n_ax, 3)))
    q = rng.normal(size=(4000, 4))
    q /= np.linalg.norm(q, axis=1, keepdims=True)
    tf2[4 * n_ax:] = g.make_pose(quat=q, T=rng.uniform(-2.2, 2.2, (4000, 3)))
    req = abi.default_distance_request() if kind == "distance" else abi.default_collision_request()
    S, V = L.shapes_array(), L.vertices_array()
    oracle.register_hull_neighbors(S, V, graphs={int(hull): (offs, ids)})
    try:
        ofn = oracle.distance_batch if kind == "distance" else oracle.collide_batch
        ref = ofn(S, V, s1, s2, tf1, tf2, req, n_threads=8)
    finally:
        oracle.lib().orc_clear_neighbors()
    outs = {}
    for mode in ("climb", "scan"):
        lib = pkg.Library(L, device=0)
        try:
            if mode == "climb":
                lib.set_convex_neighbors(hull, offs, ids)
            run = lib.distance if kind == "distance" else lib.collide
            outs[mode] = run(s1, s2, tf1, tf2, req)
            assert lib.last_bucket_counts()["large"] == n
        finally:
            lib.close()
    # analytic: a sphere of radius 0.5 whose centre sits at distance t on an axis from a unit-half-side box: t - 1 - 0.5
    for rep in (0, 1):
        sl = slice(rep * n_ax, (rep + 1) * n_ax)
        want = np.linalg.norm(offsets, axis=1) - 1.5
        for mode in outs:
            assert np.abs(outs[mode]["distance"][sl] - want).max() < 1e-6, (mode, rep)
    for mode in outs:
        check_parity(abi, outs[mode], ref, dist_tol=1e-6, point_tol=2e-5, flag_band=1e-9, name="subdivided-box-" + mode)
    assert np.abs(outs["climb"]["distance"] - outs["scan"]["distance"]).max() < 1e-9
