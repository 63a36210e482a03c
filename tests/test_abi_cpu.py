"""CPU checks of the C-ABI library: it loads, exports every symbol include/hppfcl_amd.h declares,
and refuses to compute without a GPU (no fallback)."""
import os
import re

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_library_exports_every_declared_symbol(pkg):
    pkg.engine.build_native()
    lib = pkg.engine.dll()
    hdr = open(os.path.join(ROOT, "include", "hppfcl_amd.h")).read()
    declared = set(re.findall(r"\b(hfcl_[a-z0-9_]+)\s*\(", hdr))
    declared -= {"hfcl_lib"}  # type name
    assert len(declared) >= 19
    for sym in sorted(declared):
        assert hasattr(lib, sym), "missing export: " + sym
    assert lib.hfcl_abi_version() == 5
    assert set(pkg.engine.EXPORTED_SYMBOLS) <= declared


def test_struct_layouts_match_header(pkg):
    import ctypes as C
    abi = pkg.abi
    assert C.sizeof(abi.Shape) == 56
    assert C.sizeof(abi.QueryRequest) == 80
    assert C.sizeof(abi.CollisionRequest) == 80 + 32
    assert C.sizeof(abi.DistanceRequest) == 80 + 24
    # defaults filled by the C library equal the Python-side defaults (= the reference's)
    lib = pkg.engine.dll()
    c = abi.CollisionRequest()
    lib.hfcl_collision_request_init(C.byref(c))
    d = abi.default_collision_request()
    assert bytes(c) == bytes(d)
    c = abi.DistanceRequest()
    lib.hfcl_distance_request_init(C.byref(c))
    assert bytes(c) == bytes(abi.default_distance_request())


def test_options_are_an_api_not_an_environment(pkg):
    """hfcl_lib_set_option: the option names are enumerable without a device, a null library is refused, and the host code reads the
    environment in ONE place (the fallback loop of hfcl_lib_create) -- no getenv("HFCL_...") scattered through the batch set-up."""
    import ctypes as C
    keys = pkg.engine.option_keys()
    assert len(keys) >= 40 and len(set(keys)) == len(keys)
    assert {"bvh_walk_rounds", "bvh_coop", "cvx_w", "epa_cc_staged", "bvhd_pool", "shape_dist_pool"} <= set(keys)
    assert all(k == k.lower() and not k.startswith("hfcl_") for k in keys)
    d = pkg.engine.dll()
    assert d.hfcl_lib_set_option(None, b"bvh_coop", b"1") == pkg.abi.ERR_INVALID_ARGUMENT
    assert d.hfcl_multi_set_option(None, b"bvh_coop", b"1") == pkg.abi.ERR_INVALID_ARGUMENT
    n_getenv = 0
    for f in ("hfcl_host.hip", "hfcl_multi.hip"):
        txt = open(os.path.join(ROOT, "hpp-fcl_amd", "csrc", f)).read()
        txt = re.sub(r"//[^\n]*", "", txt)
        n_getenv += len(re.findall(r"\bgetenv\s*\(", txt))
    assert n_getenv <= 1, n_getenv
    # the forms kept only as identity references are refused by the product build (and say so through the API, not by crashing)
    assert d.hfcl_has_ab_forms() in (0, 1)
    # every option is described for integrators
    doc = open(os.path.join(ROOT, "INTEGRATION.md")).read()
    missing = [k for k in keys if ("`%s`" % k) not in doc]
    assert not missing, missing


def test_no_gpu_means_loud_failure(pkg):
    if pkg.engine.device_count() > 0:
        pytest.skip("a GPU is visible")
    L = pkg.ShapeLibrary()
    L.add_sphere(1.0)
    with pytest.raises(pkg.EngineError) as e:
        pkg.Library(L)
    assert e.value.code == pkg.abi.ERR_NO_DEVICE
    assert "no CPU fallback" in str(e.value)


def test_product_never_references_the_oracle():
    """The shipped package must not import, link or mention anything under oracle/ or tests/."""
    pkg_dir = os.path.join(ROOT, "hpp-fcl_amd")
    for dirpath, _, files in os.walk(pkg_dir):
        for f in files:
            if f.endswith((".py", ".hpp", ".hip", ".cpp", ".h", "Makefile")):
                txt = open(os.path.join(dirpath, f), errors="ignore").read()
                assert "oracle_binding" not in txt and "liboracle" not in txt and "hostsim_binding" not in txt, f
                assert not re.search(r'#include\s+"[^"]*oracle/', txt), f


def test_cpp_shim_compiles_and_fails_loudly_without_gpu(pkg):
    """include/hppfcl_amd_compat.hpp: the hpp::fcl-named shim compiles with plain g++ against the C ABI."""
    import subprocess
    pkg.engine.build_native()
    d = os.path.join(ROOT, "tests", "cpp")
    subprocess.check_call(["make", "-s", "-C", d])
    if pkg.engine.device_count() > 0:
        pytest.skip("a GPU is visible (the GPU suite runs the binary)")
    r = subprocess.run([os.path.join(d, "test_compat")], capture_output=True, text=True)
    assert r.returncode == 3 and "no CPU fallback" in r.stdout


def test_pair_supported_matches_the_reference_function_matrices(pkg):
    """hfcl_pair_supported = "is there an entry in collision_matrix / distance_matrix" (collision_func_matrix.cpp:
    244-469 for the kinds in scope): every solid x solid, flat and TriangleP row, BVHModel<OBBRSS> x itself and
    x every shape; no (BVHModel, TriangleP) and no unknown node types.  Host-only: needs no device."""
    a = pkg.abi
    f = pkg.engine.dll().hfcl_pair_supported
    solids = [a.GEOM_BOX, a.GEOM_SPHERE, a.GEOM_CAPSULE, a.GEOM_CONE, a.GEOM_CYLINDER, a.GEOM_CONVEX, a.GEOM_ELLIPSOID]
    flats = [a.GEOM_PLANE, a.GEOM_HALFSPACE]
    shapes = solids + flats + [a.GEOM_TRIANGLE]
    for dist in (0, 1):
        for s1 in shapes:
            for s2 in shapes:
                tri = a.GEOM_TRIANGLE in (s1, s2)  # src/distance_func_matrix.cpp has no TriangleP entries
                assert f(s1, s2, dist) == (0 if (dist and tri) else 1), (s1, s2, dist)
        assert f(a.BV_OBBRSS, a.BV_OBBRSS, dist) == 1
        for s in solids + flats:
            assert f(a.BV_OBBRSS, s, dist) == 1 and f(s, a.BV_OBBRSS, dist) == 1, s
        assert f(a.BV_OBBRSS, a.GEOM_TRIANGLE, dist) == 0 and f(a.GEOM_TRIANGLE, a.BV_OBBRSS, dist) == 0
        for bad in (0, 1, 8, 18, 20, 22, 300, -1):  # BV_AABB.., GEOM_OCTREE, HF_*, out of range
            assert f(bad, a.GEOM_BOX, dist) == 0 and f(a.GEOM_BOX, bad, dist) == 0, bad


def test_dispatch_equals_the_reference_function_matrices(pkg):
    """tests/golden/function_matrices.json = the entries src/collision_func_matrix.cpp / src/distance_func_matrix.cpp
    set for the node types in scope (tools/extract_function_matrices.py, run where /root/reference exists).
    hfcl_pair_supported answers what collide() / distance() accept: an entry, or -- for (GEOM, BVH) -- the entry of
    the swapped pair (src/collision.cpp:87-108, src/distance.cpp:76-92)."""
    import json
    a = pkg.abi
    f = pkg.engine.dll().hfcl_pair_supported
    g = json.load(open(os.path.join(ROOT, "tests", "golden", "function_matrices.json")))
    val = {n: getattr(a, n) for n in g["node_types"]}
    for n in g["node_types"]:  # the ABI passes NODE_TYPE values through unchanged (collision_object.h:65-89)
        assert val[n] == g["node_type_values"][n], n
    assert len(g["collision"]) == 110 and len(g["distance"]) == 91
    for mode, key in ((0, "collision"), (1, "distance")):
        ref = {(x, y) for x, y in g[key]}
        for x in g["node_types"]:
            for y in g["node_types"]:
                swap = x.startswith("GEOM_") and y.startswith("BV_")
                expect = ((y, x) in ref) if swap else ((x, y) in ref)
                assert bool(f(val[x], val[y], mode)) == expect, (key, x, y)


def test_request_defaults_equal_the_reference_headers(pkg):
    """tests/golden/request_defaults.json = default-constructor values and enum orders parsed from the reference's
    collision_data.h / narrowphase_defaults.h / data_types.h (tools/extract_request_defaults.py); the C library's
    hfcl_*_request_init must fill exactly those, and the ABI's integer codes must follow the enum orders."""
    import ctypes as C
    import json
    a = pkg.abi
    g = json.load(open(os.path.join(ROOT, "tests", "golden", "request_defaults.json")))
    lib = pkg.engine.dll()
    c = a.CollisionRequest()
    lib.hfcl_collision_request_init(C.byref(c))
    d = a.DistanceRequest()
    lib.hfcl_distance_request_init(C.byref(d))
    en = g["enums"]
    for q in (c.q, d.q):
        r = g["QueryRequest"]
        assert q.gjk_initial_guess == en["GJKInitialGuess"].index(r["gjk_initial_guess"])
        assert q.gjk_variant == en["GJKVariant"].index(r["gjk_variant"])
        assert q.gjk_convergence_criterion == en["GJKConvergenceCriterion"].index(r["gjk_convergence_criterion"])
        assert q.gjk_convergence_criterion_type == en["GJKConvergenceCriterionType"].index(r["gjk_convergence_criterion_type"])
        assert q.gjk_max_iterations == r["gjk_max_iterations"] and q.epa_max_iterations == r["epa_max_iterations"]
        assert q.gjk_tolerance == r["gjk_tolerance"] and q.epa_tolerance == r["epa_tolerance"]
        assert q.collision_distance_threshold == r["collision_distance_threshold"]
        assert list(q.cached_gjk_guess) == [float(x) for x in r["cached_gjk_guess"].split(",")]
    r = g["CollisionRequest"]
    assert c.num_max_contacts == r["num_max_contacts"] and bool(c.enable_contact) == r["enable_contact"]
    assert c.security_margin == r["security_margin"] and c.break_distance == r["break_distance"]
    assert c.distance_upper_bound == r["distance_upper_bound"]
    r = g["DistanceRequest"]
    assert bool(d.enable_nearest_points) == r["enable_nearest_points"] and bool(d.enable_signed_distance) == r["enable_signed_distance"]
    assert d.rel_err == r["rel_err"] and d.abs_err == r["abs_err"]
    # integer codes of the ABI follow the reference's enum orders
    assert [a.DefaultGJK, a.PolyakAcceleration, a.NesterovAcceleration] == [en["GJKVariant"].index(n) for n in
                                                                           ("DefaultGJK", "PolyakAcceleration", "NesterovAcceleration")]
    assert [a.Default, a.DualityGap, a.Hybrid] == [en["GJKConvergenceCriterion"].index(n) for n in ("Default", "DualityGap", "Hybrid")]


def test_graft_entry_build_runs():
    """__graft_entry__.build() is what the driver runs on CPU every round: it must pass on the tree as it is (it checks the ABI version and the
    exported symbols itself; the build is incremental, so this is seconds)."""
    import importlib.util
    spec = importlib.util.spec_from_file_location("graft_entry_under_test", os.path.join(ROOT, "__graft_entry__.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    mod.build()
