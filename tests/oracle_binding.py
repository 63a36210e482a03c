"""ctypes binding of oracle/liboracle.so -- the fp64 CPU restatement of the reference.

TEST INFRASTRUCTURE: imported only by tests/, __graft_entry__.smoke() and bench.py's
cpu_baseline leg.  The product (hpp-fcl_amd/) never imports this."""
import ctypes as C
import os
import subprocess

import numpy as np

_ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
_ORACLE_DIR = os.path.join(_ROOT, "oracle")
_LIB = None


def build():
    subprocess.check_call(["make", "-s", "-C", _ORACLE_DIR])


def lib():
    global _LIB
    if _LIB is None:
        path = os.path.join(_ORACLE_DIR, "liboracle.so")
        if not os.path.exists(path):
            build()
        _LIB = C.CDLL(path)
        _LIB.orc_project_origin.restype = C.c_uint
    return _LIB


class native_build:
    """with native_build(): ... -- the batch functions of this module run through a second build of the same sources, `-O3 -march=native`
    (BASELINE.md section 3), made on THIS host under /tmp.  For bench.py's cpu_baseline.native figure only: never a checker (its
    arithmetic contracts, the reference's default build does not)."""

    def __enter__(self):
        global _LIB
        import tempfile
        lib()  # (the default build is loaded first: it is what comes back)
        out = os.path.join(tempfile.gettempdir(), "liboracle_native_%d.so" % os.getpid())
        subprocess.check_call(["make", "-s", "-C", _ORACLE_DIR, "native", "NATIVE_OUT=" + out])
        self._old = _LIB
        _LIB = C.CDLL(out)
        _LIB.orc_project_origin.restype = C.c_uint
        self._path = out
        return self

    def __exit__(self, *exc):
        global _LIB
        _LIB = self._old
        try:
            os.remove(self._path)
        except OSError:
            pass
        return False


def _pkg():
    import importlib.util
    import sys
    if "hppfcl_amd" in sys.modules:
        return sys.modules["hppfcl_amd"]
    spec = importlib.util.spec_from_file_location(
        "hppfcl_amd", os.path.join(_ROOT, "hpp-fcl_amd", "__init__.py"),
        submodule_search_locations=[os.path.join(_ROOT, "hpp-fcl_amd")])
    mod = importlib.util.module_from_spec(spec)
    sys.modules["hppfcl_amd"] = mod
    spec.loader.exec_module(mod)
    return mod


def _batch(fn, shapes, verts, s1, s2, tf1, tf2, req, guess_in, want_guess, n_threads):
    abi = _pkg().abi
    shapes = np.ascontiguousarray(shapes)
    verts = np.ascontiguousarray(verts, dtype=np.float64)
    s1 = np.ascontiguousarray(s1, dtype=np.uint32)
    s2 = np.ascontiguousarray(s2, dtype=np.uint32)
    tf1 = np.ascontiguousarray(tf1, dtype=np.float64).reshape(-1, 12)
    tf2 = np.ascontiguousarray(tf2, dtype=np.float64).reshape(-1, 12)
    n = len(s1)
    assert len(s2) == n and len(tf1) == n and len(tf2) == n
    out = np.zeros(n, dtype=abi.RESULT_DTYPE)
    gout = np.zeros(n, dtype=abi.GUESS_DTYPE) if want_guess else None
    if guess_in is not None:
        guess_in = np.ascontiguousarray(guess_in, dtype=abi.GUESS_DTYPE)
    rc = fn(abi.ptr(shapes), C.c_size_t(len(shapes)), abi.ptr(verts), abi.ptr(s1), abi.ptr(s2), abi.ptr(tf1),
            abi.ptr(tf2), C.c_size_t(n), C.byref(req), abi.ptr(out), abi.ptr(guess_in), abi.ptr(gout),
            C.c_int(n_threads))
    return rc, out, gout


def distance_batch(shapes, verts, s1, s2, tf1, tf2, req=None, guess_in=None, want_guess=False, n_threads=1):
    abi = _pkg().abi
    req = req or abi.default_distance_request()
    rc, out, gout = _batch(lib().orc_distance_batch, shapes, verts, s1, s2, tf1, tf2, req, guess_in, want_guess,
                           n_threads)
    if rc:
        raise ValueError("oracle distance: error %d" % rc)
    return (out, gout) if want_guess else out


def collide_batch(shapes, verts, s1, s2, tf1, tf2, req=None, guess_in=None, want_guess=False, n_threads=1):
    abi = _pkg().abi
    req = req or abi.default_collision_request()
    rc, out, gout = _batch(lib().orc_collide_batch, shapes, verts, s1, s2, tf1, tf2, req, guess_in, want_guess,
                           n_threads)
    if rc:
        raise ValueError("oracle collide: error %d" % rc)
    return (out, gout) if want_guess else out


def gjk_raw(shape0, verts0, shape1, verts1, tf0, tf1, max_it=128, tol=1e-6, variant=0, criterion=0,
            criterion_type=0, upper_bound=np.finfo(np.float64).max, guess=(1, 0, 0), run_epa=False,
            epa_max_it=64, epa_tol=1e-6, epa_guess=(1, 0, 0)):
    """Raw GJK (+EPA) on one MinkowskiDiff: the level test/gjk.cpp exercises.  shapeN: 1-element SHAPE arrays."""
    abi = _pkg().abi
    s0 = np.ascontiguousarray(shape0)
    s1 = np.ascontiguousarray(shape1)
    v0 = np.ascontiguousarray(verts0 if verts0 is not None else np.zeros((1, 3)), dtype=np.float64)
    v1 = np.ascontiguousarray(verts1 if verts1 is not None else np.zeros((1, 3)), dtype=np.float64)
    tf0 = np.ascontiguousarray(tf0, dtype=np.float64)
    tf1 = np.ascontiguousarray(tf1, dtype=np.float64)
    g = np.array(guess, dtype=np.float64)
    eg = np.array(epa_guess, dtype=np.float64)
    out = np.zeros(14)
    istat = np.zeros(6, dtype=np.int32)
    lib().orc_gjk_raw(abi.ptr(s0), abi.ptr(v0), abi.ptr(s1), abi.ptr(v1), abi.ptr(tf0), abi.ptr(tf1),
                      C.c_uint(max_it), C.c_double(tol), C.c_int(variant), C.c_int(criterion),
                      C.c_int(criterion_type), C.c_double(upper_bound), abi.ptr(g), C.c_int(1 if run_epa else 0),
                      C.c_uint(epa_max_it), C.c_double(epa_tol), abi.ptr(eg), abi.ptr(out), abi.ptr(istat))
    return dict(w0=out[0:3].copy(), w1=out[3:6].copy(), normal=out[6:9].copy(), distance=out[9],
                ray=out[10:13].copy(), epa_depth=out[13], gjk_status=int(istat[0]), gjk_iterations=int(istat[1]),
                epa_status=int(istat[2]), epa_iterations=int(istat[3]), rank=int(istat[4]),
                momentum_stop=int(istat[5]))


def project_origin(points):
    abi = _pkg().abi
    pts = np.ascontiguousarray(points, dtype=np.float64).reshape(-1, 3)
    out = np.zeros(5)
    enc = lib().orc_project_origin(C.c_int(len(pts)), abi.ptr(pts), abi.ptr(out))
    return dict(param=out[:4].copy(), sqr_distance=out[4], encode=int(enc))


def shape_support(shape, verts, direction):
    abi = _pkg().abi
    s = np.ascontiguousarray(shape)
    v = np.ascontiguousarray(verts if verts is not None else np.zeros((1, 3)), dtype=np.float64)
    d = np.array(direction, dtype=np.float64)
    out = np.zeros(3)
    hint = C.c_int(0)
    lib().orc_shape_support(abi.ptr(s), abi.ptr(v), abi.ptr(d), abi.ptr(out), C.byref(hint))
    return out, hint.value


def bvh_collide_batch(meshlib, m1, m2, tf1, tf2, req=None, max_contacts=0, n_threads=1, want_stats=False):
    """BVHModel<OBBRSS> x BVHModel<OBBRSS> collide() through the oracle.  meshlib: bvh_builder.MeshLibrary."""
    abi = _pkg().abi
    req = req or abi.default_collision_request()
    m1 = np.ascontiguousarray(m1, dtype=np.uint32)
    m2 = np.ascontiguousarray(m2, dtype=np.uint32)
    tf1 = np.ascontiguousarray(tf1, dtype=np.float64).reshape(-1, 12)
    tf2 = np.ascontiguousarray(tf2, dtype=np.float64).reshape(-1, 12)
    n = len(m1)
    out = np.zeros(n, dtype=abi.RESULT_DTYPE)
    stats = np.zeros((n, 2), dtype=np.uint32)
    contacts = np.zeros(max(1, max_contacts), dtype=abi.CONTACT_DTYPE)
    nc = C.c_size_t(0)
    nodes = np.ascontiguousarray(meshlib.nodes)
    rc = lib().orc_bvh_collide_batch(abi.ptr(nodes), abi.ptr(meshlib.verts), abi.ptr(meshlib.tris),
                                     abi.ptr(meshlib.table), C.c_size_t(len(meshlib.table)), abi.ptr(m1), abi.ptr(m2),
                                     abi.ptr(tf1), abi.ptr(tf2), C.c_size_t(n), C.byref(req), abi.ptr(out),
                                     abi.ptr(stats), abi.ptr(contacts) if max_contacts else None,
                                     C.c_size_t(max_contacts), C.byref(nc), C.c_int(n_threads))
    if rc:
        raise ValueError("oracle bvh collide: error %d" % rc)
    res = [out]
    if max_contacts:
        res.append(contacts[:nc.value])
    if want_stats:
        res.append(stats)
    return res[0] if len(res) == 1 else tuple(res)


def bvh_distance_batch(meshlib, m1, m2, tf1, tf2, n_threads=1, want_stats=False):
    """BVHModel<OBBRSS> x BVHModel<OBBRSS> distance() through the oracle."""
    abi = _pkg().abi
    m1 = np.ascontiguousarray(m1, dtype=np.uint32)
    m2 = np.ascontiguousarray(m2, dtype=np.uint32)
    tf1 = np.ascontiguousarray(tf1, dtype=np.float64).reshape(-1, 12)
    tf2 = np.ascontiguousarray(tf2, dtype=np.float64).reshape(-1, 12)
    n = len(m1)
    out = np.zeros(n, dtype=abi.RESULT_DTYPE)
    stats = np.zeros((n, 2), dtype=np.uint32)
    nodes = np.ascontiguousarray(meshlib.nodes)
    lib().orc_bvh_distance_batch(abi.ptr(nodes), abi.ptr(meshlib.verts), abi.ptr(meshlib.tris), abi.ptr(meshlib.table),
                                 C.c_size_t(len(meshlib.table)), abi.ptr(m1), abi.ptr(m2), abi.ptr(tf1), abi.ptr(tf2),
                                 C.c_size_t(n), abi.ptr(out), abi.ptr(stats), C.c_int(n_threads))
    return (out, stats) if want_stats else out


def bvh_leaf_distance(meshlib, m1, m2, tf1, tf2, pid1, pid2):
    """The distance the oracle's traversal assigns to triangle pair (pid1, pid2) of the query (mesh m1 at tf1, mesh m2 at tf2)."""
    abi = _pkg().abi
    f = lib().orc_bvh_leaf_distance
    f.restype = C.c_double
    nodes = np.ascontiguousarray(meshlib.nodes)
    t1 = np.ascontiguousarray(tf1, dtype=np.float64).reshape(12)
    t2 = np.ascontiguousarray(tf2, dtype=np.float64).reshape(12)
    return f(abi.ptr(nodes), abi.ptr(meshlib.verts), abi.ptr(meshlib.tris), abi.ptr(meshlib.table), C.c_size_t(len(meshlib.table)),
             C.c_uint32(int(m1)), C.c_uint32(int(m2)), abi.ptr(t1), abi.ptr(t2), C.c_int(int(pid1)), C.c_int(int(pid2)))


def mixed_leaf_distance(shapes, verts, meshlib, s1, s2, tf1, tf2, pid, req=None):
    """The distance the oracle's mesh x solid traversal assigns to triangle `pid` of the query (shape s1 at tf1, shape s2 at tf2; one of
    them a mesh), with the request's solver settings and the default guess."""
    abi = _pkg().abi
    req = req or abi.default_distance_request()
    f = lib().orc_mixed_leaf_distance
    f.restype = C.c_double
    shapes = np.ascontiguousarray(shapes)
    verts = np.ascontiguousarray(verts, dtype=np.float64)
    nodes = np.ascontiguousarray(meshlib.nodes)
    t1 = np.ascontiguousarray(tf1, dtype=np.float64).reshape(12)
    t2 = np.ascontiguousarray(tf2, dtype=np.float64).reshape(12)
    return f(abi.ptr(shapes), C.c_size_t(len(shapes)), abi.ptr(verts), abi.ptr(nodes), abi.ptr(meshlib.verts), abi.ptr(meshlib.tris),
             abi.ptr(meshlib.table), C.c_size_t(len(meshlib.table)), C.c_uint32(int(s1)), C.c_uint32(int(s2)), abi.ptr(t1), abi.ptr(t2),
             C.byref(req), C.c_int(int(pid)))


def rect_distance(Rab, Tab, a, b):
    abi = _pkg().abi
    L = lib()
    L.orc_rect_distance.restype = C.c_double
    R = np.ascontiguousarray(Rab, dtype=np.float64)
    T = np.ascontiguousarray(Tab, dtype=np.float64)
    a = np.ascontiguousarray(a, dtype=np.float64)
    b = np.ascontiguousarray(b, dtype=np.float64)
    return L.orc_rect_distance(abi.ptr(R), abi.ptr(T), abi.ptr(a), abi.ptr(b))


def sqr_tri_distance(S, T):
    abi = _pkg().abi
    L = lib()
    L.orc_sqr_tri_distance.restype = C.c_double
    S = np.ascontiguousarray(S, dtype=np.float64)
    T = np.ascontiguousarray(T, dtype=np.float64)
    out = np.zeros(6)
    d2 = L.orc_sqr_tri_distance(abi.ptr(S), abi.ptr(T), abi.ptr(out))
    return d2, out[:3].copy(), out[3:].copy()


def bvh_build(vertices, triangles):
    """Oracle restatement of BVHModel<OBBRSS>::endModel() (oracle/bvh_build.cpp)."""
    abi = _pkg().abi
    v = np.ascontiguousarray(vertices, dtype=np.float64).reshape(-1, 3)
    t = np.ascontiguousarray(triangles, dtype=np.uint32).reshape(-1, 3)
    nodes = np.zeros(2 * len(t) - 1, dtype=abi.BVH_NODE_DTYPE)
    prim = np.zeros(len(t), dtype=np.uint32)
    rc = lib().orc_bvh_build(C.c_void_p(v.ctypes.data), C.c_size_t(len(v)), C.c_void_p(t.ctypes.data),
                             C.c_size_t(len(t)), C.c_void_p(nodes.ctypes.data), C.c_void_p(prim.ctypes.data))
    assert rc == 0
    return nodes, prim


def world_aabbs(shapes, verts, obj_shape, obj_tf):
    """Oracle restatement of CollisionObject::computeAABB over posed shapes (oracle/broadphase.cpp)."""
    shapes = np.ascontiguousarray(shapes)
    verts = np.ascontiguousarray(verts, dtype=np.float64)
    ids = np.ascontiguousarray(obj_shape, dtype=np.uint32)
    tf = np.ascontiguousarray(obj_tf, dtype=np.float64).reshape(-1, 12)
    out = np.zeros((len(ids), 6))
    lib().orc_world_aabbs(C.c_void_p(shapes.ctypes.data), C.c_void_p(verts.ctypes.data), C.c_void_p(ids.ctypes.data),
                          C.c_void_p(tf.ctypes.data), C.c_size_t(len(ids)), C.c_void_p(out.ctypes.data))
    return out


def bruteforce_pairs(aabbs):
    a = np.ascontiguousarray(aabbs, dtype=np.float64).reshape(-1, 6)
    fn = lib().orc_bruteforce_pairs
    fn.restype = C.c_size_t
    n = fn(C.c_void_p(a.ctypes.data), C.c_size_t(len(a)), None, C.c_size_t(0))
    out = np.zeros((n, 2), dtype=np.uint32)
    fn(C.c_void_p(a.ctypes.data), C.c_size_t(len(a)), C.c_void_p(out.ctypes.data), C.c_size_t(n))
    return out


def register_hull_neighbors(shapes, verts, graphs=None):
    """ConvexBase::neighbors for every hull of >= 32 vertices, as Convex<Triangle>::fillNeighbors builds them
    (details/convex.hxx:231-280: per vertex the ascending set of vertices sharing a face edge) from the facets
    scipy's Qhull wrapper reports -- the oracle's hill-climbing support (getShapeSupportLog) walks them.
    graphs: {shape index: (offsets, ids)} -- adjacency given by the caller (a triangulated surface whose points are not
    all extreme, which Qhull would not report) instead of the Qhull facets."""
    from scipy.spatial import ConvexHull
    abi = _pkg().abi
    L = lib()
    L.orc_clear_neighbors()
    verts = np.ascontiguousarray(verts, dtype=np.float64).reshape(-1, 3)
    keep = []
    for si, s in enumerate(shapes):
        if s["type"] != abi.GEOM_CONVEX or s["num_points"] < 32:
            continue
        off, n = int(s["vertex_offset"]), int(s["num_points"])
        if graphs is not None and si in graphs:
            offs = np.ascontiguousarray(graphs[si][0], dtype=np.uint32)
            ids = np.ascontiguousarray(graphs[si][1], dtype=np.uint32)
            keep.append((offs, ids))
            L.orc_register_neighbors(C.c_uint32(off), C.c_void_p(offs.ctypes.data), C.c_uint32(n), C.c_void_p(ids.ctypes.data))
            continue
        hull = ConvexHull(verts[off:off + n])
        assert len(hull.vertices) == n, "every point must be a hull vertex"
        nb = [set() for _ in range(n)]
        for tri in hull.simplices:
            for j in range(3):
                a, b = int(tri[j]), int(tri[(j + 1) % 3])
                nb[a].add(b)
                nb[b].add(a)
        offs = np.zeros(n + 1, dtype=np.uint32)
        offs[1:] = np.cumsum([len(x) for x in nb])
        ids = np.array([v for x in nb for v in sorted(x)], dtype=np.uint32)
        keep.append((offs, ids))
        L.orc_register_neighbors(C.c_uint32(off), C.c_void_p(offs.ctypes.data), C.c_uint32(n), C.c_void_p(ids.ctypes.data))
    return len(keep)


def mixed_collide_batch(shapes, verts, meshlib, s1, s2, tf1, tf2, req=None, max_contacts=0, n_threads=1, want_guess=False,
                        want_stats=False):
    """collide() over a shape table that mixes BVHModel<OBBRSS> entries (bvh_index -> meshlib) and convex
    shapes: mesh x mesh, mesh x shape, shape x mesh, shape x shape (oracle/capi.cpp orc_mixed_collide_batch)."""
    abi = _pkg().abi
    req = req or abi.default_collision_request()
    shapes = np.ascontiguousarray(shapes)
    verts = np.ascontiguousarray(verts, dtype=np.float64)
    s1 = np.ascontiguousarray(s1, dtype=np.uint32)
    s2 = np.ascontiguousarray(s2, dtype=np.uint32)
    tf1 = np.ascontiguousarray(tf1, dtype=np.float64).reshape(-1, 12)
    tf2 = np.ascontiguousarray(tf2, dtype=np.float64).reshape(-1, 12)
    n = len(s1)
    out = np.zeros(n, dtype=abi.RESULT_DTYPE)
    gout = np.zeros(n, dtype=abi.GUESS_DTYPE) if want_guess else None
    contacts = np.zeros(max(1, max_contacts), dtype=abi.CONTACT_DTYPE)
    nc = C.c_size_t(0)
    nodes = np.ascontiguousarray(meshlib.nodes)
    stats = np.zeros((n, 2), dtype=np.uint32) if want_stats else None  # (num_bv_tests, num_leaf_tests) per pair
    rc = lib().orc_mixed_collide_batch_stats(abi.ptr(shapes), C.c_size_t(len(shapes)), abi.ptr(verts), abi.ptr(nodes),
                                             abi.ptr(meshlib.verts), abi.ptr(meshlib.tris), abi.ptr(meshlib.table),
                                             C.c_size_t(len(meshlib.table)), abi.ptr(s1), abi.ptr(s2), abi.ptr(tf1), abi.ptr(tf2),
                                             C.c_size_t(n), C.byref(req), abi.ptr(out), abi.ptr(gout),
                                             abi.ptr(contacts) if max_contacts else None, C.c_size_t(max_contacts), C.byref(nc),
                                             C.c_int(n_threads), abi.ptr(stats))
    if rc:
        raise ValueError("oracle mixed collide: error %d" % rc)
    res = [out]
    if max_contacts:
        res.append(contacts[:nc.value])
    if want_guess:
        res.append(gout)
    if want_stats:
        res.append(stats)
    return res[0] if len(res) == 1 else tuple(res)


def mixed_distance_batch(shapes, verts, meshlib, s1, s2, tf1, tf2, req=None, n_threads=1, want_guess=False):
    """distance() counterpart of mixed_collide_batch."""
    abi = _pkg().abi
    req = req or abi.default_distance_request()
    shapes = np.ascontiguousarray(shapes)
    verts = np.ascontiguousarray(verts, dtype=np.float64)
    s1 = np.ascontiguousarray(s1, dtype=np.uint32)
    s2 = np.ascontiguousarray(s2, dtype=np.uint32)
    tf1 = np.ascontiguousarray(tf1, dtype=np.float64).reshape(-1, 12)
    tf2 = np.ascontiguousarray(tf2, dtype=np.float64).reshape(-1, 12)
    n = len(s1)
    out = np.zeros(n, dtype=abi.RESULT_DTYPE)
    gout = np.zeros(n, dtype=abi.GUESS_DTYPE) if want_guess else None
    nodes = np.ascontiguousarray(meshlib.nodes)
    rc = lib().orc_mixed_distance_batch(abi.ptr(shapes), C.c_size_t(len(shapes)), abi.ptr(verts), abi.ptr(nodes),
                                        abi.ptr(meshlib.verts), abi.ptr(meshlib.tris), abi.ptr(meshlib.table),
                                        C.c_size_t(len(meshlib.table)), abi.ptr(s1), abi.ptr(s2), abi.ptr(tf1), abi.ptr(tf2),
                                        C.c_size_t(n), C.byref(req), abi.ptr(out), abi.ptr(gout), C.c_int(n_threads))
    if rc:
        raise ValueError("oracle mixed distance: error %d" % rc)
    return (out, gout) if want_guess else out
