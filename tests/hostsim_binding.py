"""ctypes binding of tests/hostsim/libhostsim.so: the device per-pair headers compiled for the
host (TEST INFRASTRUCTURE, see tests/hostsim/hostsim.cpp).  Never used by the product."""
import ctypes as C
import os
import subprocess

import numpy as np

_DIR = os.path.join(os.path.dirname(os.path.abspath(__file__)), "hostsim")
_LIB = None


def lib():
    global _LIB
    if _LIB is None:
        subprocess.check_call(["make", "-s", "-C", _DIR])
        _LIB = C.CDLL(os.path.join(_DIR, "libhostsim.so"))
    return _LIB


def batch_f64(abi, shapes, verts, s1, s2, tf1, tf2, req, guess_in=None, want_guess=False):
    shapes = np.ascontiguousarray(shapes)
    verts = np.ascontiguousarray(verts, dtype=np.float64)
    s1 = np.ascontiguousarray(s1, dtype=np.uint32)
    s2 = np.ascontiguousarray(s2, dtype=np.uint32)
    tf1 = np.ascontiguousarray(tf1, dtype=np.float64).reshape(-1, 12)
    tf2 = np.ascontiguousarray(tf2, dtype=np.float64).reshape(-1, 12)
    n = len(s1)
    out = np.zeros(n, dtype=abi.RESULT_DTYPE)
    gout = np.zeros(n, dtype=abi.GUESS_DTYPE) if want_guess else None
    is_coll = isinstance(req, abi.CollisionRequest)
    lib().sim_batch_f64(abi.ptr(shapes), C.c_size_t(len(shapes)), abi.ptr(verts), abi.ptr(s1), abi.ptr(s2),
                        abi.ptr(tf1), abi.ptr(tf2), C.c_size_t(n), C.byref(req) if is_coll else None,
                        None if is_coll else C.byref(req), abi.ptr(out), abi.ptr(guess_in), abi.ptr(gout))
    return (out, gout) if want_guess else out


def batch_f32(abi, shapes, verts, s1, s2, pose1, pose2, req):
    shapes = np.ascontiguousarray(shapes)
    verts = np.ascontiguousarray(verts, dtype=np.float64)
    s1 = np.ascontiguousarray(s1, dtype=np.uint32)
    s2 = np.ascontiguousarray(s2, dtype=np.uint32)
    pose1 = np.ascontiguousarray(pose1, dtype=np.float32).reshape(-1, 7)
    pose2 = np.ascontiguousarray(pose2, dtype=np.float32).reshape(-1, 7)
    n = len(s1)
    out = np.zeros(n, dtype=abi.RESULT_F32_DTYPE)
    is_coll = isinstance(req, abi.CollisionRequest)
    lib().sim_batch_f32(abi.ptr(shapes), C.c_size_t(len(shapes)), abi.ptr(verts), C.c_size_t(len(verts)),
                        abi.ptr(s1), abi.ptr(s2), abi.ptr(pose1), abi.ptr(pose2), C.c_size_t(n),
                        C.byref(req) if is_coll else None, None if is_coll else C.byref(req), abi.ptr(out))
    return out


def bvh_collide_f64(abi, meshlib, m1, m2, tf1, tf2, req, max_contacts=0):
    m1 = np.ascontiguousarray(m1, dtype=np.uint32)
    m2 = np.ascontiguousarray(m2, dtype=np.uint32)
    tf1 = np.ascontiguousarray(tf1, dtype=np.float64).reshape(-1, 12)
    tf2 = np.ascontiguousarray(tf2, dtype=np.float64).reshape(-1, 12)
    n = len(m1)
    out = np.zeros(n, dtype=abi.RESULT_DTYPE)
    contacts = np.zeros(max(1, max_contacts), dtype=abi.CONTACT_DTYPE)
    nc = C.c_size_t(0)
    nodes = np.ascontiguousarray(meshlib.nodes)
    lib().sim_bvh_collide_f64(abi.ptr(nodes), C.c_size_t(len(nodes)), abi.ptr(meshlib.verts),
                              C.c_size_t(len(meshlib.verts)), abi.ptr(meshlib.tris), abi.ptr(meshlib.table),
                              abi.ptr(m1), abi.ptr(m2), abi.ptr(tf1), abi.ptr(tf2), C.c_size_t(n), C.byref(req),
                              abi.ptr(out), abi.ptr(contacts) if max_contacts else None, C.c_size_t(max_contacts),
                              C.byref(nc))
    if max_contacts:
        return out, contacts[:min(nc.value, max_contacts)]
    return out


def bvh_collide_filtered_f64(abi, meshlib, m1, m2, tf1, tf2, req):
    """The walk of k_bvh_collide with the fp32 separating-axis filter in front of the fp64 test; returns (records, stats):
    stats = dict of the 7 counters of hostsim.cpp: bvh_pair (unsafe / rank_mismatch must be 0)."""
    m1 = np.ascontiguousarray(m1, dtype=np.uint32)
    m2 = np.ascontiguousarray(m2, dtype=np.uint32)
    tf1 = np.ascontiguousarray(tf1, dtype=np.float64).reshape(-1, 12)
    tf2 = np.ascontiguousarray(tf2, dtype=np.float64).reshape(-1, 12)
    n = len(m1)
    out = np.zeros(n, dtype=abi.RESULT_DTYPE)
    st = np.zeros(7, dtype=np.uint64)
    nodes = np.ascontiguousarray(meshlib.nodes)
    lib().sim_bvh_collide_filtered_f64(abi.ptr(nodes), C.c_size_t(len(nodes)), abi.ptr(meshlib.verts),
                                       C.c_size_t(len(meshlib.verts)), abi.ptr(meshlib.tris), abi.ptr(meshlib.table),
                                       abi.ptr(m1), abi.ptr(m2), abi.ptr(tf1), abi.ptr(tf2), C.c_size_t(n), C.byref(req),
                                       abi.ptr(out), abi.ptr(st))
    keys = ["bv_tests", "overlap", "disjoint_skipped", "disjoint_value_needed", "unsure", "unsafe", "rank_mismatch"]
    return out, dict(zip(keys, [int(x) for x in st]))


def bvh_distance_f64(abi, meshlib, m1, m2, tf1, tf2):
    m1 = np.ascontiguousarray(m1, dtype=np.uint32)
    m2 = np.ascontiguousarray(m2, dtype=np.uint32)
    tf1 = np.ascontiguousarray(tf1, dtype=np.float64).reshape(-1, 12)
    tf2 = np.ascontiguousarray(tf2, dtype=np.float64).reshape(-1, 12)
    n = len(m1)
    out = np.zeros(n, dtype=abi.RESULT_DTYPE)
    nodes = np.ascontiguousarray(meshlib.nodes)
    lib().sim_bvh_distance_f64(abi.ptr(nodes), C.c_size_t(len(nodes)), abi.ptr(meshlib.verts),
                               C.c_size_t(len(meshlib.verts)), abi.ptr(meshlib.tris), abi.ptr(meshlib.table),
                               abi.ptr(m1), abi.ptr(m2), abi.ptr(tf1), abi.ptr(tf2), C.c_size_t(n), abi.ptr(out))
    return out


def rect_distance(abi, Rab, Tab, a, b):
    L = lib()
    L.sim_rect_distance.restype = C.c_double
    R = np.ascontiguousarray(Rab, dtype=np.float64)
    T = np.ascontiguousarray(Tab, dtype=np.float64)
    a = np.ascontiguousarray(a, dtype=np.float64)
    b = np.ascontiguousarray(b, dtype=np.float64)
    return L.sim_rect_distance(abi.ptr(R), abi.ptr(T), abi.ptr(a), abi.ptr(b))


def sqr_tri_distance(abi, S, T):
    L = lib()
    L.sim_sqr_tri_distance.restype = C.c_double
    S = np.ascontiguousarray(S, dtype=np.float64)
    T = np.ascontiguousarray(T, dtype=np.float64)
    out = np.zeros(6)
    return L.sim_sqr_tri_distance(abi.ptr(S), abi.ptr(T), abi.ptr(out)), out[:3].copy(), out[3:].copy()


def mesh_shape_collide_f64(abi, shapes, verts, meshlib, s1, s2, tf1, tf2, req, max_contacts=0, want_guess=False):
    """BVHModel<OBBRSS> x convex shape pairs (either order) through the device header's mesh_shape_collide."""
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
    lib().sim_mesh_shape_collide_f64(abi.ptr(shapes), C.c_size_t(len(shapes)), abi.ptr(verts), abi.ptr(nodes),
                                     C.c_size_t(len(nodes)), abi.ptr(meshlib.verts), C.c_size_t(len(meshlib.verts)),
                                     abi.ptr(meshlib.tris), abi.ptr(meshlib.table), abi.ptr(s1), abi.ptr(s2), abi.ptr(tf1),
                                     abi.ptr(tf2), C.c_size_t(n), C.byref(req), abi.ptr(out), abi.ptr(gout),
                                     abi.ptr(contacts) if max_contacts else None, C.c_size_t(max_contacts), C.byref(nc))
    res = [out]
    if max_contacts:
        res.append(contacts[:min(nc.value, max_contacts)])
    if want_guess:
        res.append(gout)
    return res[0] if len(res) == 1 else tuple(res)


def set_shape_lane(on):
    """1: the mesh x solid sims take the one-query-per-lane forms (k_bvh_collide<SOLID> / k_bvh_shape_finish) where the request admits them."""
    lib().sim_set_shape_lane(C.c_int(1 if on else 0))


def mesh_shape_distance_f64(abi, shapes, verts, meshlib, s1, s2, tf1, tf2, req, want_guess=False):
    """distance() of BVHModel<OBBRSS> x convex shape pairs through the device header's mesh_shape_distance."""
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
    lib().sim_mesh_shape_distance_f64(abi.ptr(shapes), C.c_size_t(len(shapes)), abi.ptr(verts), abi.ptr(nodes),
                                      C.c_size_t(len(nodes)), abi.ptr(meshlib.verts), C.c_size_t(len(meshlib.verts)),
                                      abi.ptr(meshlib.tris), abi.ptr(meshlib.table), abi.ptr(s1), abi.ptr(s2), abi.ptr(tf1),
                                      abi.ptr(tf2), C.c_size_t(n), C.byref(req), abi.ptr(out), abi.ptr(gout))
    return (out, gout) if want_guess else out
