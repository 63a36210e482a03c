"""ctypes mirror of include/hppfcl_amd.h (the C ABI).  Plumbing only: no compute here."""
import ctypes as C

import numpy as np

# geometry kinds = hpp-fcl NODE_TYPE values (include/hpp/fcl/collision_object.h:65-89)
BV_OBBRSS = 5
GEOM_BOX = 9
GEOM_SPHERE = 10
GEOM_CAPSULE = 11
GEOM_CONE = 12
GEOM_CYLINDER = 13
GEOM_CONVEX = 14
GEOM_PLANE = 15
GEOM_HALFSPACE = 16
GEOM_TRIANGLE = 17
GEOM_ELLIPSOID = 19

# include/hpp/fcl/data_types.h:85-98
DefaultGuess, CachedGuess, BoundingVolumeGuess = 0, 1, 2
DefaultGJK, PolyakAcceleration, NesterovAcceleration = 0, 1, 2
Default, DualityGap, Hybrid = 0, 1, 2
Relative, Absolute = 0, 1

# GJK::Status / EPA::Status (include/hpp/fcl/narrowphase/gjk.h:95-102, 330-341)
GJK_DidNotRun, GJK_Failed, GJK_NoCollisionEarlyStopped, GJK_NoCollision = 0, 1, 2, 3
GJK_CollisionWithPenetrationInformation, GJK_Collision = 4, 5
EPA_Failed, EPA_Valid, EPA_AccuracyReached, EPA_Degenerated = 0, 1, 3, 2
EPA_NonConvex, EPA_InvalidHull, EPA_OutOfFaces, EPA_OutOfVertices, EPA_FallBack = 4, 6, 8, 10, 12
EPA_DidNotRun = 15

OK, ERR_INVALID_ARGUMENT, ERR_UNSUPPORTED_PAIR, ERR_NO_DEVICE, ERR_HIP, ERR_LIMIT = 0, 1, 2, 3, 4, 5


class Shape(C.Structure):
    _fields_ = [("type", C.c_int32), ("num_points", C.c_uint32), ("vertex_offset", C.c_uint32),
                ("bvh_index", C.c_uint32), ("params", C.c_double * 4), ("swept_sphere_radius", C.c_double)]


class QueryRequest(C.Structure):
    _fields_ = [("gjk_initial_guess", C.c_int32), ("gjk_variant", C.c_int32),
                ("gjk_convergence_criterion", C.c_int32), ("gjk_convergence_criterion_type", C.c_int32),
                ("gjk_max_iterations", C.c_uint32), ("epa_max_iterations", C.c_uint32),
                ("gjk_tolerance", C.c_double), ("epa_tolerance", C.c_double),
                ("collision_distance_threshold", C.c_double), ("cached_gjk_guess", C.c_double * 3),
                ("cached_support_func_guess", C.c_int32 * 2)]


class CollisionRequest(C.Structure):
    _fields_ = [("q", QueryRequest), ("num_max_contacts", C.c_uint32), ("enable_contact", C.c_int32),
                ("security_margin", C.c_double), ("break_distance", C.c_double),
                ("distance_upper_bound", C.c_double)]


class DistanceRequest(C.Structure):
    _fields_ = [("q", QueryRequest), ("enable_nearest_points", C.c_int32), ("enable_signed_distance", C.c_int32),
                ("rel_err", C.c_double), ("abs_err", C.c_double)]


class BvhNode(C.Structure):
    _fields_ = [("first_child", C.c_int32), ("first_primitive", C.c_int32), ("num_primitives", C.c_int32),
                ("_pad", C.c_int32), ("obb_axes", C.c_double * 9), ("obb_To", C.c_double * 3),
                ("obb_extent", C.c_double * 3), ("rss_axes", C.c_double * 9), ("rss_Tr", C.c_double * 3),
                ("rss_length", C.c_double * 2), ("rss_radius", C.c_double)]


# numpy views of the record types (same memory layout as the C structs)
RESULT_DTYPE = np.dtype([("distance", "<f8"), ("normal", "<f8", 3), ("p1", "<f8", 3), ("p2", "<f8", 3),
                         ("b1", "<i4"), ("b2", "<i4"), ("status", "<u4"), ("num_contacts", "<i4")])
assert RESULT_DTYPE.itemsize == 96
RESULT_F32_DTYPE = np.dtype([("distance", "<f4"), ("p1", "<f4", 3), ("p2", "<f4", 3), ("normal", "<f4", 3),
                             ("status", "<u4")])
assert RESULT_F32_DTYPE.itemsize == 44
RESULT_COMPACT_DTYPE = np.dtype([("distance", "<f8"), ("b1", "<i4"), ("b2", "<i4"), ("status", "<u4"), ("num_contacts", "<i4")])
assert RESULT_COMPACT_DTYPE.itemsize == 24
RESULT_COMPACT_F32_DTYPE = np.dtype([("distance", "<f4"), ("status", "<u4")])
assert RESULT_COMPACT_F32_DTYPE.itemsize == 8
GUESS_DTYPE = np.dtype([("gjk_guess", "<f8", 3), ("support_guess", "<i4", 2)])
assert GUESS_DTYPE.itemsize == 32
CONTACT_DTYPE = np.dtype([("pair", "<u4"), ("b1", "<i4"), ("b2", "<i4"), ("_pad", "<u4"),
                          ("penetration_depth", "<f8"), ("normal", "<f8", 3), ("p1", "<f8", 3), ("p2", "<f8", 3)])
assert CONTACT_DTYPE.itemsize == 96
SHAPE_DTYPE = np.dtype([("type", "<i4"), ("num_points", "<u4"), ("vertex_offset", "<u4"), ("bvh_index", "<u4"),
                        ("params", "<f8", 4), ("swept_sphere_radius", "<f8")])
assert SHAPE_DTYPE.itemsize == C.sizeof(Shape) == 56
BVH_NODE_DTYPE = np.dtype([("first_child", "<i4"), ("first_primitive", "<i4"), ("num_primitives", "<i4"),
                           ("_pad", "<i4"), ("obb_axes", "<f8", 9), ("obb_To", "<f8", 3), ("obb_extent", "<f8", 3),
                           ("rss_axes", "<f8", 9), ("rss_Tr", "<f8", 3), ("rss_length", "<f8", 2),
                           ("rss_radius", "<f8")])
assert BVH_NODE_DTYPE.itemsize == C.sizeof(BvhNode) == 256


def compact_records(records):
    """Host-side image of hfcl_compact_results_device: the fields a compact record keeps (bit copies)."""
    f32 = records.dtype == RESULT_F32_DTYPE
    out = np.zeros(len(records), dtype=RESULT_COMPACT_F32_DTYPE if f32 else RESULT_COMPACT_DTYPE)
    for k in out.dtype.names:
        out[k] = records[k]
    return out


def status_gjk(s):
    return np.asarray(s) & 7


def status_epa(s):
    return (np.asarray(s) >> 3) & 15


def status_contact(s):
    return (np.asarray(s) >> 7) & 1


def status_gjk_iters(s):
    return (np.asarray(s) >> 8) & 255


def status_epa_iters(s):
    return (np.asarray(s) >> 16) & 127


def status_skipped(s):
    return (np.asarray(s) >> 31) & 1


def default_distance_request():
    """DistanceRequest defaults, include/hpp/fcl/collision_data.h:171-237, 987-1031."""
    r = DistanceRequest()
    _query_defaults(r.q)
    r.enable_nearest_points = 1
    r.enable_signed_distance = 1
    r.rel_err = 0.0
    r.abs_err = 0.0
    return r


def default_collision_request():
    """CollisionRequest defaults, include/hpp/fcl/collision_data.h:171-237, 312-366."""
    r = CollisionRequest()
    _query_defaults(r.q)
    r.num_max_contacts = 1
    r.enable_contact = 1
    r.security_margin = 0.0
    r.break_distance = 1e-3
    r.distance_upper_bound = np.finfo(np.float64).max
    return r


def _query_defaults(q):
    q.gjk_initial_guess = DefaultGuess
    q.gjk_variant = DefaultGJK
    q.gjk_convergence_criterion = Default
    q.gjk_convergence_criterion_type = Relative
    q.gjk_max_iterations = 128
    q.epa_max_iterations = 64
    q.gjk_tolerance = 1e-6
    q.epa_tolerance = 1e-6
    q.collision_distance_threshold = 1e-12
    q.cached_gjk_guess[0], q.cached_gjk_guess[1], q.cached_gjk_guess[2] = 1.0, 0.0, 0.0
    q.cached_support_func_guess[0] = q.cached_support_func_guess[1] = 0


def ptr(a, ctype=C.c_void_p):
    """numpy array -> ctypes pointer (or None)."""
    if a is None:
        return None
    return a.ctypes.data_as(ctype)
