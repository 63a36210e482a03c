"""ctypes front-end of csrc/libhppfcl_amd.so (the C ABI of include/hppfcl_amd.h).

Plumbing only: argument marshalling and device-pointer hand-off.  All compute is in the HIP
library.  There is NO fallback: a missing library or a missing GPU raises."""
import ctypes as C
import os
import subprocess

import numpy as np

from . import abi

_HERE = os.path.dirname(os.path.abspath(__file__))
_CSRC = os.path.join(_HERE, "csrc")
LIB_PATH = os.environ.get("HFCL_LIB_PATH", os.path.join(_CSRC, "libhppfcl_amd.so"))  # override: A/B builds

EXPORTED_SYMBOLS = [
    "hfcl_abi_version", "hfcl_device_count", "hfcl_last_error", "hfcl_collision_request_init",
    "hfcl_distance_request_init", "hfcl_lib_create", "hfcl_lib_destroy", "hfcl_lib_num_shapes", "hfcl_lib_device", "hfcl_lib_climb_min",
    "hfcl_lib_add_bvh", "hfcl_collide_batch", "hfcl_distance_batch", "hfcl_collide_batch_device",
    "hfcl_distance_batch_device", "hfcl_distance_batch_device_f32", "hfcl_collide_batch_device_f32", "hfcl_collide_batch_f32", "hfcl_distance_batch_f32",
    "hfcl_collide_batch_contacts", "hfcl_last_kernel_ms", "hfcl_last_kernel_name", "hfcl_bvh_build",
    "hfcl_world_aabbs", "hfcl_broadphase_self_pairs", "hfcl_broadphase_pairs_between", "hfcl_pairlist_size",
    "hfcl_pairlist_data", "hfcl_pairlist_free", "hfcl_lib_set_kernel_timing", "hfcl_pair_supported", "hfcl_last_kernel_breakdown", "hfcl_last_bucket_counts", "hfcl_last_ordered_reruns", "hfcl_lib_set_split", "hfcl_lib_get_split", "hfcl_lib_last_split_parts",
    "hfcl_collide_batch_qt", "hfcl_distance_batch_qt", "hfcl_lib_set_host_chunk", "hfcl_lib_set_shapes",
    "hfcl_lib_set_convex_neighbors", "hfcl_compact_results_device", "hfcl_compact_results_device_f32",
 "hfcl_shard_range", "hfcl_multi_create", "hfcl_multi_destroy", "hfcl_multi_size", "hfcl_multi_replica",
    "hfcl_multi_set_shapes", "hfcl_multi_set_convex_neighbors", "hfcl_multi_add_bvh", "hfcl_collide_batch_multi", "hfcl_distance_batch_multi",
    "hfcl_collide_batch_multi_device", "hfcl_distance_batch_multi_device", "hfcl_collide_batch_multi_f32", "hfcl_distance_batch_multi_f32",
    "hfcl_lib_set_option", "hfcl_lib_option_key", "hfcl_has_ab_forms", "hfcl_multi_set_option", "hfcl_multi_last_gather",
]


class EngineError(RuntimeError):
    def __init__(self, code, msg):
        super().__init__("hfcl error %d: %s" % (code, msg))
        self.code = code


def build_native(verbose=False):
    """Compile the HIP extension in-tree for gfx950 (hipcc cross-compiles without a GPU)."""
    cmd = ["make", "-j8", "-C", _CSRC]
    if not verbose:
        cmd.insert(1, "-s")
    subprocess.check_call(cmd)
    return LIB_PATH


_DLL = None


def dll():
    """Load the native library; raises if it has not been built (no fallback)."""
    global _DLL
    if _DLL is None:
        if not os.path.exists(LIB_PATH):
            raise EngineError(abi.ERR_NO_DEVICE, "native library %s is missing: run __graft_entry__.build()" % LIB_PATH)
        # One HIP runtime per process: PyTorch-ROCm bundles its own libamdhip64.  If torch is going to
        # be used in this process (device tensors, streams, torch.distributed) it must be loaded first
        # so that this library binds to the same runtime; loading ours first makes torch.cuda unusable.
        try:
            import torch  # noqa: F401
        except Exception:
            pass
        d = C.CDLL(LIB_PATH)
        d.hfcl_last_error.restype = C.c_char_p
        d.hfcl_lib_create.restype = C.c_void_p
        d.hfcl_lib_num_shapes.restype = C.c_size_t
        d.hfcl_lib_climb_min.restype = C.c_uint32
        d.hfcl_last_kernel_ms.restype = C.c_double
        d.hfcl_last_kernel_name.restype = C.c_char_p
        d.hfcl_broadphase_self_pairs.restype = C.c_void_p
        d.hfcl_broadphase_pairs_between.restype = C.c_void_p
        d.hfcl_pairlist_size.restype = C.c_size_t
        d.hfcl_pairlist_data.restype = C.c_void_p
        d.hfcl_lib_option_key.restype = C.c_char_p
        _DLL = d
    return _DLL


def last_error():
    return dll().hfcl_last_error().decode()


def device_count():
    return int(dll().hfcl_device_count())


def has_ab_forms():
    """Does the loaded build carry the forms kept only as identity references (hfcl_has_ab_forms)?"""
    return bool(dll().hfcl_has_ab_forms())


def option_keys():
    """The names hfcl_lib_set_option accepts (hfcl_lib_option_key)."""
    out, i = [], 0
    while True:
        k = dll().hfcl_lib_option_key(C.c_int(i))
        if k is None:
            return out
        out.append(k.decode())
        i += 1


def bvh_build(vertices, triangles, n_threads=0):
    """hfcl_bvh_build: (nodes[BVH_NODE_DTYPE], primitive_indices) of BVHModel<OBBRSS> (host, no GPU needed)."""
    v = np.ascontiguousarray(vertices, dtype=np.float64).reshape(-1, 3)
    t = np.ascontiguousarray(triangles, dtype=np.uint32).reshape(-1, 3)
    nodes = np.zeros(max(2 * len(t) - 1, 0), dtype=abi.BVH_NODE_DTYPE)
    prim = np.zeros(len(t), dtype=np.uint32)
    _check(dll().hfcl_bvh_build(C.c_void_p(v.ctypes.data), C.c_size_t(len(v)), C.c_void_p(t.ctypes.data),
                                C.c_size_t(len(t)), C.c_void_p(nodes.ctypes.data), C.c_void_p(prim.ctypes.data),
                                C.c_int(n_threads)))
    return nodes, prim


def world_aabbs(shape_library, object_shape, object_tf, n_threads=0):
    """hfcl_world_aabbs: (n, 6) world AABBs (min, max) of posed objects (host)."""
    shapes = np.ascontiguousarray(shape_library.shapes_array())
    verts = np.ascontiguousarray(shape_library.vertices_array(), dtype=np.float64)
    ids = np.ascontiguousarray(object_shape, dtype=np.uint32)
    tf = np.ascontiguousarray(object_tf, dtype=np.float64).reshape(-1, 12)
    out = np.zeros((len(ids), 6), dtype=np.float64)
    _check(dll().hfcl_world_aabbs(abi.ptr(shapes), C.c_size_t(len(shapes)), abi.ptr(verts), abi.ptr(ids), abi.ptr(tf),
                                  C.c_size_t(len(ids)), abi.ptr(out), C.c_int(n_threads)))
    return out


def _take_pairlist(h):
    d = dll()
    h = C.c_void_p(h)
    n = d.hfcl_pairlist_size(h)
    if n:
        buf = (C.c_uint32 * (2 * n)).from_address(d.hfcl_pairlist_data(h))
        out = np.frombuffer(buf, dtype=np.uint32).reshape(-1, 2).copy()
    else:
        out = np.zeros((0, 2), dtype=np.uint32)
    d.hfcl_pairlist_free(h)
    return out


def broadphase_self_pairs(aabbs, n_threads=0):
    """All (i < j) with overlapping AABBs: what DynamicAABBTreeCollisionManager::collide reports."""
    a = np.ascontiguousarray(aabbs, dtype=np.float64).reshape(-1, 6)
    return _take_pairlist(dll().hfcl_broadphase_self_pairs(abi.ptr(a), C.c_size_t(len(a)), C.c_int(n_threads)))


def broadphase_pairs_between(aabbs_a, aabbs_b, n_threads=0):
    a = np.ascontiguousarray(aabbs_a, dtype=np.float64).reshape(-1, 6)
    b = np.ascontiguousarray(aabbs_b, dtype=np.float64).reshape(-1, 6)
    return _take_pairlist(dll().hfcl_broadphase_pairs_between(abi.ptr(a), C.c_size_t(len(a)), abi.ptr(b),
                                                              C.c_size_t(len(b)), C.c_int(n_threads)))


def _check(rc):
    if rc != 0:
        raise EngineError(rc, last_error())


def _dptr(x):
    """Device pointer of a torch tensor / int / None."""
    if x is None:
        return C.c_void_p(0)
    if isinstance(x, int):
        return C.c_void_p(x)
    return C.c_void_p(x.data_ptr())


class Library:
    """hfcl_lib: a shape library resident on one GPU."""

    def __init__(self, shape_library, device=0, options=None):
        """options: {key: value} for hfcl_lib_set_option, applied before the first batch."""
        d = dll()
        self._shapes = np.ascontiguousarray(shape_library.shapes_array())
        self._verts = np.ascontiguousarray(shape_library.vertices_array(), dtype=np.float64)
        self.device = device
        h = d.hfcl_lib_create(abi.ptr(self._shapes), C.c_size_t(len(self._shapes)), abi.ptr(self._verts),
                              C.c_size_t(len(self._verts)), C.c_int(device))
        if not h:
            raise EngineError(abi.ERR_NO_DEVICE, last_error())
        self._h = C.c_void_p(h)
        for k, v in (options or {}).items():
            self.set_option(k, v)

    def set_option(self, key, value):
        """hfcl_lib_set_option: a tuning option by name (the same names, upper-cased behind HFCL_, are the environment fallback)."""
        if isinstance(value, bool):
            value = int(value)
        if isinstance(value, (list, tuple)):
            value = ",".join(str(int(x)) for x in value)
        _check(dll().hfcl_lib_set_option(self._h, str(key).encode(), str(value).encode()))

    def close(self):
        if getattr(self, "_h", None):
            dll().hfcl_lib_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def add_bvh(self, mesh):
        """Register a BVHModel<OBBRSS> (bvh_builder.Mesh); returns its bvh_index."""
        nodes = np.ascontiguousarray(mesh.nodes)
        verts = np.ascontiguousarray(mesh.vertices, dtype=np.float64)
        tris = np.ascontiguousarray(mesh.triangles, dtype=np.uint32)
        idx = dll().hfcl_lib_add_bvh(self._h, abi.ptr(nodes), C.c_size_t(len(nodes)), abi.ptr(verts),
                                     C.c_size_t(len(verts)), abi.ptr(tris), C.c_size_t(len(tris)))
        if idx < 0:
            raise EngineError(abi.ERR_INVALID_ARGUMENT, last_error())
        return idx

    def collide_contacts(self, s1, s2, tf1, tf2, req, max_contacts):
        """Batched collide() returning (records, contacts[CONTACT_DTYPE], n_produced)."""
        s1 = np.ascontiguousarray(s1, dtype=np.uint32)
        s2 = np.ascontiguousarray(s2, dtype=np.uint32)
        tf1 = np.ascontiguousarray(tf1, dtype=np.float64).reshape(-1, 12)
        tf2 = np.ascontiguousarray(tf2, dtype=np.float64).reshape(-1, 12)
        n = len(s1)
        out = np.zeros(n, dtype=abi.RESULT_DTYPE)
        contacts = np.zeros(max_contacts, dtype=abi.CONTACT_DTYPE)
        nc = C.c_size_t(0)
        _check(dll().hfcl_collide_batch_contacts(self._h, abi.ptr(s1), abi.ptr(s2), abi.ptr(tf1), abi.ptr(tf2),
                                                 C.c_size_t(n), C.byref(req), abi.ptr(out), abi.ptr(contacts),
                                                 C.c_size_t(max_contacts), C.byref(nc)))
        return out, contacts[:min(nc.value, max_contacts)], int(nc.value)

    # ---- host-buffer entry points (H2D + kernels + D2H inside the call) ----
    def _host(self, fn, s1, s2, tf1, tf2, req, guess_in, want_guess, pose_width=12):
        s1 = np.ascontiguousarray(s1, dtype=np.uint32)
        s2 = np.ascontiguousarray(s2, dtype=np.uint32)
        tf1 = np.ascontiguousarray(tf1, dtype=np.float64).reshape(-1, pose_width)
        tf2 = np.ascontiguousarray(tf2, dtype=np.float64).reshape(-1, pose_width)
        n = len(s1)
        if not (len(s2) == n and len(tf1) == n and len(tf2) == n):
            raise ValueError("batch arrays must have equal length")
        out = np.zeros(n, dtype=abi.RESULT_DTYPE)
        if guess_in is not None:
            guess_in = np.ascontiguousarray(guess_in, dtype=abi.GUESS_DTYPE)
        gout = np.zeros(n, dtype=abi.GUESS_DTYPE) if want_guess else None
        rc = fn(self._h, abi.ptr(s1), abi.ptr(s2), abi.ptr(tf1), abi.ptr(tf2), C.c_size_t(n), C.byref(req),
                abi.ptr(out), abi.ptr(guess_in), abi.ptr(gout))
        _check(rc)
        return (out, gout) if want_guess else out

    def collide(self, s1, s2, tf1, tf2, req=None, guess_in=None, want_guess=False):
        """Batched hpp::fcl::collide (src/collision.cpp:69-130)."""
        return self._host(dll().hfcl_collide_batch, s1, s2, tf1, tf2, req or abi.default_collision_request(),
                          guess_in, want_guess)

    def distance(self, s1, s2, tf1, tf2, req=None, guess_in=None, want_guess=False):
        """Batched hpp::fcl::distance (src/distance.cpp:60-109)."""
        return self._host(dll().hfcl_distance_batch, s1, s2, tf1, tf2, req or abi.default_distance_request(),
                          guess_in, want_guess)

    def _host_f32(self, fn, s1, s2, pose1, pose2, req):
        s1 = np.ascontiguousarray(s1, dtype=np.uint32)
        s2 = np.ascontiguousarray(s2, dtype=np.uint32)
        p1 = np.ascontiguousarray(pose1, dtype=np.float32).reshape(-1, 7)
        p2 = np.ascontiguousarray(pose2, dtype=np.float32).reshape(-1, 7)
        n = len(s1)
        if not (len(s2) == len(p1) == len(p2) == n):
            raise ValueError("array lengths differ")
        out = np.zeros(n, dtype=abi.RESULT_F32_DTYPE)
        _check(fn(self._h, C.c_void_p(s1.ctypes.data), C.c_void_p(s2.ctypes.data), C.c_void_p(p1.ctypes.data), C.c_void_p(p2.ctypes.data),
                  C.c_size_t(n), C.byref(req), C.c_void_p(out.ctypes.data)))
        return out

    def collide_f32(self, s1, s2, pose1, pose2, req=None):
        """collide() through the fp32 path from host arrays: (n, 7) float32 poses (quaternion w, x, y, z + translation), hfcl_result_f32 records."""
        return self._host_f32(dll().hfcl_collide_batch_f32, s1, s2, pose1, pose2, req or abi.default_collision_request())

    def distance_f32(self, s1, s2, pose1, pose2, req=None):
        """distance() through the fp32 path from host arrays (see collide_f32)."""
        return self._host_f32(dll().hfcl_distance_batch_f32, s1, s2, pose1, pose2, req or abi.default_distance_request())

    def collide_qt(self, s1, s2, pose1, pose2, req=None, guess_in=None, want_guess=False):
        """collide() with compact host poses: (n, 7) float64 = quaternion (w, x, y, z) + translation."""
        return self._host(dll().hfcl_collide_batch_qt, s1, s2, pose1, pose2, req or abi.default_collision_request(),
                          guess_in, want_guess, pose_width=7)

    def distance_qt(self, s1, s2, pose1, pose2, req=None, guess_in=None, want_guess=False):
        """distance() with compact host poses: (n, 7) float64 = quaternion (w, x, y, z) + translation."""
        return self._host(dll().hfcl_distance_batch_qt, s1, s2, pose1, pose2, req or abi.default_distance_request(),
                          guess_in, want_guess, pose_width=7)

    def set_convex_neighbors(self, shape_id, offsets, neighbors):
        """ConvexBase::neighbors of one convex shape (CSR: offsets[num_points + 1], vertex indices relative to the
        shape): large hulls that have them hill-climb instead of scanning (hfcl_lib_set_convex_neighbors)."""
        off = np.ascontiguousarray(offsets, dtype=np.uint32)
        ids = np.ascontiguousarray(neighbors, dtype=np.uint32)
        _check(dll().hfcl_lib_set_convex_neighbors(self._h, C.c_uint32(int(shape_id)), C.c_void_p(off.ctypes.data),
                                                   C.c_void_p(ids.ctypes.data)))

    def climb_min(self):
        """Smallest hull (vertices) this library answers by hill-climbing a registered adjacency (hfcl_lib_climb_min)."""
        return int(dll().hfcl_lib_climb_min(self._h))

    def set_host_chunk(self, pairs):
        """Pairs per chunk of the host-buffer pipeline (0 = automatic)."""
        dll().hfcl_lib_set_host_chunk(self._h, C.c_size_t(int(pairs)))

    # ---- device-resident entry points (torch tensors or raw device pointers) ----
    def collide_device(self, d_s1, d_s2, d_tf1, d_tf2, n, req, d_out, d_gin=None, d_gout=None, stream=0):
        _check(dll().hfcl_collide_batch_device(self._h, _dptr(d_s1), _dptr(d_s2), _dptr(d_tf1), _dptr(d_tf2),
                                                C.c_size_t(n), C.byref(req), _dptr(d_out), _dptr(d_gin),
                                                _dptr(d_gout), C.c_void_p(stream)))

    def distance_device(self, d_s1, d_s2, d_tf1, d_tf2, n, req, d_out, d_gin=None, d_gout=None, stream=0):
        _check(dll().hfcl_distance_batch_device(self._h, _dptr(d_s1), _dptr(d_s2), _dptr(d_tf1), _dptr(d_tf2),
                                                 C.c_size_t(n), C.byref(req), _dptr(d_out), _dptr(d_gin),
                                                 _dptr(d_gout), C.c_void_p(stream)))

    def distance_device_f32(self, d_s1, d_s2, d_pose1, d_pose2, n, req, d_out, stream=0):
        _check(dll().hfcl_distance_batch_device_f32(self._h, _dptr(d_s1), _dptr(d_s2), _dptr(d_pose1),
                                                     _dptr(d_pose2), C.c_size_t(n), C.byref(req), _dptr(d_out),
                                                     C.c_void_p(stream)))

    def collide_device_f32(self, d_s1, d_s2, d_pose1, d_pose2, n, req, d_out, stream=0):
        _check(dll().hfcl_collide_batch_device_f32(self._h, _dptr(d_s1), _dptr(d_s2), _dptr(d_pose1),
                                                    _dptr(d_pose2), C.c_size_t(n), C.byref(req), _dptr(d_out),
                                                    C.c_void_p(stream)))

    def compact_results_device(self, d_records, n, d_out, f32=False, stream=0):
        """Full device records -> hfcl_result_compact{,_f32} records (24 / 8 B): the multi-GPU exchange format."""
        fn = dll().hfcl_compact_results_device_f32 if f32 else dll().hfcl_compact_results_device
        _check(fn(self._h, _dptr(d_records), C.c_size_t(n), _dptr(d_out), C.c_void_p(stream)))

    # ---- instrumentation ----
    def set_split(self, parts):
        """2: large batches run as two halves on two streams (hfcl_lib_set_split); 1: one stream."""
        dll().hfcl_lib_set_split(self._h, C.c_int(int(parts)))

    def get_split(self):
        return int(dll().hfcl_lib_get_split(self._h))

    def last_split_parts(self):
        """1 or 2: how the last batch actually ran."""
        return int(dll().hfcl_lib_last_split_parts(self._h))

    def set_kernel_timing(self, on):
        """Per-kernel HIP events on/off (on by default; off saves two stream markers per launch)."""
        dll().hfcl_lib_set_kernel_timing(self._h, C.c_int(1 if on else 0))

    def last_kernel_ms(self):
        return float(dll().hfcl_last_kernel_ms(self._h))

    def last_kernel_name(self):
        return dll().hfcl_last_kernel_name(self._h).decode()

    def last_kernel_breakdown(self):
        names = (C.c_char_p * 16)()
        ms = (C.c_double * 16)()
        k = dll().hfcl_last_kernel_breakdown(self._h, names, ms, 16)
        return [(names[i].decode(), float(ms[i])) for i in range(k)]

    def last_bucket_counts(self):
        out = (C.c_uint32 * 12)()
        dll().hfcl_last_bucket_counts(self._h, out)
        keys = ["closed", "prim", "cc", "pc", "cp", "bvh", "unsupported", "large", "bvh_shape", "tri", "epa_queue", "epa_overflow"]
        return dict(zip(keys, [int(v) for v in out]))


    def last_ordered_reruns(self):
        """distance() on meshes, last call: walks a wave continued / of those re-run in the reference's order, mesh x mesh then mesh x solid."""
        out = (C.c_uint32 * 4)()
        dll().hfcl_last_ordered_reruns(self._h, out)
        return dict(zip(["mesh_continued", "mesh_rerun", "solid_continued", "solid_rerun"], [int(v) for v in out]))


def shard_range(n, rank, world):
    """hfcl_shard_range of the C ABI (= sharding.shard_range)."""
    lo, hi = C.c_size_t(0), C.c_size_t(0)
    dll().hfcl_shard_range(C.c_size_t(int(n)), C.c_int(int(rank)), C.c_int(int(world)), C.byref(lo), C.byref(hi))
    return lo.value, hi.value


class MultiLibrary:
    """hfcl_multi: replicas of a shape library on several devices of this process (include/hppfcl_amd.h); a batch is cut
    into contiguous shards, one per replica.  `devices` may list a device more than once (host-buffer entry points)."""

    def __init__(self, shape_library, devices=(0,), options=None):
        d = dll()
        self._shapes = np.ascontiguousarray(shape_library.shapes_array())
        self._verts = np.ascontiguousarray(shape_library.vertices_array(), dtype=np.float64)
        self.devices = [int(x) for x in devices]
        dev = (C.c_int * len(self.devices))(*self.devices)
        d.hfcl_multi_create.restype = C.c_void_p
        h = d.hfcl_multi_create(dev, C.c_int(len(self.devices)), abi.ptr(self._shapes), C.c_size_t(len(self._shapes)), abi.ptr(self._verts),
                                C.c_size_t(len(self._verts)))
        if not h:
            raise EngineError(abi.ERR_NO_DEVICE, last_error())
        self._h = C.c_void_p(h)
        for k, v in (options or {}).items():
            self.set_option(k, v)

    def set_option(self, key, value):
        """hfcl_multi_set_option: hfcl_lib_set_option on every replica."""
        if isinstance(value, bool):
            value = int(value)
        if isinstance(value, (list, tuple)):
            value = ",".join(str(int(x)) for x in value)
        _check(dll().hfcl_multi_set_option(self._h, str(key).encode(), str(value).encode()))

    def last_gather(self):
        """hfcl_multi_last_gather: ranks the communicator reports, milliseconds of the all-gather (None: no collective), bytes per rank."""
        ranks, ms, nbytes = C.c_int(0), C.c_double(-1.0), C.c_size_t(0)
        _check(dll().hfcl_multi_last_gather(self._h, C.byref(ranks), C.byref(ms), C.byref(nbytes)))
        return {"ranks": int(ranks.value), "ms": float(ms.value) if ms.value >= 0 else None, "bytes_per_rank": int(nbytes.value)}

    def close(self):
        if getattr(self, "_h", None):
            dll().hfcl_multi_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def __len__(self):
        return int(dll().hfcl_multi_size(self._h))

    def add_bvh(self, mesh):
        nodes = np.ascontiguousarray(mesh.nodes)
        verts = np.ascontiguousarray(mesh.vertices, dtype=np.float64)
        tris = np.ascontiguousarray(mesh.triangles, dtype=np.uint32)
        idx = dll().hfcl_multi_add_bvh(self._h, abi.ptr(nodes), C.c_size_t(len(nodes)), abi.ptr(verts), C.c_size_t(len(verts)), abi.ptr(tris),
                                       C.c_size_t(len(tris)))
        if idx < 0:
            raise EngineError(abi.ERR_INVALID_ARGUMENT, last_error())
        return idx

    _host = Library._host

    def collide(self, s1, s2, tf1, tf2, req=None, guess_in=None, want_guess=False):
        return self._host(dll().hfcl_collide_batch_multi, s1, s2, tf1, tf2, req or abi.default_collision_request(), guess_in, want_guess)

    def distance(self, s1, s2, tf1, tf2, req=None, guess_in=None, want_guess=False):
        return self._host(dll().hfcl_distance_batch_multi, s1, s2, tf1, tf2, req or abi.default_distance_request(), guess_in, want_guess)

    def collide_f32(self, s1, s2, pose1, pose2, req=None):
        return Library._host_f32(self, dll().hfcl_collide_batch_multi_f32, s1, s2, pose1, pose2, req or abi.default_collision_request())

    def distance_f32(self, s1, s2, pose1, pose2, req=None):
        return Library._host_f32(self, dll().hfcl_distance_batch_multi_f32, s1, s2, pose1, pose2, req or abi.default_distance_request())

    def _gathered(self, fn, d_s1, d_s2, d_tf1, d_tf2, n, req, d_gathered, streams=None):
        """Per-replica lists of device buffers (torch tensors or raw pointers); d_gathered[g]: len(self) * ceil(n / len(self)) records."""
        G = len(self)
        arr = lambda xs: (C.c_void_p * G)(*[_dptr(x) for x in xs])  # noqa: E731
        st = (C.c_void_p * G)(*[C.c_void_p(int(x)) for x in streams]) if streams is not None else None
        _check(fn(self._h, arr(d_s1), arr(d_s2), arr(d_tf1), arr(d_tf2), C.c_size_t(int(n)), C.byref(req), arr(d_gathered), st))

    def collide_device_gathered(self, d_s1, d_s2, d_tf1, d_tf2, n, req, d_gathered, streams=None):
        self._gathered(dll().hfcl_collide_batch_multi_device, d_s1, d_s2, d_tf1, d_tf2, n, req, d_gathered, streams)

    def distance_device_gathered(self, d_s1, d_s2, d_tf1, d_tf2, n, req, d_gathered, streams=None):
        self._gathered(dll().hfcl_distance_batch_multi_device, d_s1, d_s2, d_tf1, d_tf2, n, req, d_gathered, streams)
