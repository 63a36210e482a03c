"""Host-side mirror of the reference's geometry constructors and pose type (plumbing only).

Mirrors: Box/Sphere/Capsule/Ellipsoid/ConvexBase (include/hpp/fcl/shape/geometric_shapes.h),
Transform3f (include/hpp/fcl/math/transform.h:56-218), makeQuat(w,x,y,z)."""
import numpy as np

from . import abi


class ShapeLibrary:
    """Flat table of shapes + one vertex array: the host image of hfcl_lib_create's inputs."""

    def __init__(self):
        self._shapes = []   # pending single entries (tuples)
        self._chunks = []   # finished SHAPE_DTYPE arrays, in shape-id order
        self._count = 0
        self._verts = []
        self._nverts = 0

    def _add(self, type_, params=(0, 0, 0, 0), ssr=0.0, num_points=0, vertex_offset=0, bvh_index=0):
        self._shapes.append((type_, num_points, vertex_offset, bvh_index, (tuple(params) + (0, 0, 0, 0))[:4], ssr))
        self._count += 1
        return self._count - 1

    def _flush(self):
        if self._shapes:
            a = np.zeros(len(self._shapes), dtype=abi.SHAPE_DTYPE)
            for i, (t, n, off, bi, p, ssr) in enumerate(self._shapes):
                a[i]["type"] = t
                a[i]["num_points"] = n
                a[i]["vertex_offset"] = off
                a[i]["bvh_index"] = bi
                a[i]["params"] = p
                a[i]["swept_sphere_radius"] = ssr
            self._chunks.append(a)
            self._shapes = []

    def add_box(self, x, y, z, swept_sphere_radius=0.0):
        """Box(x,y,z): halfSide = (x/2,y/2,z/2)   geometric_shapes.h:166"""
        return self._add(abi.GEOM_BOX, (x / 2.0, y / 2.0, z / 2.0), swept_sphere_radius)

    def add_sphere(self, radius, swept_sphere_radius=0.0):
        return self._add(abi.GEOM_SPHERE, (radius, 0, 0), swept_sphere_radius)

    def add_capsule(self, radius, lz, swept_sphere_radius=0.0):
        """Capsule(radius, lz): halfLength = lz/2   geometric_shapes.h:386-387"""
        return self._add(abi.GEOM_CAPSULE, (radius, lz / 2.0, 0), swept_sphere_radius)

    def add_cone(self, radius, lz, swept_sphere_radius=0.0):
        """Cone(radius, lz): params = (radius, halfLength)  (geometric_shapes.h:437-500)."""
        return self._add(abi.GEOM_CONE, (radius, lz / 2.0, 0), swept_sphere_radius)

    def add_cylinder(self, radius, lz, swept_sphere_radius=0.0):
        return self._add(abi.GEOM_CYLINDER, (radius, lz / 2.0, 0), swept_sphere_radius)

    def _unit_plane(self, n, d):
        """unitNormalTest (geometric_shapes.cpp:121-143): normalise (n, d); zero normal -> (1,0,0), 0."""
        n = np.asarray(n, dtype=np.float64)
        l = float(np.sqrt((n * n).sum()))
        if l > 0:
            inv = 1.0 / l
            return tuple(float(x) for x in n * inv) + (float(d) * inv,)
        return (1.0, 0.0, 0.0, 0.0)

    def add_halfspace(self, n, d, swept_sphere_radius=0.0):
        """Halfspace {x : n.x <= d} (geometric_shapes.h:873-962)."""
        return self._add(abi.GEOM_HALFSPACE, self._unit_plane(n, d), swept_sphere_radius)

    def add_plane(self, n, d, swept_sphere_radius=0.0):
        """Plane {x : n.x = d} (geometric_shapes.h:968-1049)."""
        return self._add(abi.GEOM_PLANE, self._unit_plane(n, d), swept_sphere_radius)

    def add_ellipsoid(self, rx, ry, rz, swept_sphere_radius=0.0):
        return self._add(abi.GEOM_ELLIPSOID, (rx, ry, rz), swept_sphere_radius)

    def add_convex(self, points, swept_sphere_radius=0.0):
        pts = np.ascontiguousarray(points, dtype=np.float64).reshape(-1, 3)
        off = self._nverts
        self._verts.append(pts)
        self._nverts += len(pts)
        return self._add(abi.GEOM_CONVEX, (0, 0, 0), swept_sphere_radius, len(pts), off)

    def add_convex_many(self, points, swept_sphere_radius=0.0):
        """m hulls of k points each from one (m, k, 3) array (bulk form of add_convex: one hull per
        pair workloads carry millions of them).  Returns the id of the first."""
        pts = np.ascontiguousarray(points, dtype=np.float64)
        m, k = pts.shape[0], pts.shape[1]
        self._flush()
        a = np.zeros(m, dtype=abi.SHAPE_DTYPE)
        a["type"] = abi.GEOM_CONVEX
        a["num_points"] = k
        a["vertex_offset"] = self._nverts + k * np.arange(m, dtype=np.uint32)
        a["swept_sphere_radius"] = swept_sphere_radius
        self._chunks.append(a)
        self._verts.append(pts.reshape(-1, 3))
        self._nverts += m * k
        first = self._count
        self._count += m
        return first

    def add_triangle(self, a, b, c, swept_sphere_radius=0.0):
        pts = np.array([a, b, c], dtype=np.float64)
        off = self._nverts
        self._verts.append(pts)
        self._nverts += 3
        return self._add(abi.GEOM_TRIANGLE, (0, 0, 0), swept_sphere_radius, 3, off)

    def add_bvh(self, bvh_index, num_vertices=0):
        return self._add(abi.BV_OBBRSS, (0, 0, 0), 0.0, num_vertices, 0, bvh_index)

    def __len__(self):
        return self._count

    def shapes_array(self):
        self._flush()
        if not self._chunks:
            return np.zeros(0, dtype=abi.SHAPE_DTYPE)
        if len(self._chunks) > 1:
            self._chunks = [np.concatenate(self._chunks)]
        return self._chunks[0]

    def vertices_array(self):
        if not self._verts:
            return np.zeros((0, 3), dtype=np.float64)
        return np.ascontiguousarray(np.concatenate(self._verts, axis=0))


def quat_to_matrix(q):
    """Unit quaternion(s) (w,x,y,z) -> rotation matrix (Eigen Quaternion::toRotationMatrix)."""
    q = np.asarray(q, dtype=np.float64)
    w, x, y, z = q[..., 0], q[..., 1], q[..., 2], q[..., 3]
    tx, ty, tz = 2 * x, 2 * y, 2 * z
    twx, twy, twz = tx * w, ty * w, tz * w
    txx, txy, txz = tx * x, ty * x, tz * x
    tyy, tyz, tzz = ty * y, tz * y, tz * z
    R = np.empty(q.shape[:-1] + (3, 3), dtype=np.float64)
    R[..., 0, 0] = 1 - (tyy + tzz)
    R[..., 0, 1] = txy - twz
    R[..., 0, 2] = txz + twy
    R[..., 1, 0] = txy + twz
    R[..., 1, 1] = 1 - (txx + tzz)
    R[..., 1, 2] = tyz - twx
    R[..., 2, 0] = txz - twy
    R[..., 2, 1] = tyz + twx
    R[..., 2, 2] = 1 - (txx + tyy)
    return R


def make_pose(R=None, T=None, quat=None):
    """One or many poses in the ABI layout: 9 doubles column-major R then T (Transform3f image)."""
    if quat is not None:
        R = quat_to_matrix(quat)
    if R is None:
        R = np.eye(3)
    R = np.asarray(R, dtype=np.float64)
    if T is None:
        T = np.zeros(R.shape[:-2] + (3,))
    T = np.asarray(T, dtype=np.float64)
    batch = np.broadcast_shapes(R.shape[:-2], T.shape[:-1])  # one rotation for many translations, or the reverse
    R = np.broadcast_to(R, batch + (3, 3))
    out = np.empty(batch + (12,), dtype=np.float64)
    out[..., 0:9] = np.swapaxes(R, -1, -2).reshape(R.shape[:-2] + (9,))  # column-major
    out[..., 9:12] = T
    return out


def pose_R(p):
    p = np.asarray(p)
    return np.swapaxes(p[..., 0:9].reshape(p.shape[:-1] + (3, 3)), -1, -2)


def pose_T(p):
    return np.asarray(p)[..., 9:12]


def compose(a, b):
    """Transform3f a * b."""
    Ra, Rb = pose_R(a), pose_R(b)
    return make_pose(Ra @ Rb, (Ra @ pose_T(b)[..., None])[..., 0] + pose_T(a))


def transform_point(p, v):
    return (pose_R(p) @ np.asarray(v, dtype=np.float64)[..., None])[..., 0] + pose_T(p)


def pose_f32_from_quat(quat, T):
    """7-float compact pose (quat wxyz + translation) for the fp32 device path."""
    q = np.asarray(quat, dtype=np.float64)
    out = np.empty(q.shape[:-1] + (7,), dtype=np.float32)
    out[..., 0:4] = q
    out[..., 4:7] = T
    return out
