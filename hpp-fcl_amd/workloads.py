"""Synthetic workloads of BASELINE.json's configs.  Plumbing: numpy only.

All randomness comes from a counter-based generator (Philox) keyed by (seed, stream) so the
same batch can be regenerated anywhere (host, GPU box, any rank) without shipping data.
Default seed = 1, echoing the reference's unseeded rand() (test/utility.cpp:93-96)."""
import numpy as np

from . import geometry


def _rng(seed, stream):
    return np.random.Generator(np.random.Philox(key=[int(seed), int(stream)]))


def uniform_quaternions(rng, n):
    """uniformRandomQuaternion, include/hpp/fcl/math/transform.h:228-249 -> (w,x,y,z)."""
    u1, u2, u3 = rng.random(n), rng.random(n), rng.random(n)
    m1, m2 = np.sqrt(1.0 - u1), np.sqrt(u1)
    q = np.empty((n, 4))
    q[:, 0] = m1 * np.sin(2 * np.pi * u2)
    q[:, 1] = m1 * np.cos(2 * np.pi * u2)
    q[:, 2] = m2 * np.sin(2 * np.pi * u3)
    q[:, 3] = m2 * np.cos(2 * np.pi * u3)
    return q


def fibonacci_sphere(n):
    i = np.arange(n) + 0.5
    phi = np.arccos(1 - 2 * i / n)
    theta = np.pi * (1 + 5 ** 0.5) * i
    return np.stack([np.cos(theta) * np.sin(phi), np.sin(theta) * np.sin(phi), np.cos(phi)], axis=1)


class Batch:
    """One batch of queries: shape library + pair list + poses (both precisions)."""

    def __init__(self, name, lib, s1, s2, quat1, T1, quat2, T2, kind, request_overrides=None):
        self.name = name
        self.lib = lib
        self.shapes = lib.shapes_array()
        self.verts = lib.vertices_array()
        self.s1 = np.ascontiguousarray(s1, dtype=np.uint32)
        self.s2 = np.ascontiguousarray(s2, dtype=np.uint32)
        self.quat1, self.T1, self.quat2, self.T2 = quat1, T1, quat2, T2
        self.kind = kind  # "distance" | "collide"
        self.request_overrides = request_overrides or {}
        self.meshes = None

    def __len__(self):
        return len(self.s1)

    @property
    def tf1(self):
        return geometry.make_pose(quat=self.quat1, T=self.T1)

    @property
    def tf2(self):
        return geometry.make_pose(quat=self.quat2, T=self.T2)

    @property
    def pose1_f32(self):
        return geometry.pose_f32_from_quat(self.quat1, self.T1)

    @property
    def pose2_f32(self):
        return geometry.pose_f32_from_quat(self.quat2, self.T2)

    @property
    def pose1_qt(self):
        """(n, 7) float64 compact host poses (quaternion w, x, y, z + translation) for the *_qt entry points."""
        return np.concatenate([np.asarray(self.quat1, dtype=np.float64), np.asarray(self.T1, dtype=np.float64)], axis=-1)

    @property
    def pose2_qt(self):
        return np.concatenate([np.asarray(self.quat2, dtype=np.float64), np.asarray(self.T2, dtype=np.float64)], axis=-1)

    def tf_from_f32(self):
        """fp64 poses built from the *rounded* fp32 quaternion/translation (what the fp32 path sees)."""
        p1, p2 = self.pose1_f32.astype(np.float64), self.pose2_f32.astype(np.float64)

        def mk(p):
            q = p[:, :4] / np.linalg.norm(p[:, :4], axis=1, keepdims=True)
            return geometry.make_pose(quat=q, T=p[:, 4:7])

        return mk(p1), mk(p2)

    def slice(self, lo, hi):
        b = Batch.__new__(Batch)
        b.__dict__.update(self.__dict__)
        b.s1, b.s2 = self.s1[lo:hi], self.s2[lo:hi]
        b.quat1, b.T1, b.quat2, b.T2 = self.quat1[lo:hi], self.T1[lo:hi], self.quat2[lo:hi], self.T2[lo:hi]
        return b


def _poses(rng, n, half_width):
    q1, q2 = uniform_quaternions(rng, n), uniform_quaternions(rng, n)
    T1 = rng.uniform(-half_width, half_width, (n, 3))
    T2 = rng.uniform(-half_width, half_width, (n, 3))
    return q1, T1, q2, T2


def cfg1_sphere_sphere(n=1000, seed=1):
    """cfg1: Sphere-Sphere distance(), radii U[0.1,1] (test/utility.cpp:565-567)."""
    rng = _rng(seed, 1)
    lib = geometry.ShapeLibrary()
    nlib = 256
    for r in rng.uniform(0.1, 1.0, nlib):
        lib.add_sphere(float(r))
    s1, s2 = rng.integers(0, nlib, n), rng.integers(0, nlib, n)
    q1, T1, q2, T2 = _poses(rng, n, 0.9)
    return Batch("cfg1_sphere_sphere_distance", lib, s1, s2, q1, T1, q2, T2, "distance")


def cfg2_box_capsule(n=1_000_000, seed=1, nlib=1024, half_width=0.92):
    """cfg2: Box-Capsule collide(), default request.  Box side U[0.1,1]^3 (makeRandomBox,
    test/utility.cpp:559-563), Capsule r U[0.1,0.8], lz U[0.2,1.0] (:575-579).  The translation cube is
    sized so ~30 % of the pairs penetrate (the reference's GJK benchmark had 28 % colliding,
    test/benchmark/test_fcl_gjk.output:3)."""
    rng = _rng(seed, 2)
    lib = geometry.ShapeLibrary()
    for s in rng.uniform(0.1, 1.0, (nlib, 3)):
        lib.add_box(*map(float, s))
    for r, lz in zip(rng.uniform(0.1, 0.8, nlib), rng.uniform(0.2, 1.0, nlib)):
        lib.add_capsule(float(r), float(lz))
    s1 = rng.integers(0, nlib, n)
    s2 = nlib + rng.integers(0, nlib, n)
    q1, T1, q2, T2 = _poses(rng, n, half_width)
    return Batch("cfg2_box_capsule_collide", lib, s1, s2, q1, T1, q2, T2, "collide")


def convex_hull_library(rng, nlib, nverts=32):
    lib = geometry.ShapeLibrary()
    base = fibonacci_sphere(nverts)
    for radii in rng.uniform(0.1, 1.0, (nlib, 3)):
        lib.add_convex(base * radii)
    return lib


def cfg3_convex_convex(n=1_000_000, seed=1, nlib=4096, half_width=1.0, nverts=32):
    """cfg3: Convex-Convex distance(), signed, NesterovAcceleration.  Hulls: 32 Fibonacci-sphere
    directions scaled by ellipsoid radii U[0.1,1]^3 (every point is a hull vertex, linear
    support path; no qhull needed).  Shared library of `nlib` hulls."""
    rng = _rng(seed, 3)
    lib = convex_hull_library(rng, nlib, nverts)
    s1, s2 = rng.integers(0, nlib, n), rng.integers(0, nlib, n)
    q1, T1, q2, T2 = _poses(rng, n, half_width)
    from . import abi
    return Batch("cfg3_convex32_distance_nesterov", lib, s1, s2, q1, T1, q2, T2, "distance",
                 {"gjk_variant": abi.NesterovAcceleration})


def cfg3_unique_hulls(n=1_000_000, seed=1, half_width=1.0, nverts=32):
    """cfg3, variant (ii) of SURVEY.md 8d: every pair brings its own two hulls (pair i = hulls 2i, 2i+1 of a
    2n-hull library), so the 2 x 32 x 12 B of vertices are compulsory HBM traffic per query (876 B/query)."""
    rng = _rng(seed, 33)
    base = fibonacci_sphere(nverts)
    radii = rng.uniform(0.1, 1.0, (2 * n, 1, 3))
    lib = geometry.ShapeLibrary()
    lib.add_convex_many(base[None, :, :] * radii)
    s1 = 2 * np.arange(n, dtype=np.int64)
    s2 = s1 + 1
    q1, T1, q2, T2 = _poses(rng, n, half_width)
    from . import abi
    return Batch("cfg3_convex32_unique_hulls_distance_nesterov", lib, s1, s2, q1, T1, q2, T2, "distance",
                 {"gjk_variant": abi.NesterovAcceleration})


def _mixed_library(rng, nper):
    lib = geometry.ShapeLibrary()
    for s in rng.uniform(0.1, 1.0, (nper, 3)):
        lib.add_box(*map(float, s))
    for r in rng.uniform(0.1, 1.0, nper):
        lib.add_sphere(float(r))
    for r, lz in zip(rng.uniform(0.1, 0.8, nper), rng.uniform(0.2, 1.0, nper)):
        lib.add_capsule(float(r), float(lz))
    for r in rng.uniform(0.1, 1.0, (nper, 3)):
        lib.add_ellipsoid(*map(float, r))
    base = fibonacci_sphere(32)
    for radii in rng.uniform(0.1, 1.0, (nper, 3)):
        lib.add_convex(base * radii)
    return lib


def all_primitives(n=100_000, seed=1, nper=128, half_width=0.8, kind="collide"):
    """Every supported shape kind against every other (Box, Sphere, Capsule, Cone, Cylinder, Ellipsoid,
    Convex32): exercises the whole dispatch table of shape_shape_func.h:185-211 that is in scope."""
    rng = _rng(seed, 6)
    lib = _mixed_library(rng, nper)
    for r, lz in zip(rng.uniform(0.1, 0.8, nper), rng.uniform(0.2, 1.0, nper)):
        lib.add_cone(float(r), float(lz))
    for r, lz in zip(rng.uniform(0.1, 0.8, nper), rng.uniform(0.2, 1.0, nper)):
        lib.add_cylinder(float(r), float(lz))
    s1, s2 = rng.integers(0, 7 * nper, n), rng.integers(0, 7 * nper, n)
    q1, T1, q2, T2 = _poses(rng, n, half_width)
    return Batch("all_primitives_" + kind, lib, s1, s2, q1, T1, q2, T2, kind)


def triangle_pairs(n=50_000, seed=1, nper=48, half_width=0.45, kind="collide"):
    """Top-level TriangleP rows of the dispatch table (collision_func_matrix.cpp:295-469): a TriangleP against
    every solid kind, both operand orders, and TriangleP x TriangleP.  collide() only: the reference's distance
    matrix has no TriangleP entries (src/distance_func_matrix.cpp)."""
    rng = _rng(seed, 9)
    lib = _mixed_library(rng, nper)
    for r, lz in zip(rng.uniform(0.1, 0.8, nper), rng.uniform(0.2, 1.0, nper)):
        lib.add_cone(float(r), float(lz))
    for r, lz in zip(rng.uniform(0.1, 0.8, nper), rng.uniform(0.2, 1.0, nper)):
        lib.add_cylinder(float(r), float(lz))
    base48 = fibonacci_sphere(48)
    for radii in rng.uniform(0.1, 1.0, (nper, 3)):
        lib.add_convex(base48 * radii)  # above the 32-vertex threshold: scanned from memory
    n_solid = 8 * nper
    ntri = 4 * nper
    for k in range(ntri):
        c = rng.uniform(-0.3, 0.3, 3)
        lib.add_triangle(*(c + rng.uniform(-0.6, 0.6, (3, 3))))
    other = rng.integers(0, n_solid + ntri, n)  # a solid or another triangle
    tri = n_solid + rng.integers(0, ntri, n)
    first = rng.integers(0, 2, n).astype(bool)
    s1 = np.where(first, tri, other)
    s2 = np.where(first, other, tri)
    q1, T1, q2, T2 = _poses(rng, n, half_width)
    return Batch("triangle_pairs_" + kind, lib, s1, s2, q1, T1, q2, T2, kind)


def flat_pairs(n=50_000, seed=1, nper=64, half_width=1.0, kind="collide"):
    """Plane / Halfspace rows of the dispatch table: (solid, flat), (flat, solid) and (flat, flat) pairs, the
    solids being every other supported kind (some with a swept-sphere radius)."""
    rng = _rng(seed, 7)
    lib = _mixed_library(rng, nper)
    for r, lz in zip(rng.uniform(0.1, 0.8, nper), rng.uniform(0.2, 1.0, nper)):
        lib.add_cone(float(r), float(lz))
    for r, lz in zip(rng.uniform(0.1, 0.8, nper), rng.uniform(0.2, 1.0, nper)):
        lib.add_cylinder(float(r), float(lz), swept_sphere_radius=float(rng.uniform(0, 0.1)))
    n_solid = 7 * nper
    nflat = 2 * nper
    for k in range(nflat):
        nrm, d, ssr = rng.normal(size=3), float(rng.uniform(-0.5, 0.5)), float(rng.uniform(0, 0.05)) * (k % 3 == 0)
        if k % 8 == 0:
            nrm = np.array([0.0, 0.0, 1.0])  # parallel flats when both poses share a rotation
        (lib.add_halfspace if k % 2 == 0 else lib.add_plane)(nrm, d, swept_sphere_radius=ssr)
    u = rng.random(n)
    solid, flat1, flat2 = rng.integers(0, n_solid, n), n_solid + rng.integers(0, nflat, n), n_solid + rng.integers(0, nflat, n)
    s1 = np.where(u < 0.45, solid, flat1)
    s2 = np.where(u < 0.45, flat2, np.where(u < 0.9, solid, flat2))
    q1, T1, q2, T2 = _poses(rng, n, half_width)
    par = rng.random(n) < 0.3  # same rotation on both sides: exercises the parallel branches of flat-flat
    q2[par] = q1[par]
    return Batch("flat_pairs_" + kind, lib, s1, s2, q1, T1, q2, T2, kind)


def hull_adjacency(points):
    """Vertex adjacency of a convex point set as Convex<Triangle>::fillNeighbors builds it (shape/details/convex.hxx:
    231-280: per vertex the ascending set of vertices it shares a facet edge with), from the facets of scipy's Qhull
    wrapper.  Returns CSR (offsets[n + 1], ids) for hfcl_lib_set_convex_neighbors / Library.set_convex_neighbors."""
    from scipy.spatial import ConvexHull
    pts = np.asarray(points, dtype=np.float64).reshape(-1, 3)
    n = len(pts)
    hull = ConvexHull(pts)
    edges = np.concatenate([hull.simplices[:, [0, 1]], hull.simplices[:, [1, 2]], hull.simplices[:, [2, 0]]])
    edges = np.concatenate([edges, edges[:, ::-1]])
    edges = np.unique(edges, axis=0)  # sorted by (vertex, neighbour)
    offs = np.zeros(n + 1, dtype=np.uint32)
    np.cumsum(np.bincount(edges[:, 0], minlength=n), out=offs[1:])
    return offs, edges[:, 1].astype(np.uint32)


def subdivided_box(m=11, half=(1.0, 1.0, 1.0)):
    """A box whose faces are m x m grids of points, triangulated: (points, offsets, ids).  Most of its points are NOT
    extreme -- they lie inside a flat facet or an edge, with every neighbour in that facet's plane -- which is what a
    meshed CAD part handed to Convex<Triangle> looks like (m = 11: 602 points, above the hill-climb threshold)."""
    idx = {}
    pts = []
    for i in range(m):
        for j in range(m):
            for k in range(m):
                if i in (0, m - 1) or j in (0, m - 1) or k in (0, m - 1):
                    idx[(i, j, k)] = len(pts)
                    pts.append([half[0] * (2.0 * i / (m - 1) - 1), half[1] * (2.0 * j / (m - 1) - 1), half[2] * (2.0 * k / (m - 1) - 1)])
    nb = [set() for _ in pts]

    def edge(a, b):
        nb[a].add(b)
        nb[b].add(a)

    for axis in range(3):
        for side in (0, m - 1):
            for u in range(m - 1):
                for w in range(m - 1):
                    def at(du, dw):
                        c = [0, 0, 0]
                        c[axis] = side
                        c[(axis + 1) % 3] = u + du
                        c[(axis + 2) % 3] = w + dw
                        return idx[tuple(c)]
                    a, b, c, d = at(0, 0), at(1, 0), at(1, 1), at(0, 1)
                    for x, y in ((a, b), (b, c), (c, d), (d, a), (a, c)):  # two triangles per grid cell
                        edge(x, y)
    offs = np.zeros(len(pts) + 1, dtype=np.uint32)
    offs[1:] = np.cumsum([len(x) for x in nb])
    return np.array(pts), offs, np.array([v for x in nb for v in sorted(x)], dtype=np.uint32)


def register_adjacency(library, shapes, verts, min_points=33):
    """hull_adjacency for every convex shape of at least `min_points` vertices of a shape table, registered with the
    engine library.  Returns the number of shapes registered."""
    from . import abi
    count = 0
    verts = np.asarray(verts, dtype=np.float64).reshape(-1, 3)
    for i, s in enumerate(shapes):
        if s["type"] != abi.GEOM_CONVEX or s["num_points"] < min_points:
            continue
        off, n = int(s["vertex_offset"]), int(s["num_points"])
        offs, ids = hull_adjacency(verts[off:off + n])
        library.set_convex_neighbors(i, offs, ids)
        count += 1
    return count


def large_convex(n=50_000, seed=1, sizes=(33, 48, 64, 100, 256), nlib_each=12, half_width=1.0, kind="distance"):
    """Hulls above ConvexBase::num_vertices_large_convex_threshold (32) against each other, small hulls and
    primitives.  Points are random directions scaled onto an ellipsoid, so every point is a hull vertex."""
    rng = _rng(seed, 8)
    lib = geometry.ShapeLibrary()
    for nv in sizes:
        for radii in rng.uniform(0.2, 1.0, (nlib_each, 3)):
            d = rng.normal(size=(nv, 3))
            lib.add_convex(d / np.linalg.norm(d, axis=1, keepdims=True) * radii)
    n_large = len(sizes) * nlib_each
    base = fibonacci_sphere(32)
    for radii in rng.uniform(0.2, 1.0, (nlib_each, 3)):
        lib.add_convex(base * radii)
    for s in rng.uniform(0.2, 1.0, (nlib_each, 3)):
        lib.add_box(*map(float, s))
    for r, lz in zip(rng.uniform(0.1, 0.6, nlib_each), rng.uniform(0.2, 1.0, nlib_each)):
        lib.add_capsule(float(r), float(lz))
    n_all = len(lib)
    s1 = rng.integers(0, n_large, n)                       # always a large hull on one side ...
    s2 = rng.integers(0, n_all, n)
    swap = rng.random(n) < 0.5                             # ... which side is random
    s1, s2 = np.where(swap, s2, s1), np.where(swap, s1, s2)
    q1, T1, q2, T2 = _poses(rng, n, half_width)
    b = Batch("large_convex_" + kind, lib, s1, s2, q1, T1, q2, T2, kind)
    b.n_large = n_large
    return b


def mesh_vs_shapes(n=20_000, seed=1, seg=14, ring=14, n_variants=3, nper=16, half_width=1.2, flats_only=False):
    """BVHModel<OBBRSS> against the convex shape kinds, both operand orders, plus some mesh x mesh and
    shape x shape pairs: the BVH rows / columns of the collision matrix (collision_func_matrix.cpp:471-733)."""
    rng = _rng(seed, 9)
    meshes = mesh_variants(n_variants, seg, ring)
    lib = geometry.ShapeLibrary()
    for k, m in enumerate(meshes):
        lib.add_bvh(k, len(m.vertices))
    for s in rng.uniform(0.2, 0.9, (nper, 3)):
        lib.add_box(*map(float, s))
    for r in rng.uniform(0.1, 0.6, nper):
        lib.add_sphere(float(r))
    for r, lz in zip(rng.uniform(0.1, 0.4, nper), rng.uniform(0.2, 1.0, nper)):
        lib.add_capsule(float(r), float(lz))
    for r, lz in zip(rng.uniform(0.1, 0.5, nper), rng.uniform(0.2, 1.0, nper)):
        lib.add_cone(float(r), float(lz))
    for r, lz in zip(rng.uniform(0.1, 0.5, nper), rng.uniform(0.2, 1.0, nper)):
        lib.add_cylinder(float(r), float(lz))
    for r in rng.uniform(0.1, 0.7, (nper, 3)):
        lib.add_ellipsoid(*map(float, r))
    base = fibonacci_sphere(24)
    for radii in rng.uniform(0.1, 0.7, (nper, 3)):
        lib.add_convex(base * radii)
    n_solid_end = len(lib)
    for k in range(nper):  # Plane / Halfspace: unbounded BVs, every triangle of the mesh is tested
        nrm, d = rng.normal(size=3), float(rng.uniform(-0.6, 0.6))
        (lib.add_halfspace if k % 2 == 0 else lib.add_plane)(nrm, d)
    n_all = len(lib)
    u = rng.random(n)
    mesh = rng.integers(0, n_variants, n)
    solid = rng.integers(n_solid_end, n_all, n) if flats_only else rng.integers(n_variants, n_solid_end, n)
    s1 = np.where(u < 0.45, mesh, np.where(u < 0.9, solid, np.where(u < 0.95, mesh, solid)))
    s2 = np.where(u < 0.45, solid, np.where(u < 0.9, mesh, np.where(u < 0.95, rng.integers(0, n_variants, n), rng.integers(n_variants, n_solid_end, n))))
    q1, T1, q2, T2 = _poses(rng, n, half_width)
    b = Batch("mesh_vs_shapes_collide", lib, s1, s2, q1, T1, q2, T2, "collide")
    b.meshes = meshes
    return b


SOLID_KINDS = ("box", "sphere", "capsule", "cylinder", "cone", "ellipsoid", "convex32")


def mesh_vs_solid(kind, n=100_000, seed=1, seg=50, nper=64, half_width=1.6):
    """(mesh, solid) collide() / distance() queries, cfg4-size models (seg = ring = 50: 5 000 triangles): SURVEY.md 8(f3),
    the workload of tools/mesh_solid_bench.py (one solid kind) and of bench.py's `cfg4s` line (kind = "mixed": box, sphere,
    capsule, cylinder, ellipsoid, convex32 in equal parts).  The steps per query have a heavy tail (median 1: the boxes at
    the roots are disjoint; mean ~60; maximum in the thousands with hundreds of leaf tests)."""
    rng = _rng(seed, 77)
    meshes = mesh_variants(8, seg, seg)
    lib = geometry.ShapeLibrary()
    for k, m in enumerate(meshes):
        lib.add_bvh(k, len(m.vertices))
    n0 = len(lib)
    kinds = ("box", "sphere", "capsule", "cylinder", "ellipsoid", "convex32") if kind == "mixed" else tuple(kind.split(","))
    for kd in kinds:
        if kd not in SOLID_KINDS:
            raise ValueError(kd)
        for _ in range(nper):
            if kd == "box":
                lib.add_box(*map(float, rng.uniform(0.2, 0.9, 3)))
            elif kd == "sphere":
                lib.add_sphere(float(rng.uniform(0.1, 0.6)))
            elif kd == "capsule":
                lib.add_capsule(float(rng.uniform(0.1, 0.4)), float(rng.uniform(0.2, 1.0)))
            elif kd == "cylinder":
                lib.add_cylinder(float(rng.uniform(0.1, 0.5)), float(rng.uniform(0.2, 1.0)))
            elif kd == "cone":
                lib.add_cone(float(rng.uniform(0.1, 0.5)), float(rng.uniform(0.2, 1.0)))
            elif kd == "ellipsoid":
                lib.add_ellipsoid(*map(float, rng.uniform(0.1, 0.7, 3)))
            else:
                lib.add_convex(fibonacci_sphere(32) * rng.uniform(0.1, 0.7, 3))
    s1 = rng.integers(0, 8, n)
    s2 = rng.integers(n0, n0 + nper * len(kinds), n)
    q1, T1, q2, T2 = _poses(rng, n, half_width)
    b = Batch("mesh_x_" + kind.replace(",", "_"), lib, s1, s2, q1, T1, q2, T2, "collide")
    b.meshes = meshes
    return b


def mixed_scene(n=200_000, seed=1, frac_solid=0.4, frac_mesh_solid=0.4):
    """One batch with every kind of pair a scene of meshes and solids produces -- solid x solid (frac_solid; the six solid kinds of
    mesh_vs_solid("mixed") among themselves), mesh x solid in both operand orders (frac_mesh_solid), mesh x mesh (the rest) -- on cfg4-size
    models: bench.py's `cfgmix` row (what the mesh walks beside the solids' kernels are for; SURVEY.md 8 f3 + configs[3] + configs[4])."""
    b = mesh_vs_solid("mixed", n=n, seed=seed)
    rng = _rng(seed, 91)
    n_lib = len(b.lib)
    u = rng.uniform(size=n)
    mesh_a, mesh_b = rng.integers(0, 8, n), rng.integers(0, 8, n)
    sol_a, sol_b = rng.integers(8, n_lib, n), rng.integers(8, n_lib, n)
    swap = rng.uniform(size=n) < 0.5
    solid = u < frac_solid
    ms = (~solid) & (u < frac_solid + frac_mesh_solid)
    mm = ~(solid | ms)
    s1 = np.where(solid, sol_a, np.where(ms, np.where(swap, sol_a, mesh_a), mesh_a))
    s2 = np.where(solid, sol_b, np.where(ms, np.where(swap, mesh_b, sol_b), mesh_b))
    b.s1, b.s2 = s1.astype(b.s1.dtype), s2.astype(b.s2.dtype)
    b.name = "mixed_scene_collide"
    b.mix = {"solid_x_solid": float(solid.mean()), "mesh_x_solid": float(ms.mean()), "mesh_x_mesh": float(mm.mean())}
    return b


def cfg5_mixed(n=100_000, seed=1, nper=256, half_width=0.8):
    """cfg5-style mixed primitive+convex pairs (type mix 20 % each of Box/Sphere/Capsule/
    Ellipsoid/Convex32), synthetic pair list (cfg5_broadphase_scene takes its pairs from the host broadphase)."""
    rng = _rng(seed, 5)
    lib = _mixed_library(rng, nper)
    s1, s2 = rng.integers(0, 5 * nper, n), rng.integers(0, 5 * nper, n)
    q1, T1, q2, T2 = _poses(rng, n, half_width)
    return Batch("cfg5_mixed_collide", lib, s1, s2, q1, T1, q2, T2, "collide")


def cfg5_broadphase_scene(n_objects=100_000, target_pairs=1_000_000, seed=1, nper=256, n_threads=0):
    """cfg5: a scene of posed {Box, Sphere, Capsule, Ellipsoid, Convex32} objects (20 % each); the
    pair list comes from the host broadphase (engine.world_aabbs -> engine.broadphase_self_pairs,
    i.e. what DynamicAABBTreeCollisionManager::collide + CollisionCallBackCollect would collect).
    The cube side is chosen so that about `target_pairs` AABB pairs overlap."""
    from . import engine
    rng = _rng(seed, 55)
    lib = _mixed_library(rng, nper)
    obj_shape = rng.integers(0, 5 * nper, n_objects).astype(np.uint32)
    quat = uniform_quaternions(rng, n_objects)
    # expected pairs ~ n^2/2 * E[prod_k (w_ik + w_jk)/2 * 2] / L^3; the constant is measured on this library
    side = (n_objects ** 2 * 30.5 / (2.0 * max(target_pairs, 1))) ** (1.0 / 3.0)
    T = rng.uniform(-side / 2, side / 2, (n_objects, 3))
    tf = geometry.make_pose(quat=quat, T=T)
    aabbs = engine.world_aabbs(lib, obj_shape, tf, n_threads)
    pairs = engine.broadphase_self_pairs(aabbs, n_threads)
    i, j = pairs[:, 0], pairs[:, 1]
    b = Batch("cfg5_broadphase_mixed_collide", lib, obj_shape[i], obj_shape[j], quat[i], T[i], quat[j], T[j], "collide")
    b.scene = dict(n_objects=n_objects, side=side, aabbs=aabbs, pairs=pairs, obj_shape=obj_shape, obj_tf=tf)
    return b


_MESH_CACHE = {}


def mesh_variants(n_variants=8, seg=50, ring=50):
    """cfg4 mesh library: perturbed UV spheres (seg=ring=50 -> 5000 triangles, 2502 vertices, 9999 nodes)."""
    from . import bvh_builder
    key = (n_variants, seg, ring)
    if key not in _MESH_CACHE:
        _MESH_CACHE[key] = [bvh_builder.Mesh(*bvh_builder.bumpy_sphere(seg, ring, r=1.0, amp=0.12 + 0.01 * k,
                                                                       freq=2 + (k % 3), phase=0.7 * k))
                            for k in range(n_variants)]
    return _MESH_CACHE[key]


def cfg4_mesh_mesh(n=100_000, seed=1, n_variants=8, seg=50, ring=50, half_width=1.25):
    """cfg4: BVHModel<OBBRSS> x BVHModel<OBBRSS> collide(), default request (first contact)."""
    rng = _rng(seed, 4)
    meshes = mesh_variants(n_variants, seg, ring)
    lib = geometry.ShapeLibrary()
    for k, m in enumerate(meshes):
        lib.add_bvh(k, len(m.vertices))
    s1, s2 = rng.integers(0, n_variants, n), rng.integers(0, n_variants, n)
    q1, T1, q2, T2 = _poses(rng, n, half_width)
    b = Batch("cfg4_mesh_mesh_collide_%dtri" % (2 * seg * ring), lib, s1, s2, q1, T1, q2, T2, "collide")
    b.meshes = meshes
    return b


def cfg4_mesh_mesh_distance(n=100_000, seed=1, n_variants=8, seg=50, ring=50, half_width=2.2):
    """cfg4's distance() variant ("100 000 mesh-mesh collide() (and distance())"): the same models, poses
    spread so that about two thirds of the pairs are separated (a distance() on intersecting meshes ends at the first
    overlapping triangle pair)."""
    b = cfg4_mesh_mesh(n=n, seed=seed, n_variants=n_variants, seg=seg, ring=ring, half_width=half_width)
    b.kind = "distance"
    b.name = "cfg4_mesh_mesh_distance_%dtri" % (2 * seg * ring)
    return b


def make_library(pkg, batch, device=0, options=None):
    """engine.Library for a batch (registers the batch's meshes, if any); options: {key: value} for hfcl_lib_set_option."""
    lib = pkg.Library(batch.lib, device=device, options=options)
    for m in getattr(batch, "meshes", []) or []:
        lib.add_bvh(m)
    return lib


def make_request(batch, abi, **kw):
    req = abi.default_distance_request() if batch.kind == "distance" else abi.default_collision_request()
    over = dict(batch.request_overrides)
    over.update(kw)
    for k, v in over.items():
        if hasattr(req, k):
            setattr(req, k, v)
        else:
            setattr(req.q, k, v)
    return req
