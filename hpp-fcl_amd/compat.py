"""`hppfcl`-named Python surface over the C ABI: the names, argument meaning and error behaviour of the
reference's Python module (python/collision.cc:257-266, python/distance.cc:150-159, python/collision-geometries.cc,
python/math.cc) for the path in scope, so that code written against `import hppfcl` -- e.g. the reference's
test/python_unit/api.py, collision.py, collision_manager.py -- runs on `import hppfcl_amd.compat as hppfcl`.

Plumbing only: every query is a batch of one (or, through `collide_pairs` / the collision manager, one batch)
on the HIP library.  No CPU fallback."""
import os

import numpy as np

from . import abi, bvh_builder, engine, geometry


class NODE_TYPE:  # include/hpp/fcl/collision_object.h:65-89
    BV_OBBRSS = abi.BV_OBBRSS
    GEOM_BOX, GEOM_SPHERE, GEOM_CAPSULE, GEOM_CONE = abi.GEOM_BOX, abi.GEOM_SPHERE, abi.GEOM_CAPSULE, abi.GEOM_CONE
    GEOM_CYLINDER, GEOM_CONVEX, GEOM_PLANE = abi.GEOM_CYLINDER, abi.GEOM_CONVEX, abi.GEOM_PLANE
    GEOM_HALFSPACE, GEOM_TRIANGLE, GEOM_ELLIPSOID = abi.GEOM_HALFSPACE, abi.GEOM_TRIANGLE, abi.GEOM_ELLIPSOID


class GJKInitialGuess:
    DefaultGuess, CachedGuess, BoundingVolumeGuess = abi.DefaultGuess, abi.CachedGuess, abi.BoundingVolumeGuess


class GJKVariant:
    DefaultGJK, PolyakAcceleration, NesterovAcceleration = abi.DefaultGJK, abi.PolyakAcceleration, abi.NesterovAcceleration


class GJKConvergenceCriterion:
    Default, DualityGap, Hybrid = abi.Default, abi.DualityGap, abi.Hybrid


class GJKConvergenceCriterionType:
    Relative, Absolute = abi.Relative, abi.Absolute


def _v3(v):
    a = np.asarray(v, dtype=np.float64).reshape(-1)
    if a.shape != (3,):
        raise ValueError("expected a 3-vector")
    return a.copy()


class Transform3f:  # include/hpp/fcl/math/transform.h:56-218
    def __init__(self, R=None, T=None):
        self._R = np.eye(3) if R is None else np.asarray(R, dtype=np.float64).reshape(3, 3).copy()
        self._T = np.zeros(3) if T is None else _v3(T)

    @staticmethod
    def Identity():
        return Transform3f()

    def getRotation(self):
        return self._R

    def getTranslation(self):
        return self._T

    def setRotation(self, R):
        self._R = np.asarray(R, dtype=np.float64).reshape(3, 3).copy()

    def setTranslation(self, T):
        self._T = _v3(T)

    def setTransform(self, R, T):
        self.setRotation(R)
        self.setTranslation(T)

    def transform(self, v):
        return self._R @ _v3(v) + self._T

    def inverse(self):
        return Transform3f(self._R.T, -self._R.T @ self._T)

    def __mul__(self, other):  # transform.h:187-190
        return Transform3f(self._R @ other._R, self._R @ other._T + self._T)

    def _abi(self):
        return geometry.make_pose(R=self._R, T=self._T)


class CollisionGeometry:
    def getNodeType(self):
        return self._node_type


class ShapeBase(CollisionGeometry):
    _swept = 0.0

    def getSweptSphereRadius(self):
        return self._swept

    def setSweptSphereRadius(self, r):
        if r < 0:
            raise ValueError("Swept-sphere radius must be positive.")  # geometric_shapes.h:74-80
        self._swept = float(r)


class Box(ShapeBase):
    _node_type = NODE_TYPE.GEOM_BOX

    def __init__(self, x, y=None, z=None):
        side = _v3(x) if y is None else np.array([x, y, z], dtype=np.float64)
        self.halfSide = side / 2.0

    def _register(self, L):
        return L.add_box(*(2.0 * np.asarray(self.halfSide, dtype=np.float64)), swept_sphere_radius=self._swept)


class Sphere(ShapeBase):
    _node_type = NODE_TYPE.GEOM_SPHERE

    def __init__(self, radius):
        self.radius = float(radius)

    def _register(self, L):
        return L.add_sphere(self.radius, self._swept)


class Ellipsoid(ShapeBase):
    _node_type = NODE_TYPE.GEOM_ELLIPSOID

    def __init__(self, rx, ry=None, rz=None):
        self.radii = _v3(rx) if ry is None else np.array([rx, ry, rz], dtype=np.float64)

    def _register(self, L):
        return L.add_ellipsoid(*map(float, self.radii), swept_sphere_radius=self._swept)


class _RadiusHalfLength(ShapeBase):
    def __init__(self, radius, lz):
        self.radius = float(radius)
        self.halfLength = float(lz) / 2.0  # geometric_shapes.h:386-387


class Capsule(_RadiusHalfLength):
    _node_type = NODE_TYPE.GEOM_CAPSULE

    def _register(self, L):
        return L.add_capsule(self.radius, 2.0 * self.halfLength, self._swept)


class Cone(_RadiusHalfLength):
    _node_type = NODE_TYPE.GEOM_CONE

    def _register(self, L):
        return L.add_cone(self.radius, 2.0 * self.halfLength, self._swept)


class Cylinder(_RadiusHalfLength):
    _node_type = NODE_TYPE.GEOM_CYLINDER

    def _register(self, L):
        return L.add_cylinder(self.radius, 2.0 * self.halfLength, self._swept)


class _Flat(ShapeBase):
    def __init__(self, n=(1.0, 0.0, 0.0), d=0.0, *rest):
        if rest:  # (a, b, c, d)
            n, d = (n, d, rest[0]), rest[1]
        self.n, self.d = _v3(n), float(d)
        l = np.linalg.norm(self.n)  # unitNormalTest, geometric_shapes.cpp:121-143
        if l > 0:
            self.n, self.d = self.n / l, self.d / l
        else:
            self.n, self.d = np.array([1.0, 0.0, 0.0]), 0.0


class Halfspace(_Flat):
    _node_type = NODE_TYPE.GEOM_HALFSPACE

    def _register(self, L):
        return L.add_halfspace(self.n, self.d, self._swept)


class Plane(_Flat):
    _node_type = NODE_TYPE.GEOM_PLANE

    def _register(self, L):
        return L.add_plane(self.n, self.d, self._swept)


class TriangleP(ShapeBase):
    _node_type = NODE_TYPE.GEOM_TRIANGLE

    def __init__(self, a, b, c):
        self.a, self.b, self.c = _v3(a), _v3(b), _v3(c)

    def _register(self, L):
        return L.add_triangle(self.a, self.b, self.c, self._swept)


class Triangle:  # include/hpp/fcl/data_types.h:101-144
    def __init__(self, p1=0, p2=0, p3=0):
        self.vids = [int(p1), int(p2), int(p3)]

    def __getitem__(self, i):
        return self.vids[i]


class StdVec_Vec3f(list):
    pass


class StdVec_Triangle(list):
    pass


class Convex(ShapeBase):
    """Convex<Triangle>(points, triangles).  Up to 32 vertices the narrow phase scans the points in registers, above it
    scans them from memory; with facets given, hulls of HFCL_CLIMB_MIN (512) vertices and more climb the vertex adjacency
    the facets define, as the reference does (fillNeighbors, shape/details/convex.hxx:231-280; getShapeSupportLog)."""
    _node_type = NODE_TYPE.GEOM_CONVEX

    def __init__(self, points, polygons=None):
        self.points = np.array([_v3(p) for p in points], dtype=np.float64)
        self.num_points = len(self.points)
        self.polygons = list(polygons) if polygons is not None else []

    def neighbors(self):
        """ConvexBase::neighbors as CSR (offsets[num_points + 1], ids), or None without facets.  Cached per object, keyed on the facets' indices."""
        if not self.polygons:
            return None
        polys = [tuple(int(poly[k]) for k in range(3)) if isinstance(poly, Triangle) else tuple(int(k) for k in poly) for poly in self.polygons]
        key = (len(self.points), hash(tuple(polys)))  # the facets themselves: an edit in place must not leave a stale adjacency behind
        if getattr(self, "_nb_cache", None) is not None and self._nb_cache[0] == key:
            return self._nb_cache[1]
        npts = len(self.points)
        nb = [set() for _ in range(npts)]
        for idx in polys:
            n = len(idx)
            for j in range(n):
                nb[idx[j]].add(idx[j - 1])
                nb[idx[j]].add(idx[(j + 1) % n])
        offs = np.zeros(npts + 1, dtype=np.uint32)
        offs[1:] = np.cumsum([len(x) for x in nb])
        out = (offs, np.array([v for x in nb for v in sorted(x)], dtype=np.uint32))
        self._nb_cache = (key, out)
        return out

    def _register(self, L):
        return L.add_convex(self.points, self._swept)


class BVHModelOBBRSS(CollisionGeometry):  # BVHModel<OBBRSS>, src/BVH/BVH_model.cpp:264-576
    _node_type = NODE_TYPE.BV_OBBRSS

    def __init__(self):
        self._v, self._t, self._mesh = [], [], None

    def beginModel(self, num_tris=0, num_vertices=0):
        self._v, self._t, self._mesh = [], [], None
        return 0

    def addVertex(self, p):
        self._v.append(_v3(p))
        return 0

    def addTriangle(self, p1, p2, p3):
        base = len(self._v)
        self._v += [_v3(p1), _v3(p2), _v3(p3)]
        self._t.append((base, base + 1, base + 2))
        return 0

    def addSubModel(self, vertices, triangles=None):
        base = len(self._v)
        self._v += [_v3(p) for p in vertices]
        for t in ([] if triangles is None else triangles):
            self._t.append((base + int(t[0]), base + int(t[1]), base + int(t[2])))
        return 0

    def endModel(self):
        if not self._t:
            return -3  # BVH_ERR_BUILD_EMPTY_MODEL
        self._mesh = bvh_builder.Mesh(np.array(self._v), np.array(self._t, dtype=np.uint32))
        return 0

    @property
    def num_vertices(self):
        return len(self._v)

    @property
    def num_tris(self):
        return len(self._t)


class _Query:
    def __init__(self):
        self.gjk_initial_guess = GJKInitialGuess.DefaultGuess
        self.cached_gjk_guess = np.array([1.0, 0.0, 0.0])
        self.cached_support_func_guess = np.zeros(2, dtype=np.int32)
        self.gjk_max_iterations, self.gjk_tolerance = 128, 1e-6
        self.gjk_variant = GJKVariant.DefaultGJK
        self.gjk_convergence_criterion = GJKConvergenceCriterion.Default
        self.gjk_convergence_criterion_type = GJKConvergenceCriterionType.Relative
        self.epa_max_iterations, self.epa_tolerance = 64, 1e-6
        self.enable_timings = False
        self.collision_distance_threshold = 1e-12

    def _fill(self, q):
        q.gjk_initial_guess, q.gjk_variant = int(self.gjk_initial_guess), int(self.gjk_variant)
        q.gjk_convergence_criterion = int(self.gjk_convergence_criterion)
        q.gjk_convergence_criterion_type = int(self.gjk_convergence_criterion_type)
        q.gjk_max_iterations, q.epa_max_iterations = int(self.gjk_max_iterations), int(self.epa_max_iterations)
        q.gjk_tolerance, q.epa_tolerance = float(self.gjk_tolerance), float(self.epa_tolerance)
        q.collision_distance_threshold = float(self.collision_distance_threshold)
        for k in range(3):
            q.cached_gjk_guess[k] = float(self.cached_gjk_guess[k])
        for k in range(2):
            q.cached_support_func_guess[k] = int(self.cached_support_func_guess[k])

    def updateGuess(self, result):  # collision_data.h:258-265
        if self.gjk_initial_guess == GJKInitialGuess.CachedGuess:
            self.cached_gjk_guess = np.array(result.cached_gjk_guess)
            self.cached_support_func_guess = np.array(result.cached_support_func_guess)


class CollisionRequest(_Query):  # collision_data.h:312-366
    def __init__(self, flag=None, num_max_contacts=1):
        super().__init__()
        self.num_max_contacts, self.enable_contact = int(num_max_contacts), True
        self.security_margin, self.break_distance = 0.0, 1e-3
        self.distance_upper_bound = np.finfo(np.float64).max

    def _abi(self):
        r = abi.default_collision_request()
        self._fill(r.q)
        r.num_max_contacts, r.enable_contact = int(self.num_max_contacts), int(bool(self.enable_contact))
        r.security_margin, r.break_distance = float(self.security_margin), float(self.break_distance)
        r.distance_upper_bound = float(self.distance_upper_bound)
        return r


class DistanceRequest(_Query):  # collision_data.h:987-1031
    def __init__(self, enable_nearest_points=True, enable_signed_distance=True, rel_err=0.0, abs_err=0.0):
        super().__init__()
        self.enable_nearest_points, self.enable_signed_distance = enable_nearest_points, enable_signed_distance
        self.rel_err, self.abs_err = rel_err, abs_err

    def _abi(self):
        r = abi.default_distance_request()
        self._fill(r.q)
        r.enable_nearest_points, r.enable_signed_distance = int(bool(self.enable_nearest_points)), int(bool(self.enable_signed_distance))
        r.rel_err, r.abs_err = float(self.rel_err), float(self.abs_err)
        return r


_NAN3 = np.full(3, np.nan)


class Contact:  # collision_data.h:59-166
    NONE = -1

    def __init__(self, o1=None, o2=None, b1=-1, b2=-1, p1=_NAN3, p2=_NAN3, normal=_NAN3, depth=np.finfo(np.float64).max):
        self.o1, self.o2, self.b1, self.b2 = o1, o2, int(b1), int(b2)
        self.nearest_points = [np.array(p1), np.array(p2)]
        self.normal = np.array(normal)
        self.pos = (self.nearest_points[0] + self.nearest_points[1]) / 2.0
        self.penetration_depth = float(depth)

    def getNearestPoint1(self):
        return self.nearest_points[0]

    def getNearestPoint2(self):
        return self.nearest_points[1]


class CollisionResult:  # collision_data.h:391-494
    def __init__(self):
        self.clear()

    def clear(self):
        self._contacts = []
        self.distance_lower_bound = np.finfo(np.float64).max
        self.normal = _NAN3.copy()
        self.nearest_points = [_NAN3.copy(), _NAN3.copy()]
        self.cached_gjk_guess = np.array([1.0, 0.0, 0.0])
        self.cached_support_func_guess = np.zeros(2, dtype=np.int32)

    def isCollision(self):
        return len(self._contacts) > 0

    def numContacts(self):
        return len(self._contacts)

    def getContact(self, i):
        if not self._contacts:
            raise RuntimeError("The number of contacts is zero. No Contact can be returned.")  # :449-456
        return self._contacts[min(i, len(self._contacts) - 1)]

    def getContacts(self):
        return list(self._contacts)

    def addContact(self, c):
        self._contacts.append(c)

    def getNearestPoint1(self):
        return self.nearest_points[0]

    def getNearestPoint2(self):
        return self.nearest_points[1]


class DistanceResult:  # collision_data.h:1053-1174
    def __init__(self):
        self.clear()

    def clear(self):
        self.min_distance = np.finfo(np.float64).max
        self.normal = _NAN3.copy()
        self.nearest_points = [_NAN3.copy(), _NAN3.copy()]
        self.o1 = self.o2 = None
        self.b1 = self.b2 = -1
        self.cached_gjk_guess = np.array([1.0, 0.0, 0.0])
        self.cached_support_func_guess = np.zeros(2, dtype=np.int32)

    def getNearestPoint1(self):
        return self.nearest_points[0]

    def getNearestPoint2(self):
        return self.nearest_points[1]


class _Context:
    """Shape table of every geometry seen so far + the device library built from it (rebuilt when it grows)."""

    def __init__(self, device=0):
        self.device, self.L, self.lib = device, geometry.ShapeLibrary(), None
        self.ids, self.keep, self.meshes = {}, [], []
        self.built = 0

    @staticmethod
    def _signature(g):
        if isinstance(g, BVHModelOBBRSS):
            return ("bvh", id(g._mesh))
        vals = []
        for k, v in sorted(vars(g).items()):
            if k not in ("polygons", "_nb_cache"):  # (the cached adjacency is derived from the polygons)
                vals.append((k, np.asarray(v, dtype=np.float64).tobytes() if not isinstance(v, list) else None))
        return (type(g).__name__, tuple(vals))

    def add(self, g):
        sig = self._signature(g)
        hit = self.ids.get(id(g))
        if hit and hit[1] == sig:
            return hit[0]
        if isinstance(g, BVHModelOBBRSS):
            if g._mesh is None:
                raise ValueError("BVHModel: endModel() has not been called")
            self.meshes.append(g._mesh)
            sid = self.L.add_bvh(len(self.meshes) - 1, g.num_vertices)
        elif isinstance(g, CollisionGeometry) and hasattr(g, "_register"):
            sid = g._register(self.L)
        else:
            raise ValueError("unsupported collision geometry")
        self.ids[id(g)] = (sid, sig)
        self.keep.append(g)  # ids stay valid while the context lives
        return sid

    def library(self):
        if self.lib is None or self.built != len(self.L):
            if self.lib is not None:
                self.lib.close()
            self.lib = engine.Library(self.L, device=self.device)
            for m in self.meshes:
                self.lib.add_bvh(m)
            # large hulls with facets: the adjacency the device climbs.  Only hulls the engine will climb (its own threshold,
            # hfcl_lib_climb_min: below that the scan is faster) pay the host loop and the device copy.
            climb_min = self.lib.climb_min()
            for g in self.keep:
                if isinstance(g, Convex) and g.num_points >= max(33, climb_min):
                    nb = g.neighbors()
                    if nb is not None:
                        self.lib.set_convex_neighbors(self.ids[id(g)][0], *nb)
            self.built = len(self.L)
        return self.lib


_ctx = None


def _context():
    global _ctx
    if _ctx is None:
        _ctx = _Context()
    return _ctx


def _check_pair(o1, o2, for_distance):
    if not engine.dll().hfcl_pair_supported(int(o1.getNodeType()), int(o2.getNodeType()), int(for_distance)):
        raise ValueError("%s function between node type %d and node type %d is not yet supported." %
                         ("Distance" if for_distance else "Collision", o1.getNodeType(), o2.getNodeType()))


def _run(kind, pairs, request):
    """pairs: list of (o1, tf1, o2, tf2) -> (records, guesses, contacts or None)"""
    ctx = _context()
    ids = [(ctx.add(o1), ctx.add(o2)) for o1, _, o2, _ in pairs]
    lib = ctx.library()
    s1 = np.array([a for a, _ in ids], dtype=np.uint32)
    s2 = np.array([b for _, b in ids], dtype=np.uint32)
    tf1 = np.concatenate([p[1]._abi().reshape(1, 12) for p in pairs])
    tf2 = np.concatenate([p[3]._abi().reshape(1, 12) for p in pairs])
    req = request._abi()
    try:
        if kind == "distance":
            rec, g = lib.distance(s1, s2, tf1, tf2, req, want_guess=True)
            return rec, g, None
        if request.num_max_contacts > 1 and any(isinstance(p[0], BVHModelOBBRSS) or isinstance(p[2], BVHModelOBBRSS) for p in pairs):
            cap = int(min(request.num_max_contacts, 1 << 16)) * len(pairs)
            rec, contacts, _ = lib.collide_contacts(s1, s2, tf1, tf2, req, cap)
            return rec, None, contacts
        rec, g = lib.collide(s1, s2, tf1, tf2, req, want_guess=True)
        return rec, g, None
    except engine.EngineError as e:
        if e.code in (abi.ERR_INVALID_ARGUMENT, abi.ERR_UNSUPPORTED_PAIR):
            raise ValueError(str(e))  # std::invalid_argument in the reference
        raise


def _fill_collision(result, o1, o2, request, rec, guess, contacts, pair_index=0):
    if abi.status_skipped(rec["status"]):
        return
    d = float(rec["distance"])
    dtc = d - request.security_margin
    if dtc < result.distance_lower_bound:  # updateDistanceLowerBoundFromLeaf, collision_data.h:1186-1197
        result.distance_lower_bound = dtc
        result.normal = np.array(rec["normal"])
        result.nearest_points = [np.array(rec["p1"]), np.array(rec["p2"])]
    if contacts is not None:
        for c in contacts[contacts["pair"] == pair_index]:
            if result.numContacts() < request.num_max_contacts:
                result.addContact(Contact(o1, o2, c["b1"], c["b2"], c["p1"], c["p2"], c["normal"], c["penetration_depth"]))
    elif rec["num_contacts"] > 0 and result.numContacts() < request.num_max_contacts:
        result.addContact(Contact(o1, o2, rec["b1"], rec["b2"], rec["p1"], rec["p2"], rec["normal"], d))
    if guess is not None:
        result.cached_gjk_guess = np.array(guess["gjk_guess"])
        result.cached_support_func_guess = np.array(guess["support_guess"])
        request.updateGuess(result)


def collide(*args):
    """collide(o1, tf1, o2, tf2, request, result) or collide(obj1, obj2, request, result)  (python/collision.cc:257-266)"""
    if len(args) == 4:
        a, b, request, result = args
        o1, tf1, o2, tf2 = a.collisionGeometry(), a.getTransform(), b.collisionGeometry(), b.getTransform()
    else:
        o1, tf1, o2, tf2, request, result = args
    if request.security_margin == -np.inf:  # src/collision.cpp:73-76
        result.clear()
        return 0
    if request.num_max_contacts == 0:
        raise ValueError("Invalid number of max contacts (current value is 0).")
    _check_pair(o1, o2, False)
    if result.isCollision() and request.num_max_contacts <= result.numContacts():
        return result.numContacts()
    rec, g, contacts = _run("collide", [(o1, tf1, o2, tf2)], request)
    _fill_collision(result, o1, o2, request, rec[0], g[0] if g is not None else None, contacts)
    return result.numContacts()


def distance(*args):
    """distance(o1, tf1, o2, tf2, request, result) or distance(obj1, obj2, request, result)  (python/distance.cc:150-159)"""
    if len(args) == 4:
        a, b, request, result = args
        o1, tf1, o2, tf2 = a.collisionGeometry(), a.getTransform(), b.collisionGeometry(), b.getTransform()
    else:
        o1, tf1, o2, tf2, request, result = args
    _check_pair(o1, o2, True)
    if result.min_distance <= 0:  # DistanceRequest::isSatisfied
        return result.min_distance
    rec, g, _ = _run("distance", [(o1, tf1, o2, tf2)], request)
    r = rec[0]
    d = float(r["distance"])
    if result.min_distance > d:  # DistanceResult::update
        result.min_distance, result.o1, result.o2 = d, o1, o2
        result.b1, result.b2 = int(r["b1"]), int(r["b2"])
        result.normal = np.array(r["normal"])
        result.nearest_points = [np.array(r["p1"]), np.array(r["p2"])]
    result.cached_gjk_guess = np.array(g[0]["gjk_guess"])
    result.cached_support_func_guess = np.array(g[0]["support_guess"])
    request.updateGuess(result)
    return d


class ComputeCollision:  # collision.h:79-117
    def __init__(self, o1, o2):
        _check_pair(o1, o2, False)
        self.o1, self.o2 = o1, o2

    def __call__(self, tf1, tf2, request, result):
        return collide(self.o1, tf1, self.o2, tf2, request, result)


class ComputeDistance:  # distance.h:74-112
    def __init__(self, o1, o2):
        _check_pair(o1, o2, True)
        self.o1, self.o2 = o1, o2

    def __call__(self, tf1, tf2, request, result):
        return distance(self.o1, tf1, self.o2, tf2, request, result)


# ---- broadphase hand-off (collision_object.h:215-357, broadphase_callbacks.h, default_broadphase_callbacks.h) ----
class CollisionObject:
    def __init__(self, cgeom, tf=None):
        self._g, self._tf = cgeom, tf if tf is not None else Transform3f()

    def collisionGeometry(self):
        return self._g

    def getTransform(self):
        return self._tf

    def setTransform(self, tf):
        self._tf = tf

    def getTranslation(self):
        return self._tf.getTranslation()

    def setTranslation(self, T):
        self._tf.setTranslation(T)


class CollisionCallBackCollect:  # default_broadphase_callbacks.h:200-230
    def __init__(self, max_size):
        self.max_size, self._pairs = int(max_size), []

    def collide(self, o1, o2):
        if len(self._pairs) < self.max_size:
            self._pairs.append((o1, o2))
        return False

    def numCollisionPairs(self):
        return len(self._pairs)

    def getCollisionPairs(self):
        return list(self._pairs)

    def init(self):
        self._pairs = []


class _CollisionData:
    def __init__(self):
        self.request, self.result, self.done = CollisionRequest(), CollisionResult(), False


class CollisionCallBackDefault:  # default_broadphase_callbacks.h:129-145: one shared, accumulating CollisionResult
    def __init__(self):
        self.data = _CollisionData()

    def init(self):
        self.data.result.clear()
        self.data.done = False


def collide_pairs(pairs, request):
    """Batched narrow phase on collected (CollisionObject, CollisionObject) pairs: ONE device call; one fresh
    CollisionResult per pair."""
    if not pairs:
        return []
    for a, b in pairs:
        _check_pair(a.collisionGeometry(), b.collisionGeometry(), False)
    quad = [(a.collisionGeometry(), a.getTransform(), b.collisionGeometry(), b.getTransform()) for a, b in pairs]
    rec, g, contacts = _run("collide", quad, request)
    out = []
    for i, (a, b) in enumerate(pairs):
        r = CollisionResult()
        _fill_collision(r, a.collisionGeometry(), b.collisionGeometry(), request, rec[i], g[i] if g is not None else None,
                        contacts, i)
        out.append(r)
    return out


class DynamicAABBTreeCollisionManager:  # broadphase_dynamic_AABB_tree.h (candidate set = all AABB-overlapping pairs)
    def __init__(self):
        self._objs, self._aabbs = [], None

    def registerObject(self, obj):
        self._objs.append(obj)
        self._aabbs = None

    def registerObjects(self, objs):
        for o in objs:
            self.registerObject(o)

    def unregisterObject(self, obj):
        self._objs = [o for o in self._objs if o is not obj]
        self._aabbs = None

    def size(self):
        return len(self._objs)

    def empty(self):
        return not self._objs

    def clear(self):
        self._objs, self._aabbs = [], None

    def update(self, *_):
        self._aabbs = None

    def _world_aabbs(self, objs):
        L = geometry.ShapeLibrary()
        ids = np.array([o.collisionGeometry()._register(L) for o in objs], dtype=np.uint32)
        tf = np.concatenate([o.getTransform()._abi().reshape(1, 12) for o in objs])
        return engine.world_aabbs(L, ids, tf)

    def setup(self):
        self._aabbs = self._world_aabbs(self._objs) if self._objs else np.zeros((0, 6))

    def _candidates(self, other=None):
        if self._aabbs is None:
            self.setup()
        if other is None:
            idx = engine.broadphase_self_pairs(self._aabbs) if len(self._objs) > 1 else np.zeros((0, 2), dtype=np.uint32)
            return [(self._objs[i], self._objs[j]) for i, j in idx]
        box = self._world_aabbs([other])
        idx = engine.broadphase_pairs_between(box, self._aabbs) if self._objs else np.zeros((0, 2), dtype=np.uint32)
        return [(other, self._objs[j]) for _, j in idx]

    def collide(self, *args):
        """collide(callback): self pairs; collide(obj, callback): obj against the managed objects."""
        other, callback = (None, args[0]) if len(args) == 1 else args
        cand = self._candidates(other)
        if isinstance(callback, CollisionCallBackDefault):
            d = callback.data
            for (a, b), r in zip(cand, collide_pairs(cand, d.request)):
                for c in r.getContacts():  # the default callback accumulates into one result
                    if d.result.numContacts() < d.request.num_max_contacts:
                        d.result.addContact(c)
                d.result.distance_lower_bound = min(d.result.distance_lower_bound, r.distance_lower_bound)
                if d.result.isCollision() and d.result.numContacts() >= d.request.num_max_contacts:
                    d.done = True
                    break
            return
        for a, b in cand:
            if callback.collide(a, b):
                break
