"""`hppfcl`-named Python surface (hpp-fcl_amd/compat.py): the reference's own Python unit tests
(test/python_unit/api.py, collision.py, collision_manager.py, the constructor parts of geometric_shapes.py),
restated with `hppfcl` bound to the compat module."""
import numpy as np
import pytest


@pytest.fixture(scope="module")
def hppfcl(pkg):
    import importlib
    return importlib.import_module("hppfcl_amd.compat")


def test_geometric_shapes_constructors(hppfcl):  # geometric_shapes.py:8-20, 51-60, 79-99, 101-..., CPU only
    capsule = hppfcl.Capsule(1.0, 2.0)
    assert isinstance(capsule, hppfcl.ShapeBase) and isinstance(capsule, hppfcl.CollisionGeometry)
    assert capsule.getNodeType() == hppfcl.NODE_TYPE.GEOM_CAPSULE
    assert capsule.radius == 1.0 and capsule.halfLength == 1.0
    box = hppfcl.Box(np.array([1.0, 2.0, 3.0]))
    assert box.getNodeType() == hppfcl.NODE_TYPE.GEOM_BOX and np.array_equal(box.halfSide, [0.5, 1.0, 1.5])
    box2 = hppfcl.Box(1.0, 2.0, 3)
    assert (box2.halfSide[0], box2.halfSide[1], box2.halfSide[2]) == (0.5, 1.0, 1.5)
    assert hppfcl.Sphere(1.0).radius == 1.0 and hppfcl.Sphere(1.0).getNodeType() == hppfcl.NODE_TYPE.GEOM_SPHERE
    cyl, cone = hppfcl.Cylinder(1.0, 2.0), hppfcl.Cone(1.0, 2.0)
    assert (cyl.radius, cyl.halfLength, cone.radius, cone.halfLength) == (1.0, 1.0, 1.0, 1.0)
    assert cyl.getNodeType() == hppfcl.NODE_TYPE.GEOM_CYLINDER and cone.getNodeType() == hppfcl.NODE_TYPE.GEOM_CONE
    hs = hppfcl.Halfspace(np.array((0, 0, 2.0)), 4.0)  # normalised by the constructor (unitNormalTest)
    assert np.allclose(hs.n, [0, 0, 1]) and hs.d == 2.0
    M = hppfcl.Transform3f(np.eye(3), np.array([1.0, 2, 3])) * hppfcl.Transform3f.Identity()
    assert np.array_equal(M.getTranslation(), [1, 2, 3]) and np.array_equal(M.getRotation(), np.eye(3))


def test_unsupported_pairs_raise_like_the_reference(hppfcl):  # CPU only: the function-matrix lookup
    tri = hppfcl.TriangleP([0, 0, 0], [1, 0, 0], [0, 1, 0])
    with pytest.raises(ValueError):  # no TriangleP in the distance matrix
        hppfcl.distance(tri, hppfcl.Transform3f(), hppfcl.Sphere(1.0), hppfcl.Transform3f(), hppfcl.DistanceRequest(),
                        hppfcl.DistanceResult())
    with pytest.raises(ValueError):
        hppfcl.ComputeDistance(hppfcl.Box(1, 1, 1), tri)
    hppfcl.ComputeCollision(hppfcl.Box(1, 1, 1), tri)  # collide() knows TriangleP
    req = hppfcl.CollisionRequest()
    req.num_max_contacts = 0
    with pytest.raises(ValueError):  # src/collision.cpp:82-85
        hppfcl.collide(hppfcl.Sphere(1), hppfcl.Transform3f(), hppfcl.Sphere(1), hppfcl.Transform3f(), req, hppfcl.CollisionResult())


def tetahedron(hppfcl):  # collision.py:8-20
    pts = hppfcl.StdVec_Vec3f()
    pts.append(np.array((0, 0, 0)))
    pts.append(np.array((0, 1, 0)))
    pts.append(np.array((1, 0, 0)))
    pts.append(np.array((0, 0, 1)))
    tri = hppfcl.StdVec_Triangle()
    tri.append(hppfcl.Triangle(0, 1, 2))
    tri.append(hppfcl.Triangle(0, 1, 3))
    tri.append(hppfcl.Triangle(0, 2, 3))
    tri.append(hppfcl.Triangle(1, 2, 3))
    return hppfcl.Convex(pts, tri)


def test_convex_neighbors_from_facets(hppfcl):  # fillNeighbors, shape/details/convex.hxx:231-280 (CPU only)
    offs, ids = tetahedron(hppfcl).neighbors()
    assert offs.tolist() == [0, 3, 6, 9, 12]
    assert ids.reshape(4, 3).tolist() == [[1, 2, 3], [0, 2, 3], [0, 1, 3], [0, 1, 2]]
    assert hppfcl.Convex([(0, 0, 0), (1, 0, 0), (0, 1, 0), (0, 0, 1)]).neighbors() is None


@pytest.mark.gpu
def test_large_convex_with_facets_climbs_its_adjacency(hppfcl):
    """A 700-vertex Convex given with its facets (Qhull) answers like the same points without facets (scanned)."""
    from scipy.spatial import ConvexHull
    rng = np.random.default_rng(5)
    d = rng.normal(size=(700, 3))
    pts = d / np.linalg.norm(d, axis=1, keepdims=True) * (0.8, 0.5, 0.3)
    tris = hppfcl.StdVec_Triangle()
    for a, b, c in ConvexHull(pts).simplices:
        tris.append(hppfcl.Triangle(int(a), int(b), int(c)))
    facetted, plain = hppfcl.Convex(pts, tris), hppfcl.Convex(pts)
    offs, ids = facetted.neighbors()
    assert len(ids) == 2 * (700 + len(tris) - 2)  # 2E, E = V + F - 2
    box = hppfcl.Box(0.4, 0.3, 0.2)
    for k in range(12):
        tf = hppfcl.Transform3f(np.eye(3), np.array([0.1 + 0.1 * k, 0.05 * k, 0.2]))
        r1, r2 = hppfcl.DistanceResult(), hppfcl.DistanceResult()
        d1 = hppfcl.distance(facetted, hppfcl.Transform3f(), box, tf, hppfcl.DistanceRequest(), r1)
        d2 = hppfcl.distance(plain, hppfcl.Transform3f(), box, tf, hppfcl.DistanceRequest(), r2)
        assert abs(d1 - d2) < 1e-6 and np.allclose(r1.getNearestPoint1(), r2.getNearestPoint1(), atol=1e-4)


@pytest.mark.gpu
def test_api_collision_and_distance(hppfcl):  # api.py:9-27
    capsule = hppfcl.Capsule(1.0, 2.0)
    M1 = hppfcl.Transform3f()
    M2 = hppfcl.Transform3f(np.eye(3), np.array([3, 0, 0]))
    req, res = hppfcl.CollisionRequest(), hppfcl.CollisionResult()
    assert not hppfcl.collide(capsule, M1, capsule, M2, req, res)
    dreq, dres = hppfcl.DistanceRequest(), hppfcl.DistanceResult()
    d = hppfcl.distance(capsule, M1, capsule, M2, dreq, dres)
    assert d > 0 and abs(d - 1.0) < 1e-9 and abs(dres.min_distance - 1.0) < 1e-9
    p1 = dres.getNearestPoint1()  # parallel capsules: any point of the segment x = 1, y = 0, |z| <= 1
    assert np.allclose(dres.normal, [1, 0, 0]) and np.allclose(p1[:2], [1, 0], atol=1e-6) and abs(p1[2]) <= 1 + 1e-9


@pytest.mark.gpu
def test_convex_halfspace(hppfcl):  # collision.py:24-46
    convex = tetahedron(hppfcl)
    halfspace = hppfcl.Halfspace(np.array((0, 0, 1)), 0)
    req, res = hppfcl.CollisionRequest(), hppfcl.CollisionResult()
    M1 = hppfcl.Transform3f()
    M2 = hppfcl.Transform3f(np.eye(3), np.array([0, 0, -0.1]))
    res.clear()
    hppfcl.collide(convex, M1, halfspace, M2, req, res)
    assert not hppfcl.collide(convex, M1, halfspace, M2, req, res)
    M2 = hppfcl.Transform3f(np.eye(3), np.array([0, 0, 0.1]))
    res.clear()
    assert hppfcl.collide(convex, M1, halfspace, M2, req, res)
    M2 = hppfcl.Transform3f(np.eye(3), np.array([0, 0, 2]))
    res.clear()
    assert hppfcl.collide(convex, M1, halfspace, M2, req, res)
    c = res.getContact(0)
    assert c.o1 is convex and c.o2 is halfspace and c.penetration_depth < 0


@pytest.mark.gpu
def test_collision_manager(hppfcl):  # collision_manager.py
    fcl = hppfcl
    sphere = fcl.Sphere(0.5)
    sphere_obj = fcl.CollisionObject(sphere)
    M_sphere = fcl.Transform3f.Identity()
    M_sphere.setTranslation(np.array([-0.6, 0.0, 0.0]))
    sphere_obj.setTransform(M_sphere)
    box = fcl.Box(np.array([0.5, 0.5, 0.5]))
    box_obj = fcl.CollisionObject(box)
    M_box = fcl.Transform3f.Identity()
    M_box.setTranslation(np.array([-0.6, 0.0, 0.0]))
    box_obj.setTransform(M_box)
    collision_manager = fcl.DynamicAABBTreeCollisionManager()
    collision_manager.registerObject(sphere_obj)
    collision_manager.registerObject(box_obj)
    assert collision_manager.size() == 2
    collision_manager.setup()
    callback = fcl.CollisionCallBackDefault()
    collision_manager.collide(sphere_obj, callback)
    assert callback.data.result.numContacts() == 1
    # collector + one batched narrow-phase call
    collect = fcl.CollisionCallBackCollect(100)
    collision_manager.collide(collect)
    assert collect.numCollisionPairs() == 1
    res = fcl.collide_pairs(collect.getCollisionPairs(), fcl.CollisionRequest())
    assert len(res) == 1 and res[0].isCollision()


@pytest.mark.gpu
def test_mesh_and_warm_start(hppfcl, pkg):
    v, t = pkg.bvh_builder.uv_sphere(8, 8, 1.0)
    m = hppfcl.BVHModelOBBRSS()
    m.beginModel(len(t), len(v))
    m.addSubModel(v, t)
    assert m.endModel() == 0 and m.getNodeType() == hppfcl.NODE_TYPE.BV_OBBRSS
    req, res = hppfcl.CollisionRequest(), hppfcl.CollisionResult()
    assert hppfcl.collide(m, hppfcl.Transform3f(), hppfcl.Box(1, 1, 1), hppfcl.Transform3f(np.eye(3), [1.2, 0, 0]), req, res) == 1
    assert res.getContact(0).b1 >= 0 and res.getContact(0).b2 == -1
    dreq, dres = hppfcl.DistanceRequest(), hppfcl.DistanceResult()
    d = hppfcl.distance(m, hppfcl.Transform3f(), m, hppfcl.Transform3f(np.eye(3), [3.0, 0, 0]), dreq, dres)
    assert 0.99 < d < 1.1 and dres.b1 >= 0 and dres.b2 >= 0
    # QueryRequest::updateGuess: the cached guess comes back into the request
    dreq.gjk_initial_guess = hppfcl.GJKInitialGuess.CachedGuess
    dres.clear()
    e1, e2 = hppfcl.Ellipsoid(0.3, 0.5, 0.8), hppfcl.Ellipsoid(0.6, 0.2, 0.4)
    d1 = hppfcl.distance(e1, hppfcl.Transform3f(), e2, hppfcl.Transform3f(np.eye(3), [2, 0.3, 0.1]), dreq, dres)
    assert not np.array_equal(dreq.cached_gjk_guess, [1, 0, 0])
    dres.clear()
    d2 = hppfcl.distance(e1, hppfcl.Transform3f(), e2, hppfcl.Transform3f(np.eye(3), [2, 0.3, 0.1]), dreq, dres)
    assert abs(d1 - d2) < 1e-6
