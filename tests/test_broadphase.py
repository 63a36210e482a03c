"""Host broadphase pair producer against the oracle: world AABBs as
CollisionObject::computeAABB (collision_object.h:259-276), candidate set = all AABB-overlapping pairs
(what DynamicAABBTreeCollisionManager::collide hands to its callback).  The reference's own tests
(test/broadphase.cpp, broadphase_dynamic_AABB_tree.cpp) compare managers with brute force likewise."""
import numpy as np
import pytest


def _scene(pkg, n, pairs, seed=1):
    return pkg.workloads.cfg5_broadphase_scene(n_objects=n, target_pairs=pairs, seed=seed)


def test_world_aabbs_match_oracle_bitwise(pkg, oracle):
    b = _scene(pkg, 5000, 20000)
    sc = b.scene
    ref = oracle.world_aabbs(b.shapes, b.verts, sc["obj_shape"], sc["obj_tf"])
    assert sc["aabbs"].tobytes() == ref.tobytes()
    # identity rotations take the translate() branch
    tf = sc["obj_tf"].copy()
    tf[:100, :9] = np.eye(3).reshape(-1)
    got = pkg.engine.world_aabbs(b.lib, sc["obj_shape"], tf)
    assert got.tobytes() == oracle.world_aabbs(b.shapes, b.verts, sc["obj_shape"], tf).tobytes()
    # swept-sphere radius inflates the local box (geometric_shapes.cpp:147-151)
    L = pkg.geometry.ShapeLibrary()
    L.add_box(1, 2, 3, swept_sphere_radius=0.25)
    I = pkg.geometry.make_pose(R=np.eye(3)[None], T=np.zeros((1, 3)))
    assert np.allclose(pkg.engine.world_aabbs(L, [0], I), [[-0.75, -1.25, -1.75, 0.75, 1.25, 1.75]])


def test_world_aabb_contains_the_posed_shape(pkg):
    b = _scene(pkg, 2000, 5000, seed=3)
    sc = b.scene
    g = pkg.geometry
    rng = np.random.default_rng(0)
    dirs = rng.normal(size=(64, 3))
    shapes = b.shapes
    for i in range(0, 2000, 37):
        s = shapes[sc["obj_shape"][i]]
        if s["type"] != pkg.abi.GEOM_CONVEX:
            continue
        P = b.verts[s["vertex_offset"]:s["vertex_offset"] + s["num_points"]]
        R = sc["obj_tf"][i][:9].reshape(3, 3).T
        W = P @ R.T + sc["obj_tf"][i][9:]
        assert (W.min(0) >= sc["aabbs"][i, :3] - 1e-12).all() and (W.max(0) <= sc["aabbs"][i, 3:] + 1e-12).all()


@pytest.mark.parametrize("n,pairs,threads", [(3000, 20000, 1), (3000, 20000, 7), (6000, 3000, 0), (50, 400, 0)])
def test_pair_set_equals_brute_force(pkg, oracle, n, pairs, threads):
    b = _scene(pkg, n, pairs, seed=2)
    got = pkg.engine.broadphase_self_pairs(b.scene["aabbs"], threads)
    ref = oracle.bruteforce_pairs(b.scene["aabbs"])
    assert len(ref) > 0.3 * pairs
    assert np.array_equal(got, ref)  # same set AND the documented (i asc, j asc) order


def test_pairs_between_two_managers(pkg, oracle):
    a = _scene(pkg, 1500, 5000, seed=4).scene["aabbs"]
    b = _scene(pkg, 2500, 9000, seed=5).scene["aabbs"]
    b = b * 0.7  # shrink the second scene so both overlap in space
    got = pkg.engine.broadphase_pairs_between(a, b)
    ov = ~((a[:, None, :3] > b[None, :, 3:]).any(-1) | (a[:, None, 3:] < b[None, :, :3]).any(-1))
    ii, jj = np.nonzero(ov)
    assert len(ii) > 100
    assert np.array_equal(got, np.stack([ii, jj], 1).astype(np.uint32))


def test_degenerate_inputs(pkg):
    e = pkg.engine
    assert e.broadphase_self_pairs(np.zeros((0, 6))).shape == (0, 2)
    assert e.broadphase_self_pairs(np.zeros((1, 6))).shape == (0, 2)
    same = np.tile(np.array([[0, 0, 0, 1, 1, 1.0]]), (40, 1))  # all boxes identical: every pair, closed test
    assert len(e.broadphase_self_pairs(same)) == 40 * 39 // 2
    touching = np.array([[0, 0, 0, 1, 1, 1.0], [1, 0, 0, 2, 1, 1.0], [2.0000001, 0, 0, 3, 1, 1]])
    assert e.broadphase_self_pairs(touching).tolist() == [[0, 1]]  # AABB::overlap is inclusive (AABB.h:112-122)


def test_cfg5_scene_statistics(pkg):
    b = _scene(pkg, 20000, 100000)
    assert 0.5 * 100000 < len(b) < 2.0 * 100000
    kinds = b.shapes["type"][b.scene["obj_shape"]]
    frac = np.array([(kinds == k).mean() for k in np.unique(kinds)])
    assert len(frac) == 5 and (np.abs(frac - 0.2) < 0.02).all()


@pytest.mark.gpu
@pytest.mark.parametrize("n_objects,pairs", [(60000, 300000), (125000, 1250000)])
def test_gpu_narrowphase_on_broadphase_pairs(pkg, oracle, n_objects, pairs):
    """cfg5 end to end: host broadphase -> device collide() on the candidate pairs == oracle; the second case is BASELINE.json
    configs[4]'s per-GPU share (10M / 8 = 1.25M pairs from a scene of 125 000 objects), every record against the oracle."""
    import compare
    import os
    abi, wl = pkg.abi, pkg.workloads
    b = _scene(pkg, n_objects, int(1.15 * pairs))  # (the scene's side is chosen for ABOUT that many overlapping boxes)
    assert len(b) >= pairs, len(b)
    b = b.slice(0, pairs)
    req = wl.make_request(b, abi)
    lib = wl.make_library(pkg, b)
    got = lib.collide(b.s1, b.s2, b.tf1, b.tf2, req)
    ref = oracle.collide_batch(b.shapes, b.verts, b.s1, b.s2, b.tf1, b.tf2, req, n_threads=min(64, os.cpu_count() or 8))
    compare.check_parity(abi, got, ref, dist_tol=1e-6, point_tol=1e-5, flag_band=1e-9, name="cfg5-broadphase")
    assert 0.05 < (ref["num_contacts"] > 0).mean() < 0.9
    lib.close()


def test_plane_and_halfspace_objects(pkg, oracle):
    """Plane / Halfspace objects: unbounded local AABBs (computeBV<AABB, Halfspace|Plane>,
    geometric_shapes_utility.cpp:391-455; bounded only along an axis the normal is aligned with), posed by
    CollisionObject::computeAABB like any other; the candidate set is still the brute-force AABB-overlap set."""
    g = pkg.geometry
    rng = np.random.default_rng(6)
    L = g.ShapeLibrary()
    for s in rng.uniform(0.1, 0.6, (40, 3)):
        L.add_box(*map(float, s))
    flats = [L.add_halfspace([0, 0, 1], -1.0), L.add_halfspace([0, 0, -1], 0.5), L.add_plane([1, 0, 0], 0.3),
             L.add_plane([0, -2, 0], 0.4), L.add_halfspace([1, 1, 0], 0.0), L.add_plane([0.3, 0.2, 0.9], 0.1),
             L.add_halfspace([0, 1, 0], 0.2, swept_sphere_radius=0.05)]
    n = 600
    obj_shape = rng.integers(0, 40, n).astype(np.uint32)
    obj_shape[:: n // 20] = rng.choice(flats, len(obj_shape[:: n // 20]))
    q = rng.normal(size=(n, 4))
    tf = g.make_pose(quat=q / np.linalg.norm(q, axis=1, keepdims=True), T=rng.uniform(-3, 3, (n, 3)))
    tf[: n // 2, :9] = np.eye(3).reshape(-1)  # half of the objects unrotated: axis-aligned flats stay half-bounded
    got = pkg.engine.world_aabbs(L, obj_shape, tf)
    ref = oracle.world_aabbs(L.shapes_array(), L.vertices_array(), obj_shape, tf)
    assert not np.isnan(got).any()
    assert got.tobytes() == ref.tobytes()
    big = np.finfo(np.float64).max
    k = int(np.nonzero((obj_shape == flats[0]) & (np.arange(n) < n // 2))[0][0]) if ((obj_shape == flats[0]) & (np.arange(n) < n // 2)).any() else None
    if k is not None:  # z <= -1 shifted by T: bounded above in z only
        assert got[k, 5] == -1.0 + tf[k, 11] and got[k, 2] == -big and got[k, 0] == -big and got[k, 3] == big
    for threads in (1, 5):
        pairs = pkg.engine.broadphase_self_pairs(got, threads)
        assert np.array_equal(pairs, oracle.bruteforce_pairs(got))
    assert len(pairs) > n  # every unbounded flat meets nearly every object
