// Reads like the reference's own tests, against include/hppfcl_amd_compat.hpp:
//   test/capsule_box_1.cpp:51-116, test/box_box_distance.cpp:62-110, test/geometric_shapes.cpp:238-333 (subset),
//   src/collision.cpp:82-85 (num_max_contacts == 0 throws).
// Exit code 0 = all checks passed; 3 = no GPU (the shim has no CPU fallback).
#include <chrono>
#include <cmath>
#include <cstring>
#include <memory>
#include <vector>
#include <cstdio>
#include <stdexcept>

#include "hppfcl_amd_compat.hpp"

static int failures = 0;
#define CHECK(c) do { if (!(c)) { std::printf("FAILED %s:%d: %s\n", __FILE__, __LINE__, #c); ++failures; } } while (0)
#define CHECK_CLOSE(x, ref, pct) CHECK(std::fabs((x) - (ref)) <= std::fabs(ref) * (pct) / 100.0)
#define CHECK_CLOSE_TO_0(x, eps) CHECK(std::fabs(x) < (eps))

#define STAGE(name) std::fprintf(stderr, "[stage] %s\n", name)
int main() {
  using namespace hpp::fcl;
  std::setvbuf(stdout, nullptr, _IONBF, 0);
  if (hfcl_device_count() < 1) {
    Sphere s(1.0);
    try {
      DistanceRequest rq; DistanceResult rs;
      distance(&s, Transform3f(), &s, Transform3f(Vec3f(3, 0, 0)), rq, rs);
    } catch (const std::runtime_error& e) {
      std::printf("no GPU: %s\n", e.what());
      return 3;
    }
    return 4;
  }
  {  // distance_capsule_box (capsule_box_1.cpp)
    Capsule capsule(2., 4.);
    Box box(1., 2., 4.);
    DistanceRequest distanceRequest(true, true, 0, 0);
    DistanceResult distanceResult;
    Transform3f tf1(Vec3f(3., 0, 0)), tf2;
    distance(&capsule, tf1, &box, tf2, distanceRequest, distanceResult);
    Vec3f o1 = distanceResult.nearest_points[0], o2 = distanceResult.nearest_points[1];
    CHECK_CLOSE(distanceResult.min_distance, 0.5, 1e-1);
    CHECK_CLOSE(o1[0], 1.0, 1e-1); CHECK_CLOSE_TO_0(o1[1], 1e-1);
    CHECK_CLOSE(o2[0], 0.5, 1e-1); CHECK_CLOSE_TO_0(o2[1], 1e-1);
    tf1 = Transform3f(Vec3f(0., 0., 8.));
    distanceResult.clear();
    distance(&capsule, tf1, &box, tf2, distanceRequest, distanceResult);
    o1 = distanceResult.nearest_points[0]; o2 = distanceResult.nearest_points[1];
    CHECK_CLOSE(distanceResult.min_distance, 2.0, 1e-1);
    CHECK_CLOSE(o1[2], 4.0, 1e-1); CHECK_CLOSE(o2[2], 2.0, 1e-1);
    tf1.setTranslation(Vec3f(-10., 0., 0.));
    tf1.setQuatRotation(makeQuat(std::sqrt(2) / 2, 0, std::sqrt(2) / 2, 0));
    distanceResult.clear();
    distance(&capsule, tf1, &box, tf2, distanceRequest, distanceResult);
    o1 = distanceResult.nearest_points[0]; o2 = distanceResult.nearest_points[1];
    CHECK_CLOSE(distanceResult.min_distance, 5.5, 1e-1);
    CHECK_CLOSE(o1[0], -6, 1e-2); CHECK_CLOSE(o2[0], -0.5, 1e-2);
  }
  {  // distance_box_box_1 (box_box_distance.cpp:62-103)
    Box s1(6, 10, 2), s2(2, 2, 2);
    DistanceRequest rq(true, true, 0, 0);
    DistanceResult rs;
    distance(&s1, Transform3f(), &s2, Transform3f(Vec3f(25, 20, 5)), rq, rs);
    CHECK_CLOSE(rs.min_distance, std::sqrt(21. * 21 + 14 * 14 + 3 * 3), 1e-4);
    CHECK_CLOSE(rs.nearest_points[0][0], 3, 1e-6); CHECK_CLOSE(rs.nearest_points[0][1], 5, 1e-6);
    CHECK_CLOSE(rs.nearest_points[1][0], 24, 1e-6); CHECK_CLOSE(rs.nearest_points[1][2], 4, 1e-6);
  }
  {  // collide_spheresphere (geometric_shapes.cpp:238-333, subset) + Contact fields
    Sphere s1(20), s2(10);
    CollisionRequest rq; CollisionResult rs;
    CHECK(collide(&s1, Transform3f(), &s2, Transform3f(Vec3f(40, 0, 0)), rq, rs) == 0);
    CHECK(!rs.isCollision() && std::fabs(rs.distance_lower_bound - 10) < 1e-9);
    rs.clear();
    CHECK(collide(&s1, Transform3f(), &s2, Transform3f(Vec3f(29.9, 0, 0)), rq, rs) == 1);
    const Contact& c = rs.getContact(0);
    CHECK(std::fabs(c.normal[0] - 1) < 1e-12 && std::fabs(c.penetration_depth + 0.1) < 1e-9);
    CHECK(c.o1 == &s1 && c.o2 == &s2 && c.b1 == Contact::NONE);
    CHECK(std::fabs(c.pos[0] - (c.nearest_points[0][0] + c.nearest_points[1][0]) / 2) < 1e-12);
    rq.num_max_contacts = 0;
    bool threw = false;
    try { rs.clear(); collide(&s1, Transform3f(), &s2, Transform3f(), rq, rs); } catch (const std::invalid_argument&) { threw = true; }
    CHECK(threw);
    CollisionRequest rq2; rq2.security_margin = -std::numeric_limits<double>::infinity();
    CHECK(collide(&s1, Transform3f(), &s2, Transform3f(), rq2, rs) == 0 && !rs.isCollision());
  }
  {  // batched hand-off (CollisionCallBackCollect style) on convex hulls + warm start round trip
    auto pts = std::make_shared<std::vector<Vec3f>>();
    for (int i = 0; i < 8; ++i) pts->push_back(Vec3f((i & 1) ? 1 : -1, (i & 2) ? 1 : -1, (i & 4) ? 1 : -1));
    ConvexBase cube(pts);
    Box box(2, 2, 2);
    amd::BatchQueries batch;
    const uint32_t a = batch.add(&cube), b = batch.add(&box);
    std::vector<std::pair<uint32_t, uint32_t>> pairs;
    std::vector<Transform3f> tf1, tf2;
    for (int k = 0; k < 100; ++k) { pairs.push_back({a, b}); tf1.push_back(Transform3f()); tf2.push_back(Transform3f(Vec3f(1.5 + 0.02 * k, 0.1, 0))); }
    DistanceRequest rq; std::vector<DistanceResult> res;
    batch.distance(pairs, tf1, tf2, rq, res);
    for (int k = 0; k < 100; ++k) CHECK(std::fabs(res[k].min_distance - (1.5 + 0.02 * k - 2.0)) < 1e-6);
  }
  {  // the same hand-off over two replicas of the library (hfcl_multi_*; both on device 0 here): a 100 001-pair list cut into two shards
     // equals the single-library batch record for record (the collector that feeds it: default_broadphase_callbacks.cpp:93-123)
    STAGE("multi-device batch");
    auto pts = std::make_shared<std::vector<Vec3f>>();
    for (int i = 0; i < 8; ++i) pts->push_back(Vec3f((i & 1) ? 1 : -1, (i & 2) ? 1 : -1, (i & 4) ? 1 : -1));
    ConvexBase cube(pts);
    Box box(2, 2, 2);
    Capsule cap(0.4, 1.2);
    Sphere ball(0.7);
    amd::BatchQueries one, two(std::vector<int>{0, 0});
    CHECK(two.numDevices() == 2);
    const CollisionGeometry* geoms[4] = {&cube, &box, &cap, &ball};
    uint32_t id1[4], id2[4];
    for (int k = 0; k < 4; ++k) { id1[k] = one.add(geoms[k]); id2[k] = two.add(geoms[k]); CHECK(id1[k] == id2[k]); }
    const size_t n = 100001;
    std::vector<std::pair<uint32_t, uint32_t>> pairs(n);
    std::vector<Transform3f> tf1(n), tf2(n);
    uint64_t rng = 12345;
    auto uni = [&]() { rng = rng * 6364136223846793005ull + 1442695040888963407ull; return double(rng >> 11) / double(1ull << 53); };
    for (size_t k = 0; k < n; ++k) {
      pairs[k] = {id1[k % 4], id1[(k / 4 + k) % 4]};
      auto unit = [](double w, double x, double y, double z) { const double s = 1.0 / std::sqrt(w * w + x * x + y * y + z * z); return makeQuat(w * s, x * s, y * s, z * s); };
      tf1[k] = Transform3f(unit(1, 0.3 * uni(), 0.2 * uni(), 0.1 * uni()), Vec3f(0.3 * uni(), 0.3 * uni(), 0.3 * uni()));
      tf2[k] = Transform3f(unit(1, 0.5 * uni(), 0.4 * uni(), 0.3 * uni()), Vec3f(3.0 * uni() - 0.5, 0.8 * uni(), 0.8 * uni()));
    }
    DistanceRequest dq; std::vector<DistanceResult> r1, r2;
    one.distance(pairs, tf1, tf2, dq, r1);
    two.distance(pairs, tf1, tf2, dq, r2);
    CHECK(r1.size() == n && r2.size() == n);
    size_t differ = 0, penetrating = 0;
    for (size_t k = 0; k < n; ++k) {
      differ += std::memcmp(&one.records()[k], &two.records()[k], sizeof(hfcl_result)) != 0;
      penetrating += r1[k].min_distance <= 0;
    }
    CHECK(differ == 0);
    CHECK(penetrating > n / 20 && penetrating < n - n / 20);
    CollisionRequest cq; std::vector<CollisionResult> c1, c2;
    one.collide(pairs, tf1, tf2, cq, c1);
    two.collide(pairs, tf1, tf2, cq, c2);
    differ = 0;
    for (size_t k = 0; k < n; ++k) differ += std::memcmp(&one.records()[k], &two.records()[k], sizeof(hfcl_result)) != 0 || c1[k].numContacts() != c2[k].numContacts();
    CHECK(differ == 0);
  }
  {  // BVHModel<OBBRSS>: two box meshes as in test/collision.cpp / geometric_shape_to_BVH_model.h (12 triangles each)
    auto make_box_mesh = [](BVHModel<OBBRSS>& m, double hx, double hy, double hz) {
      std::vector<Vec3f> ps;
      for (int i = 0; i < 8; ++i) ps.push_back(Vec3f((i & 1) ? hx : -hx, (i & 2) ? hy : -hy, (i & 4) ? hz : -hz));
      const int q[6][4] = {{0, 2, 3, 1}, {4, 5, 7, 6}, {0, 1, 5, 4}, {2, 6, 7, 3}, {0, 4, 6, 2}, {1, 3, 7, 5}};
      std::vector<Triangle> ts;
      for (auto& f : q) { ts.emplace_back(f[0], f[1], f[2]); ts.emplace_back(f[0], f[2], f[3]); }
      CHECK(m.beginModel() == BVH_OK);
      CHECK(m.addSubModel(ps, ts) == BVH_OK);
      CHECK(m.endModel() == BVH_OK);
    };
    STAGE("mesh: build");
    BVHModel<OBBRSS> m1, m2;
    make_box_mesh(m1, 1, 1, 1);
    make_box_mesh(m2, 0.5, 0.5, 0.5);
    CHECK(m1.getNumBVs() == 23 && m1.num_tris == 12 && m1.getNodeType() == BV_OBBRSS);
    STAGE("mesh: collide separated");
    CollisionRequest rq; CollisionResult rs;
    CHECK(collide(&m1, Transform3f(), &m2, Transform3f(Vec3f(3, 0, 0)), rq, rs) == 0);
    CHECK(rs.distance_lower_bound > 0.0 && rs.distance_lower_bound <= 1.5 + 1e-9);  // OBB bound of a 1.5 gap
    rs.clear();
    CHECK(collide(&m1, Transform3f(), &m2, Transform3f(Vec3f(1.2, 0.1, 0.2)), rq, rs) == 1);
    CHECK(rs.getContact(0).b1 >= 0 && rs.getContact(0).b1 < 12 && rs.getContact(0).b2 >= 0 && rs.getContact(0).o1 == &m1);
    rs.clear();
    STAGE("mesh: all contacts");
    CollisionRequest all; all.num_max_contacts = 1000;
    const std::size_t nall = collide(&m1, Transform3f(), &m2, Transform3f(Vec3f(1.2, 0.1, 0.2)), all, rs);
    CHECK(nall > 1 && nall == rs.numContacts());
    STAGE("mesh: distance");
    DistanceRequest dq; DistanceResult dr;
    const double d = distance(&m1, Transform3f(), &m2, Transform3f(Vec3f(3, 0, 0)), dq, dr);
    CHECK(std::fabs(d - 1.5) < 1e-9 && dr.b1 >= 0 && dr.b2 >= 0);
  }
  {  // collide_halfspacesphere / collide_planebox (geometric_shapes.cpp:1275-1300, 1567-1590)
    STAGE("halfspace / plane");
    Sphere s(10);
    Halfspace hs(Vec3f(1, 0, 0), 0);
    CollisionRequest rq; CollisionResult rs;
    CHECK(collide(&s, Transform3f(), &hs, Transform3f(Vec3f(5, 0, 0)), rq, rs) == 1);
    CHECK(std::fabs(rs.getContact(0).penetration_depth + 15) < 1e-9 && std::fabs(rs.getContact(0).pos[0] + 2.5) < 1e-9);
    CHECK(std::fabs(rs.getContact(0).normal[0] + 1) < 1e-12);
    rs.clear();
    CHECK(collide(&s, Transform3f(), &hs, Transform3f(Vec3f(-10.1, 0, 0)), rq, rs) == 0);
    Box b(5, 10, 20);
    Plane pl(Vec3f(2, 0, 0), 0);  // normalised by the constructor
    rs.clear();
    CHECK(collide(&b, Transform3f(), &pl, Transform3f(Vec3f(1.25, 0, 0)), rq, rs) == 1);
    CHECK(std::fabs(rs.getContact(0).penetration_depth + 1.25) < 1e-9 && std::fabs(rs.getContact(0).normal[0] - 1) < 1e-12);
    Cylinder cy(5, 10);
    DistanceRequest dq; DistanceResult dr;
    CHECK(std::fabs(distance(&cy, Transform3f(), &cy, Transform3f(Vec3f(40, 0, 0)), dq, dr) - 30) < 1e-3);
  }
  {  // collide_spheretriangle (geometric_shapes.cpp:844-880) + a TriangleP against a GJK solid
    STAGE("TriangleP");
    Sphere s(10);
    TriangleP tri(Vec3f(20, 0, 0), Vec3f(-20, 0, 0), Vec3f(0, 20, 0));
    CollisionRequest rq; CollisionResult rs;
    CHECK(collide(&s, Transform3f(), &tri, Transform3f(Vec3f(0, 0, 0.001)), rq, rs) == 1);
    CHECK(std::fabs(rs.getContact(0).normal[2] - 1) < 1e-9);
    rs.clear();
    CHECK(collide(&tri, Transform3f(Vec3f(0, 0, -0.001)), &s, Transform3f(), rq, rs) == 1);
    CHECK(std::fabs(rs.getContact(0).normal[2] - 1) < 1e-9);  // from the triangle (below) towards the sphere
    Box b(2, 2, 2);
    rs.clear();
    CHECK(collide(&b, Transform3f(), &tri, Transform3f(Vec3f(0, 0, 3)), rq, rs) == 0);
    CHECK(std::fabs(rs.distance_lower_bound - 2) < 1e-6 && std::fabs(rs.normal[2] - 1) < 1e-6);
    CHECK(std::fabs(rs.nearest_points[1][2] - 3) < 1e-6);
    rs.clear();
    CHECK(collide(&tri, Transform3f(Vec3f(0, 0, 3)), &b, Transform3f(), rq, rs) == 0);
    CHECK(std::fabs(rs.distance_lower_bound - 2) < 1e-6 && std::fabs(rs.normal[2] + 1) < 1e-6);
    CHECK(std::fabs(rs.nearest_points[0][2] - 3) < 1e-6);
    rs.clear();
    CHECK(collide(&b, Transform3f(), &tri, Transform3f(Vec3f(0, 0, 0.5)), rq, rs) == 1);
    CHECK(std::fabs(rs.getContact(0).penetration_depth + 0.5) < 1e-6);
    {  // a TriangleP whose corners change between two calls (same address) must not hit the shape cache
      TriangleP moving(Vec3f(-5, -5, 3), Vec3f(5, -5, 3), Vec3f(0, 5, 3));
      rs.clear();
      CHECK(collide(&b, Transform3f(), &moving, Transform3f(), rq, rs) == 0);
      CHECK(std::fabs(rs.distance_lower_bound - 2) < 1e-6);
      moving.a[2] = moving.b[2] = moving.c[2] = 0.5;  // now cuts the box
      rs.clear();
      CHECK(collide(&b, Transform3f(), &moving, Transform3f(), rq, rs) == 1);
      CHECK(std::fabs(rs.getContact(0).penetration_depth + 0.5) < 1e-6);
    }
    // the reference's distance matrix has no TriangleP entries: distance() throws (src/distance.cpp:69-75)
    bool threw_d = false;
    DistanceRequest dq; DistanceResult dr;
    try { distance(&b, Transform3f(), &tri, Transform3f(Vec3f(0, 0, 3)), dq, dr); } catch (const std::invalid_argument&) { threw_d = true; }
    CHECK(threw_d);
    bool threw_f = false;
    try { ComputeDistance bad(&b, &tri); } catch (const std::invalid_argument&) { threw_f = true; }
    CHECK(threw_f);
  }
  {  // functors (collision.h:79-117, distance.h:74-112)
    STAGE("ComputeCollision / ComputeDistance");
    Box b(2, 2, 2);
    Capsule c(0.5, 2);
    ComputeCollision cc(&b, &c);
    ComputeDistance cd(&b, &c);
    CollisionRequest rq; CollisionResult rs, rs2;
    DistanceRequest dq; DistanceResult dr, dr2;
    const Transform3f tf2(Vec3f(1.2, 0.1, 0.3));
    CHECK(cc(Transform3f(), tf2, rq, rs) == collide(&b, Transform3f(), &c, tf2, rq, rs2));
    CHECK(rs.numContacts() == 1 && rs.getContact(0).penetration_depth == rs2.getContact(0).penetration_depth);
    const Transform3f far(Vec3f(4, 0, 0));
    CHECK(cd(Transform3f(), far, dq, dr) == distance(&b, Transform3f(), &c, far, dq, dr2));
    CHECK(std::fabs(dr.min_distance - 2.5) < 1e-6);
    BVHModel<OBBRSS> empty_model;
    TriangleP tri(Vec3f(0, 0, 0), Vec3f(1, 0, 0), Vec3f(0, 1, 0));
    bool threw = false;
    try { ComputeCollision bad(&tri, &empty_model); } catch (const std::invalid_argument&) { threw = true; }
    CHECK(threw);  // (TriangleP, BVHModel) has no entry in the reference's matrices either
  }
  {  // broadphase hand-off: manager + CollisionCallBackCollect, then one device batch (test/broadphase.cpp style)
    STAGE("broadphase");
    std::vector<std::shared_ptr<CollisionGeometry>> geoms;
    std::vector<std::unique_ptr<CollisionObject>> objects;
    unsigned seed = 12345;
    auto rnd = [&seed]() { seed = seed * 1664525u + 1013904223u; return (seed >> 8) / double(1 << 24); };
    for (int i = 0; i < 300; ++i) {
      std::shared_ptr<CollisionGeometry> g;
      if (i % 3 == 0) g = std::make_shared<Box>(0.2 + rnd(), 0.2 + rnd(), 0.2 + rnd());
      else if (i % 3 == 1) g = std::make_shared<Sphere>(0.2 + 0.5 * rnd());
      else g = std::make_shared<Capsule>(0.1 + 0.4 * rnd(), 0.2 + rnd());
      geoms.push_back(g);
      objects.emplace_back(new CollisionObject(g, Transform3f(Vec3f(6 * rnd(), 6 * rnd(), 6 * rnd()))));
    }
    DynamicAABBTreeCollisionManager manager;
    for (auto& o : objects) manager.registerObject(o.get());
    manager.setup();
    CollisionCallBackCollect collect(10000);
    manager.collide(&collect);
    size_t brute = 0;  // same candidate set as the O(n^2) AABB test
    for (size_t i = 0; i < objects.size(); ++i)
      for (size_t j = i + 1; j < objects.size(); ++j) brute += objects[i]->getAABB().overlap(objects[j]->getAABB());
    CHECK(brute > 50 && collect.numCollisionPairs() == brute);
    {  // the default callbacks (test/broadphase.cpp style): stop at the first collision / global minimum distance
      std::vector<std::unique_ptr<CollisionObject>> few;
      DynamicAABBTreeCollisionManager m2;
      for (int i = 0; i < 12; ++i) {
        few.emplace_back(new CollisionObject(std::make_shared<Sphere>(0.4), Transform3f(Vec3f(1.5 * i, 0.1 * i, 0))));
        m2.registerObject(few.back().get());
      }
      m2.setup();
      CollisionCallBackDefault cb;
      m2.collide(&cb);
      CHECK(!cb.data.result.isCollision());  // centres 1.5+ apart, radii 0.4
      DistanceCallBackDefault db;
      m2.distance(&db);
      const double expect = std::sqrt(1.5 * 1.5 + 0.1 * 0.1) - 0.8;
      CHECK(std::fabs(db.data.result.min_distance - expect) < 1e-9);
      few[3]->setTransform(Transform3f(Vec3f(1.5 * 2 + 0.5, 0.2, 0)));  // now sphere 3 cuts sphere 2
      m2.update();
      m2.collide(&cb);
      CHECK(cb.data.result.isCollision() && cb.data.done);
      m2.distance(&db);
      CHECK(db.data.result.min_distance < 0);
    }
    {  // the manager evaluates the DEFAULT callbacks in device batches; a subclass (a user callback) still sees the pairs one
       // by one through collide(): both must leave the same CollisionData / DistanceData behind, and the batched form
       // costs a fraction of a microsecond per culled pair where the per-pair form pays a launch round trip each
      struct OneByOne : CollisionCallBackDefault {  // same function, but not recognised as the default callback
        size_t calls = 0;
        bool collide(CollisionObject* a, CollisionObject* b) override { ++calls; return defaultCollisionFunction(a, b, &data); }
      };
      struct OneByOneD : DistanceCallBackDefault {
        size_t calls = 0;
        bool distance(CollisionObject* a, CollisionObject* b, FCL_REAL& d) override { ++calls; return defaultDistanceFunction(a, b, &data, d); }
      };
      auto same_contact = [](const Contact& x, const Contact& y) {
        return x.o1 == y.o1 && x.o2 == y.o2 && x.b1 == y.b1 && x.b2 == y.b2 && x.penetration_depth == y.penetration_depth &&
               (x.normal - y.normal).norm() == 0 && (x.pos - y.pos).norm() == 0;
      };
      for (size_t max_contacts : {size_t(1), size_t(100000)}) {
        CollisionCallBackDefault batched;
        OneByOne single;
        batched.data.request.num_max_contacts = single.data.request.num_max_contacts = max_contacts;
        // init() clears the data (request included in the reference too? no: CollisionData::clear keeps the request)
        manager.collide(&batched);  // (first use: the library allocates its small-batch blocks)
        auto t0 = std::chrono::steady_clock::now();
        manager.collide(&batched);
        auto t1 = std::chrono::steady_clock::now();
        manager.collide(&single);
        auto t2 = std::chrono::steady_clock::now();
        CHECK(batched.data.done == single.data.done);
        CHECK(batched.data.result.numContacts() == single.data.result.numContacts());
        CHECK(batched.data.result.distance_lower_bound == single.data.result.distance_lower_bound);
        for (size_t k = 0; k < single.data.result.numContacts(); ++k)
          CHECK(same_contact(batched.data.result.getContact(k), single.data.result.getContact(k)));
        if (max_contacts > 1) {
          CHECK(single.calls == brute && single.data.result.numContacts() > 20);
          const double us_b = std::chrono::duration<double, std::micro>(t1 - t0).count() / double(brute);
          const double us_s = std::chrono::duration<double, std::micro>(t2 - t1).count() / double(brute);
          printf("default collision callback over %zu culled pairs: %.2f us per pair in device batches, %.1f us per pair one by one\n", brute, us_b, us_s);
        } else {
          CHECK(single.data.result.numContacts() == 1 && single.calls < brute);
        }
      }
      DistanceCallBackDefault dbatched;
      OneByOneD dsingle;
      manager.distance(&dbatched);
      manager.distance(&dsingle);
      CHECK(dbatched.data.result.min_distance == dsingle.data.result.min_distance);
      CHECK(dbatched.data.result.o1 == dsingle.data.result.o1 && dbatched.data.result.o2 == dsingle.data.result.o2);
      std::vector<std::unique_ptr<CollisionObject>> apart;  // a scene without contact: the global minimum distance
      DynamicAABBTreeCollisionManager m3;
      for (int i = 0; i < 400; ++i) {
        apart.emplace_back(new CollisionObject(geoms[size_t(i) % geoms.size()], Transform3f(Vec3f(4.0 * (i % 8), 4.0 * ((i / 8) % 8), 4.0 * (i / 64) + 0.37 * rnd()))));
        m3.registerObject(apart.back().get());
      }
      m3.setup();
      DistanceCallBackDefault d3;
      OneByOneD d3s;
      auto t0 = std::chrono::steady_clock::now();
      m3.distance(&d3);
      auto t1 = std::chrono::steady_clock::now();
      m3.distance(&d3s);
      auto t2 = std::chrono::steady_clock::now();
      CHECK(d3.data.result.min_distance > 0 && d3.data.result.min_distance == d3s.data.result.min_distance);
      CHECK(d3.data.result.o1 == d3s.data.result.o1 && d3.data.result.o2 == d3s.data.result.o2);
      CHECK((d3.data.result.nearest_points[0] - d3s.data.result.nearest_points[0]).norm() == 0);
      printf("default distance callback, 400 objects: %.0f us in device batches, %.0f us one by one (%zu narrow-phase calls)\n",
             std::chrono::duration<double, std::micro>(t1 - t0).count(), std::chrono::duration<double, std::micro>(t2 - t1).count(), d3s.calls);
    }
    CollisionRequest rq; std::vector<CollisionResult> res;
    amd::collide(collect.getCollisionPairs(), rq, res);
    CHECK(res.size() == brute);
    size_t hits = 0;
    for (size_t k = 0; k < res.size(); ++k) {
      CollisionResult one;  // each batched result equals the single-pair call
      const auto& pr = collect.getCollisionPairs()[k];
      collide(pr.first->collisionGeometryPtr(), pr.first->getTransform(), pr.second->collisionGeometryPtr(), pr.second->getTransform(), rq, one);
      CHECK(one.isCollision() == res[k].isCollision());
      hits += res[k].isCollision();
      if (k > 40) break;
    }
    (void)hits;
  }
  {  // temporaries: every call registers a new geometry; the context follows with new shape tables and can be pruned
    STAGE("context growth / reset");
    CollisionRequest rq;
    const size_t before = amd::default_context().numGeometries();
    for (int i = 0; i < 20; ++i) {
      Sphere a(0.5 + 0.01 * i), b(0.4);
      CollisionResult rs;
      CHECK(collide(&a, Transform3f(), &b, Transform3f(Vec3f(0.85 + 0.01 * i, 0, 0)), rq, rs) == 1);
      CHECK(std::fabs(rs.getContact(0).penetration_depth + 0.05) < 1e-9);
    }
    CHECK(amd::default_context().numGeometries() >= before);
    amd::default_context().reset();
    CHECK(amd::default_context().numGeometries() == 0);
    Box bx(1, 1, 1);
    CollisionResult rs;
    CHECK(collide(&bx, Transform3f(), &bx, Transform3f(Vec3f(0.5, 0, 0)), rq, rs) == 1);
  }
  {  // Convex<Triangle>: the facets give ConvexBase::neighbors; a 642-vertex hull climbs them on the device, the same
     // points without facets are scanned -- same answers
    STAGE("Convex<Triangle> adjacency (hill-climbing support)");
    std::vector<Vec3f> pts = {Vec3f(-1, 1.618033988749895, 0), Vec3f(1, 1.618033988749895, 0), Vec3f(-1, -1.618033988749895, 0), Vec3f(1, -1.618033988749895, 0),
                              Vec3f(0, -1, 1.618033988749895), Vec3f(0, 1, 1.618033988749895), Vec3f(0, -1, -1.618033988749895), Vec3f(0, 1, -1.618033988749895),
                              Vec3f(1.618033988749895, 0, -1), Vec3f(1.618033988749895, 0, 1), Vec3f(-1.618033988749895, 0, -1), Vec3f(-1.618033988749895, 0, 1)};
    std::vector<Triangle> tris = {{0, 11, 5}, {0, 5, 1}, {0, 1, 7}, {0, 7, 10}, {0, 10, 11}, {1, 5, 9}, {5, 11, 4}, {11, 10, 2}, {10, 7, 6}, {7, 1, 8},
                                  {3, 9, 4}, {3, 4, 2}, {3, 2, 6}, {3, 6, 8}, {3, 8, 9}, {4, 9, 5}, {2, 4, 11}, {6, 2, 10}, {8, 6, 7}, {9, 8, 1}};
    for (Vec3f& p : pts) p = p / p.norm();
    for (int level = 0; level < 3; ++level) {  // 12 -> 42 -> 162 -> 642 vertices on the unit sphere (all extreme)
      std::map<std::pair<size_t, size_t>, size_t> mid;
      auto midpoint = [&](size_t a, size_t b) {
        const auto key = std::make_pair(std::min(a, b), std::max(a, b));
        auto it = mid.find(key);
        if (it != mid.end()) return it->second;
        Vec3f m = (pts[a] + pts[b]) * 0.5;
        pts.push_back(m / m.norm());
        return mid[key] = pts.size() - 1;
      };
      std::vector<Triangle> next;
      for (const Triangle& t : tris) {
        const size_t a = midpoint(t[0], t[1]), b = midpoint(t[1], t[2]), c = midpoint(t[2], t[0]);
        next.push_back(Triangle(t[0], a, c));
        next.push_back(Triangle(t[1], b, a));
        next.push_back(Triangle(t[2], c, b));
        next.push_back(Triangle(a, b, c));
      }
      tris.swap(next);
    }
    for (Vec3f& p : pts) p = Vec3f(0.9 * p[0], 0.6 * p[1], 0.4 * p[2]);
    CHECK(pts.size() == 642 && tris.size() == 1280);
    auto shared_pts = std::make_shared<std::vector<Vec3f>>(pts);
    Convex<Triangle> with_facets(shared_pts, unsigned(pts.size()), std::make_shared<std::vector<Triangle>>(tris), unsigned(tris.size()));
    ConvexBase points_only(shared_pts);
    CHECK(with_facets.neighbor_offsets.size() == 643 && with_facets.neighbor_ids.size() == 2 * (642 + 1280 - 2));  // 2E, E = V + F - 2
    CHECK(with_facets.neighbor_offsets[1] - with_facets.neighbor_offsets[0] == 5);                                 // an icosahedron corner
    Box other(0.5, 0.4, 0.3);
    DistanceRequest dq;
    int compared = 0;
    for (int i = 0; i < 60; ++i) {
      const double a = 0.37 * i, r = 0.2 + 0.02 * i;  // from deep penetration to well apart
      Transform3f tf(Vec3f(r * std::cos(a), r * std::sin(a), 0.3 * std::sin(1.7 * a)));
      DistanceResult r1, r2;
      const double d1 = distance(&with_facets, Transform3f(), &other, tf, dq, r1);
      const double d2 = distance(&points_only, Transform3f(), &other, tf, dq, r2);
      CHECK(std::fabs(d1 - d2) < 1e-6);
      CHECK((r1.nearest_points[0] - r2.nearest_points[0]).norm() < 1e-4);
      ++compared;
    }
    CHECK(compared == 60);
  }
  STAGE("done");
  std::printf("%s (%d failures)\n", failures ? "FAILED" : "ok", failures);
  return failures ? 1 : 0;
}
