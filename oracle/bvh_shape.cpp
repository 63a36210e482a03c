// ORACLE -- TEST INFRASTRUCTURE ONLY.  Not part of the product path (see vec3.hpp).
//
// fp64 CPU restatement of collide() between a BVHModel<OBBRSS> and a convex shape:
//   BVHShapeCollider<OBBRSS,S>::oriented      src/collision_func_matrix.cpp:102-155 (negative margin throws :109-112)
//   initialize(MeshShapeCollisionTraversalNode<BV,S,0>)   include/hpp/fcl/internal/traversal_node_setup.h:378-404
//   computeBV<OBBRSS,S> (generic)             include/hpp/fcl/shape/geometric_shapes_utility.h:73-82
//   getBoundVertices                          src/shape/geometric_shapes_utility.cpp:46-240
//   fit<OBBRSS>(points) = OBB fitn + RSS fitn src/BVH/BV_fitter.cpp:131-143,202-216,393-396,455-469
//   getCovariance / getExtentAndCenter / getRadiusAndOriginAndRectangleSize (point clouds)
//                                              src/BVH/BVH_utility.cpp:183-259,264-482,485-527
//   MeshShapeCollisionTraversalNode            include/hpp/fcl/internal/traversal_node_bvh_shape.h:98-188
//   collisionRecurse (second node always leaf) src/traversal/traversal_recurse.cpp:44-85
//   operand swap for (shape, BVH)              src/collision.cpp:93-108, collision_data.h (swapObjects)
// PARITY UNPINNED by reference numbers (test/collision.cpp compares BV types against each other);
// checked against brute force over all triangles in tests/test_bvh_shape.py.
#include <algorithm>
#include <cmath>
#include <cstring>
#include <limits>
#include <vector>
#include "bvh.hpp"

namespace orc {

// cyclic Jacobi of oracle/bvh_build.cpp (internal/tools.h:103-202)
void jacobi_eigen3(const double M[3][3], double dout[3], double vout[3][3]);

static std::vector<V3> bound_vertices(const Shape& s, const Tf& tf) {
  std::vector<V3> r;
  auto add = [&](double x, double y, double z) { r.push_back(tf.transform(V3(x, y, z))); };
  switch (s.kind) {
    case K_BOX: {
      const double a = s.p[0], b = s.p[1], c = s.p[2];
      add(a, b, c); add(a, b, -c); add(a, -b, c); add(a, -b, -c);
      add(-a, b, c); add(-a, b, -c); add(-a, -b, c); add(-a, -b, -c);
      break;
    }
    case K_SPHERE: {  // icosahedron
      const double m = (1 + std::sqrt(5.0)) / 2.0;
      const double edge = s.p[0] * 6 / (std::sqrt(27.0) + std::sqrt(15.0));
      const double a = edge, b = m * edge;
      add(0, a, b); add(0, -a, b); add(0, a, -b); add(0, -a, -b);
      add(a, b, 0); add(-a, b, 0); add(a, -b, 0); add(-a, -b, 0);
      add(b, 0, a); add(b, 0, -a); add(-b, 0, a); add(-b, 0, -a);
      break;
    }
    case K_ELLIPSOID: {
      const double phi = (1 + std::sqrt(5.0)) / 2.0;
      const double a = std::sqrt(3.0) / (phi * phi), b = phi * a;
      const double A = s.p[0], B = s.p[1], C = s.p[2];
      const double Aa = A * a, Ab = A * b, Ba = B * a, Bb = B * b, Ca = C * a, Cb = C * b;
      add(0, Ba, Cb); add(0, -Ba, Cb); add(0, Ba, -Cb); add(0, -Ba, -Cb);
      add(Aa, Bb, 0); add(-Aa, Bb, 0); add(Aa, -Bb, 0); add(-Aa, -Bb, 0);
      add(Ab, 0, Ca); add(Ab, 0, -Ca); add(-Ab, 0, Ca); add(-Ab, 0, -Ca);
      break;
    }
    case K_CAPSULE: {
      const double m = (1 + std::sqrt(5.0)) / 2.0;
      const double hl = s.p[1];
      const double edge = s.p[0] * 6 / (std::sqrt(27.0) + std::sqrt(15.0));
      const double a = edge, b = m * edge, r2 = s.p[0] * 2 / std::sqrt(3.0);
      add(0, a, b + hl); add(0, -a, b + hl); add(0, a, -b + hl); add(0, -a, -b + hl);
      add(a, b, hl); add(-a, b, hl); add(a, -b, hl); add(-a, -b, hl);
      add(b, 0, a + hl); add(b, 0, -a + hl); add(-b, 0, a + hl); add(-b, 0, -a + hl);
      add(0, a, b - hl); add(0, -a, b - hl); add(0, a, -b - hl); add(0, -a, -b - hl);
      add(a, b, -hl); add(-a, b, -hl); add(a, -b, -hl); add(-a, -b, -hl);
      add(b, 0, a - hl); add(b, 0, -a - hl); add(-b, 0, a - hl); add(-b, 0, -a - hl);
      const double c = 0.5 * r2, d = s.p[0];
      add(r2, 0, hl); add(c, d, hl); add(-c, d, hl); add(-r2, 0, hl); add(-c, -d, hl); add(c, -d, hl);
      add(r2, 0, -hl); add(c, d, -hl); add(-c, d, -hl); add(-r2, 0, -hl); add(-c, -d, -hl); add(c, -d, -hl);
      break;
    }
    case K_CONE: {
      const double hl = s.p[1], r2 = s.p[0] * 2 / std::sqrt(3.0), a = 0.5 * r2, b = s.p[0];
      add(r2, 0, -hl); add(a, b, -hl); add(-a, b, -hl); add(-r2, 0, -hl); add(-a, -b, -hl); add(a, -b, -hl);
      add(0, 0, hl);
      break;
    }
    case K_CYLINDER: {
      const double hl = s.p[1], r2 = s.p[0] * 2 / std::sqrt(3.0), a = 0.5 * r2, b = s.p[0];
      add(r2, 0, -hl); add(a, b, -hl); add(-a, b, -hl); add(-r2, 0, -hl); add(-a, -b, -hl); add(a, -b, -hl);
      add(r2, 0, hl); add(a, b, hl); add(-a, b, hl); add(-r2, 0, hl); add(-a, -b, hl); add(a, -b, hl);
      break;
    }
    case K_CONVEX:
      for (int i = 0; i < s.nverts; ++i) add(s.verts[3 * i], s.verts[3 * i + 1], s.verts[3 * i + 2]);
      break;
    default: break;
  }
  return r;
}

// fit(ps, n, OBBRSS&) for n > 3 (and n != 1,2,3): OBB fitn + RSS fitn on the same axes
int fit_points_obbrss(const std::vector<V3>& ps, hfcl_bvh_node& bv) {
  const unsigned n = unsigned(ps.size());
  if (n < 4) return HFCL_ERR_UNSUPPORTED_PAIR;  // fit1 / fit2 / fit3 are not restated
  std::memset(&bv, 0, sizeof(bv));
  // getCovariance, point-cloud branch (:222-244)
  V3 S1(0, 0, 0);
  double S2[3][3] = {{0, 0, 0}, {0, 0, 0}, {0, 0, 0}};
  for (unsigned i = 0; i < n; ++i) {
    const V3& p = ps[i];
    S1 = S1 + p;
    S2[0][0] += (p[0] * p[0]);
    S2[1][1] += (p[1] * p[1]);
    S2[2][2] += (p[2] * p[2]);
    S2[0][1] += (p[0] * p[1]);
    S2[0][2] += (p[0] * p[2]);
    S2[1][2] += (p[1] * p[2]);
  }
  const unsigned n_points = n;
  double M[3][3];
  M[0][0] = S2[0][0] - S1[0] * S1[0] / n_points;
  M[1][1] = S2[1][1] - S1[1] * S1[1] / n_points;
  M[2][2] = S2[2][2] - S1[2] * S1[2] / n_points;
  M[0][1] = S2[0][1] - S1[0] * S1[1] / n_points;
  M[1][2] = S2[1][2] - S1[1] * S1[2] / n_points;
  M[0][2] = S2[0][2] - S1[0] * S1[2] / n_points;
  M[1][0] = M[0][1];
  M[2][0] = M[0][2];
  M[2][1] = M[1][2];
  double sv[3], E[3][3];
  jacobi_eigen3(M, sv, E);
  int mn, mid, mx;  // axisFromEigen, BV_fitter.cpp:50-76
  if (sv[0] > sv[1]) {
    mx = 0;
    mn = 1;
  } else {
    mn = 0;
    mx = 1;
  }
  if (sv[2] < sv[mn]) {
    mid = mn;
    mn = 2;
  } else if (sv[2] > sv[mx]) {
    mid = mx;
    mx = 2;
  } else {
    mid = 2;
  }
  double ax[3][3];
  for (int r = 0; r < 3; ++r) {
    ax[0][r] = E[r][mx];
    ax[1][r] = E[r][mid];
  }
  ax[2][0] = E[1][mx] * E[2][mid] - E[1][mid] * E[2][mx];
  ax[2][1] = E[0][mid] * E[2][mx] - E[0][mx] * E[2][mid];
  ax[2][2] = E[0][mx] * E[1][mid] - E[0][mid] * E[1][mx];
  for (int c = 0; c < 3; ++c)
    for (int r = 0; r < 3; ++r) bv.obb_axes[3 * c + r] = bv.rss_axes[3 * c + r] = ax[c][r];
  // getExtentAndCenter_pointcloud (:485-527): proj = axes^T p
  const double real_max = std::numeric_limits<double>::max();
  double mnc[3] = {real_max, real_max, real_max}, mxc[3] = {-real_max, -real_max, -real_max};
  std::vector<double> P(size_t(3) * n);
  for (unsigned i = 0; i < n; ++i)
    for (int k = 0; k < 3; ++k) {
      const double q = ax[k][0] * ps[i][0] + ax[k][1] * ps[i][1] + ax[k][2] * ps[i][2];
      P[3 * size_t(i) + k] = q;
      if (q > mxc[k]) mxc[k] = q;
      if (q < mnc[k]) mnc[k] = q;
    }
  // center = axes * (max + min) / 2  [= (axes * (max+min)) / 2];  extent = (max - min) / 2
  double sum[3];
  for (int k = 0; k < 3; ++k) {
    sum[k] = mxc[k] + mnc[k];
    bv.obb_extent[k] = (mxc[k] - mnc[k]) / 2;
  }
  for (int r = 0; r < 3; ++r) bv.obb_To[r] = (ax[0][r] * sum[0] + ax[1][r] * sum[1] + ax[2][r] * sum[2]) / 2;
  // getRadiusAndOriginAndRectangleSize, point-cloud branch (same body as oracle/bvh_build.cpp)
  const size_t size_P = n;
  auto Px = [&](size_t i, int k) { return P[3 * i + k]; };
  double minz = Px(0, 2), maxz = Px(0, 2);
  for (size_t i = 1; i < size_P; ++i) {
    const double zv = Px(i, 2);
    if (zv < minz)
      minz = zv;
    else if (zv > maxz)
      maxz = zv;
  }
  const double r = 0.5 * (maxz - minz), radsqr = r * r, cz = 0.5 * (maxz + minz);
  double lo[2], hi[2];
  for (int k = 0; k < 2; ++k) {
    size_t minindex = 0, maxindex = 0;
    double mintmp = Px(0, k), maxtmp = Px(0, k);
    for (size_t i = 1; i < size_P; ++i) {
      const double v = Px(i, k);
      if (v < mintmp) {
        minindex = i;
        mintmp = v;
      } else if (v > maxtmp) {
        maxindex = i;
        maxtmp = v;
      }
    }
    double dz = Px(minindex, 2) - cz;
    double mnv = Px(minindex, k) + std::sqrt(std::max<double>(radsqr - dz * dz, 0));
    dz = Px(maxindex, 2) - cz;
    double mxv = Px(maxindex, k) - std::sqrt(std::max<double>(radsqr - dz * dz, 0));
    for (size_t i = 0; i < size_P; ++i) {
      if (Px(i, k) < mnv) {
        dz = Px(i, 2) - cz;
        const double x = Px(i, k) + std::sqrt(std::max<double>(radsqr - dz * dz, 0));
        if (x < mnv) mnv = x;
      } else if (Px(i, k) > mxv) {
        dz = Px(i, 2) - cz;
        const double x = Px(i, k) - std::sqrt(std::max<double>(radsqr - dz * dz, 0));
        if (x > mxv) mxv = x;
      }
    }
    lo[k] = mnv;
    hi[k] = mxv;
  }
  double minx = lo[0], maxx = hi[0], miny = lo[1], maxy = hi[1];
  const double a = std::sqrt(0.5);
  for (size_t i = 0; i < size_P; ++i) {
    double dx, dy, u, t;
    const double px = Px(i, 0), py = Px(i, 1), pz = Px(i, 2);
    if (px > maxx) {
      if (py > maxy) {
        dx = px - maxx;
        dy = py - maxy;
        u = dx * a + dy * a;
        t = (a * u - dx) * (a * u - dx) + (a * u - dy) * (a * u - dy) + (cz - pz) * (cz - pz);
        u = u - std::sqrt(std::max<double>(radsqr - t, 0));
        if (u > 0) {
          maxx += u * a;
          maxy += u * a;
        }
      } else if (py < miny) {
        dx = px - maxx;
        dy = py - miny;
        u = dx * a - dy * a;
        t = (a * u - dx) * (a * u - dx) + (-a * u - dy) * (-a * u - dy) + (cz - pz) * (cz - pz);
        u = u - std::sqrt(std::max<double>(radsqr - t, 0));
        if (u > 0) {
          maxx += u * a;
          miny -= u * a;
        }
      }
    } else if (px < minx) {
      if (py > maxy) {
        dx = px - minx;
        dy = py - maxy;
        u = dy * a - dx * a;
        t = (-a * u - dx) * (-a * u - dx) + (a * u - dy) * (a * u - dy) + (cz - pz) * (cz - pz);
        u = u - std::sqrt(std::max<double>(radsqr - t, 0));
        if (u > 0) {
          minx -= u * a;
          maxy += u * a;
        }
      } else if (py < miny) {
        dx = px - minx;
        dy = py - miny;
        u = -dx * a - dy * a;
        t = (-a * u - dx) * (-a * u - dx) + (-a * u - dy) * (-a * u - dy) + (cz - pz) * (cz - pz);
        u = u - std::sqrt(std::max<double>(radsqr - t, 0));
        if (u > 0) {
          minx -= u * a;
          miny -= u * a;
        }
      }
    }
  }
  for (int rr = 0; rr < 3; ++rr) bv.rss_Tr[rr] = ax[0][rr] * minx + ax[1][rr] * miny + ax[2][rr] * cz;
  bv.rss_length[0] = std::max<double>(maxx - minx, 0);
  bv.rss_length[1] = std::max<double>(maxy - miny, 0);
  bv.rss_radius = r;
  return HFCL_OK;
}

// include/hpp/fcl/internal/tools.h:60-87
static void generate_coordinate_system(const V3& w, V3& u, V3& v) {
  if (std::abs(w[0]) >= std::abs(w[1])) {
    const double inv_length = 1.0 / std::sqrt(w[0] * w[0] + w[2] * w[2]);
    u = V3(-w[2] * inv_length, 0, w[0] * inv_length);
    v = V3(w[1] * u[2], w[2] * u[0] - w[0] * u[2], -w[1] * u[0]);
  } else {
    const double inv_length = 1.0 / std::sqrt(w[1] * w[1] + w[2] * w[2]);
    u = V3(0, w[2] * inv_length, -w[1] * inv_length);
    v = V3(w[1] * u[2] - w[2] * u[1], -w[0] * u[2], w[0] * u[1]);
  }
}

int shape_obbrss(const Shape& s, const Tf& tf, hfcl_bvh_node& bv) {
  if (s.ssr > 0) return HFCL_ERR_UNSUPPORTED_PAIR;  // "Swept-sphere radius not yet supported."
  const double big = std::numeric_limits<double>::max();
  if (s.kind == K_HALFSPACE) {  // geometric_shapes_utility.cpp:545-581: "very rough" unbounded volumes
    std::memset(&bv, 0, sizeof(bv));
    for (int k = 0; k < 3; ++k) {
      bv.obb_axes[4 * k] = bv.rss_axes[4 * k] = 1;
      bv.obb_extent[k] = big;
    }
    bv.rss_length[0] = bv.rss_length[1] = bv.rss_radius = big;
    return HFCL_OK;
  }
  if (s.kind == K_PLANE) {  // :803-850
    std::memset(&bv, 0, sizeof(bv));
    const V3 n = tf.R * V3(s.p[0], s.p[1], s.p[2]);
    V3 u, v;
    generate_coordinate_system(n, u, v);
    for (int r = 0; r < 3; ++r) {
      bv.obb_axes[r] = bv.rss_axes[r] = n[r];
      bv.obb_axes[3 + r] = bv.rss_axes[3 + r] = u[r];
      bv.obb_axes[6 + r] = bv.rss_axes[6 + r] = v[r];
    }
    bv.obb_extent[0] = 0;
    bv.obb_extent[1] = bv.obb_extent[2] = big;
    const V3 p = tf.transform(V3(s.p[0], s.p[1], s.p[2]) * s.p[3]);
    for (int r = 0; r < 3; ++r) bv.obb_To[r] = bv.rss_Tr[r] = p[r];
    bv.rss_length[0] = bv.rss_length[1] = big;
    bv.rss_radius = 0;
    return HFCL_OK;
  }
  const std::vector<V3> pts = bound_vertices(s, tf);
  if (pts.empty()) return HFCL_ERR_UNSUPPORTED_PAIR;
  return fit_points_obbrss(pts, bv);
}

namespace {
struct MeshShapeTraversal {
  const MeshView& m1;
  const Tf tf1, tf2;
  const Shape& s2;
  const hfcl_collision_request& req;
  hfcl_bvh_node bv2;
  GJKSolver solver;
  BvhStats stats;
  double distance_lower_bound = std::numeric_limits<double>::max();
  double record_distance = std::numeric_limits<double>::max();
  V3 np1, np2, normal;
  std::vector<hfcl_contact> contacts;
  uint32_t pair_index = 0;

  MeshShapeTraversal(const MeshView& a, const Tf& t1, const Shape& b, const Tf& t2, const hfcl_collision_request& r)
      : m1(a), tf1(t1), tf2(t2), s2(b), req(r) {
    const double nanv = std::numeric_limits<double>::quiet_NaN();
    np1 = np2 = normal = V3(nanv, nanv, nanv);
    solver.set(req);
    solver.out_cached_guess = solver.cached_guess;
    solver.out_support_guess[0] = solver.support_func_cached_guess[0];
    solver.out_support_guess[1] = solver.support_func_cached_guess[1];
  }
  bool can_stop() const { return !contacts.empty() && contacts.size() >= req.num_max_contacts; }
  bool bv_disjoints(unsigned b1) {  // traversal_node_bvh_shape.h:121-138 (oriented)
    ++stats.num_bv_tests;
    double sq;
    const bool disjoint = !obb_overlap(tf1.R, tf1.T, m1.nodes[b1], bv2, req.security_margin, req.break_distance, sq);
    if (disjoint && !(distance_lower_bound <= 0)) {  // updateDistanceLowerBoundFromBV
      const double nd = std::sqrt(sq);
      if (nd < distance_lower_bound) {
        distance_lower_bound = nd;
        record_distance = nd + req.security_margin;
      }
    }
    return disjoint;
  }
  bool leaf_collides(unsigned b1) {  // :141-186; false = unsupported shape pair
    ++stats.num_leaf_tests;
    const int pid = -(m1.nodes[b1].first_child + 1);
    double t[9];
    for (int k = 0; k < 3; ++k)
      for (int c = 0; c < 3; ++c) t[3 * k + c] = m1.verts[3 * size_t(m1.tris[3 * pid + k]) + c];
    Shape tri;
    tri.kind = K_TRIANGLE;
    tri.verts = t;
    tri.nverts = 3;
    const bool compute_penetration = req.enable_contact || (req.security_margin < 0);
    double distance;
    V3 p1, p2, n;
    if (!shape_shape_distance(tri, tf1, s2, tf2, solver, compute_penetration, distance, p1, p2, n)) return false;
    // the solver persists over the traversal: runGJKAndEPA leaves its caches in it (narrowphase.h:556-586)
    solver.cached_guess = solver.out_cached_guess;
    solver.support_func_cached_guess[0] = solver.out_support_guess[0];
    solver.support_func_cached_guess[1] = solver.out_support_guess[1];
    const double dtc = distance - req.security_margin;
    if (dtc < distance_lower_bound) {
      distance_lower_bound = dtc;
      record_distance = distance;
      np1 = p1;
      np2 = p2;
      normal = n;
    }
    if (dtc <= req.q.collision_distance_threshold && contacts.size() < req.num_max_contacts) {
      hfcl_contact c;
      c.pair = pair_index;
      c.b1 = pid;
      c.b2 = -1;  // Contact::NONE
      c._pad = 0;
      c.penetration_depth = distance;
      for (int k = 0; k < 3; ++k) {
        c.normal[k] = n[k];
        c.p1[k] = p1[k];
        c.p2[k] = p2[k];
      }
      contacts.push_back(c);
    }
    return true;
  }
  bool recurse(unsigned b1) {
    const hfcl_bvh_node& n1 = m1.nodes[b1];
    if (n1.first_child < 0) return leaf_collides(b1);
    if (bv_disjoints(b1)) return true;
    const unsigned c1 = unsigned(n1.first_child);
    if (!recurse(c1)) return false;
    if (can_stop()) return true;
    return recurse(c1 + 1);
  }
};
}  // namespace

// ---- distance(): BVHShapeDistancer<OBBRSS,S> (src/distance_func_matrix.cpp:111-168),
// MeshShapeDistanceTraversalNodeOBBRSS (traversal_node_bvh_shape.h:276-478), distanceRecurse
// (src/traversal/traversal_recurse.cpp:153-203) with a leaf second node; rel_err = abs_err = 0 (the node's
// constructor values: setupMeshShapeDistanceOrientedNode does not copy them, traversal_node_setup.h:747-772).
double rss_distance(const M3& R0, const V3& T0, const hfcl_bvh_node& b1, const hfcl_bvh_node& b2);

namespace {
struct MeshShapeDistTraversal {
  const MeshView& m1;
  const Tf tf1, tf2;
  const Shape& s2;
  hfcl_bvh_node bv2;
  GJKSolver solver;
  bool signed_distance;
  double min_distance = std::numeric_limits<double>::max();
  int b1 = -1;
  V3 np1, np2, normal;
  bool ok = true;
  // (diagnostic, bvh_shape_distance_trace) one record of four numbers per event: a child's canStop test {0, node, bound, minimum at the
  // test} and a triangle's evaluation {1, triangle, distance, minimum before it}
  std::vector<double>* trace = nullptr;
  MeshShapeDistTraversal(const MeshView& a, const Tf& t1, const Shape& b, const Tf& t2, const hfcl_distance_request& r)
      : m1(a), tf1(t1), tf2(t2), s2(b), signed_distance(r.enable_signed_distance != 0) {
    const double nanv = std::numeric_limits<double>::quiet_NaN();
    np1 = np2 = normal = V3(nanv, nanv, nanv);
    solver.set(r);
    solver.out_cached_guess = solver.cached_guess;
    solver.out_support_guess[0] = solver.support_func_cached_guess[0];
    solver.out_support_guess[1] = solver.support_func_cached_guess[1];
  }
  void leaf(int pid) {
    double t[9];
    for (int k = 0; k < 3; ++k)
      for (int c = 0; c < 3; ++c) t[3 * k + c] = m1.verts[3 * size_t(m1.tris[3 * pid + k]) + c];
    Shape tri;
    tri.kind = K_TRIANGLE;
    tri.verts = t;
    tri.nverts = 3;
    double d;
    V3 p1, p2, n;
    if (!shape_shape_distance(tri, tf1, s2, tf2, solver, signed_distance, d, p1, p2, n)) {
      ok = false;
      return;
    }
    solver.cached_guess = solver.out_cached_guess;
    solver.support_func_cached_guess[0] = solver.out_support_guess[0];
    solver.support_func_cached_guess[1] = solver.out_support_guess[1];
    if (trace) trace->insert(trace->end(), {1.0, double(pid), d, min_distance});
    if (min_distance > d) {  // DistanceResult::update (collision_data.h:1115-1160)
      min_distance = d;
      b1 = pid;
      np1 = p1;
      np2 = p2;
      normal = n;
    }
  }
  bool can_stop(double c) const { return c >= min_distance; }
  double lower_bound(unsigned b) const { return rss_distance(tf1.R, tf1.T, bv2, m1.nodes[b]); }
  void recurse(unsigned b) {
    const hfcl_bvh_node& n1 = m1.nodes[b];
    if (n1.first_child < 0) {
      leaf(-(n1.first_child + 1));
      return;
    }
    const unsigned a1 = unsigned(n1.first_child), c1 = a1 + 1;
    const double d1 = lower_bound(a1), d2 = lower_bound(c1);
    auto visit = [&](unsigned c, double d) {
      if (trace) trace->insert(trace->end(), {0.0, double(c), d, min_distance});
      if (!can_stop(d)) recurse(c);
    };
    if (d2 < d1) {
      visit(c1, d2);
      visit(a1, d1);
    } else {
      visit(a1, d1);
      visit(c1, d2);
    }
  }
};
}  // namespace

// distance of ONE triangle of a mesh x solid query, as the traversal's leaf computes it with the request's solver settings
// (default guess: the value does not depend on the leaves visited before) -- test infrastructure for enumerated ties
double bvh_shape_leaf_distance(const MeshView& m1, const Tf& tf1, const Shape& s2, const Tf& tf2, const hfcl_distance_request& req, int pid) {
  MeshShapeDistTraversal t(m1, tf1, s2, tf2, req);
  t.leaf(pid);
  return t.ok ? t.min_distance : std::numeric_limits<double>::quiet_NaN();
}

// (diagnostic) the events of one query's walk, four numbers each (MeshShapeDistTraversal::trace); returns the number of events
size_t bvh_shape_distance_trace(const MeshView& m1, const Tf& tf1, const Shape& s2, const Tf& tf2, const hfcl_distance_request& req, double* out, size_t cap) {
  MeshShapeDistTraversal t(m1, tf1, s2, tf2, req);
  if (shape_obbrss(s2, tf2, t.bv2)) return 0;
  std::vector<double> ev;
  t.trace = &ev;
  t.leaf(0);
  if (t.ok) t.recurse(0);
  const size_t n = ev.size() / 4;
  for (size_t k = 0; k < std::min(n, cap) * 4; ++k) out[k] = ev[k];
  return n;
}

int bvh_shape_distance_pair(const MeshView& m1, const Tf& tf1, const Shape& s2, const Tf& tf2, const hfcl_distance_request& req,
                            bool swapped, hfcl_result& out, hfcl_guess* guess_out) {
  MeshShapeDistTraversal t(m1, tf1, s2, tf2, req);
  int rc = shape_obbrss(s2, tf2, t.bv2);
  if (rc) return rc;
  t.leaf(0);  // preprocess(): triangle 0
  if (t.ok) t.recurse(0);
  if (!t.ok) return HFCL_ERR_UNSUPPORTED_PAIR;
  out.distance = t.min_distance;
  for (int k = 0; k < 3; ++k) {  // distance.cpp:84-88 swaps o1/o2, the points and the normal -- not b1/b2
    out.normal[k] = swapped ? -t.normal[k] : t.normal[k];
    out.p1[k] = swapped ? t.np2[k] : t.np1[k];
    out.p2[k] = swapped ? t.np1[k] : t.np2[k];
  }
  out.b1 = t.b1;
  out.b2 = -1;
  out.status = (t.min_distance <= 0) ? 128u : 0u;
  out.num_contacts = 0;
  if (guess_out) {
    for (int k = 0; k < 3; ++k) guess_out->gjk_guess[k] = t.solver.cached_guess[k];
    guess_out->support_guess[0] = t.solver.support_func_cached_guess[0];
    guess_out->support_guess[1] = t.solver.support_func_cached_guess[1];
  }
  return HFCL_OK;
}

// (BVH, shape) in this order; `swapped`: the caller had (shape, BVH) -> collision.cpp:101-107
int bvh_shape_collide_pair(const MeshView& m1, const Tf& tf1, const Shape& s2, const Tf& tf2, const hfcl_collision_request& req,
                           bool swapped, hfcl_result& out, std::vector<hfcl_contact>* contacts, uint32_t pair_index,
                           hfcl_guess* guess_out, BvhStats* stats) {
  if (req.num_max_contacts == 0) return HFCL_ERR_INVALID_ARGUMENT;
  const double nanv = std::numeric_limits<double>::quiet_NaN();
  if (req.security_margin == -std::numeric_limits<double>::infinity()) {
    out.distance = std::numeric_limits<double>::max();
    for (int k = 0; k < 3; ++k) out.normal[k] = out.p1[k] = out.p2[k] = nanv;
    out.b1 = out.b2 = -1;
    out.status = 0x80000000u;
    out.num_contacts = 0;
    return HFCL_OK;
  }
  if (req.security_margin < 0) return HFCL_ERR_INVALID_ARGUMENT;  // collision_func_matrix.cpp:109-112
  MeshShapeTraversal t(m1, tf1, s2, tf2, req);
  t.pair_index = pair_index;
  int rc = shape_obbrss(s2, tf2, t.bv2);
  if (rc) return rc;
  if (!t.recurse(0)) return HFCL_ERR_UNSUPPORTED_PAIR;
  out.distance = t.record_distance;
  for (int k = 0; k < 3; ++k) {
    out.normal[k] = swapped ? -t.normal[k] : t.normal[k];
    out.p1[k] = swapped ? t.np2[k] : t.np1[k];
    out.p2[k] = swapped ? t.np1[k] : t.np2[k];
  }
  out.num_contacts = int(t.contacts.size());
  const int fb = t.contacts.empty() ? -1 : t.contacts[0].b1;
  out.b1 = swapped ? -1 : fb;
  out.b2 = swapped ? fb : -1;
  out.status = t.contacts.empty() ? 0u : 128u;
  if (contacts)
    for (hfcl_contact c : t.contacts) {
      if (swapped) {  // CollisionResult::swapObjects + Contact fields (src/collision.cpp:52-60)
        std::swap(c.b1, c.b2);
        for (int k = 0; k < 3; ++k) {
          c.normal[k] = -c.normal[k];
          std::swap(c.p1[k], c.p2[k]);
        }
      }
      contacts->push_back(c);
    }
  if (guess_out) {
    for (int k = 0; k < 3; ++k) guess_out->gjk_guess[k] = t.solver.cached_guess[k];
    guess_out->support_guess[0] = t.solver.support_func_cached_guess[0];
    guess_out->support_guess[1] = t.solver.support_func_cached_guess[1];
  }
  if (stats) *stats = t.stats;
  return HFCL_OK;
}

}  // namespace orc
