// ORACLE -- TEST INFRASTRUCTURE ONLY.  Not part of the product path (see vec3.hpp).
#include "bvh.hpp"
#include <algorithm>
#include <cmath>

namespace orc {

static inline M3 axes_of(const double* p) {  // column-major 9 doubles -> row-major M3
  M3 m;
  for (int c = 0; c < 3; ++c)
    for (int r = 0; r < 3; ++r) m.m[r][c] = p[c * 3 + r];
  return m;
}
static inline V3 v3(const double* p) { return V3(p[0], p[1], p[2]); }

// obbDisjointAndLowerBoundDistance, src/BV/OBB.cpp:344-393 (+ helpers :290-336)
static bool obb_disjoint_lb(const M3& B, const V3& T, const V3& a_, const V3& b_, double security_margin,
                            double break_distance, double& sq) {
  const double breakDistance2 = break_distance * break_distance;
  M3 Bf;
  for (int i = 0; i < 3; ++i)
    for (int j = 0; j < 3; ++j) Bf.m[i][j] = std::fabs(B.m[i][j]);
  V3 a, b;
  for (int i = 0; i < 3; ++i) {
    a[i] = std::max(a_[i] + security_margin / 2, 0.0);
    b[i] = std::max(b_[i] + security_margin / 2, 0.0);
  }
  // obbDisjoint_check_A_axis
  {
    V3 corner(std::fabs(T.x) - a.x, std::fabs(T.y) - a.y, std::fabs(T.z) - a.z);
    corner -= Bf * b;
    double s = 0;
    for (int i = 0; i < 3; ++i) {
      double c = std::max(corner[i], 0.0);
      s += c * c;
    }
    sq = s;
  }
  if (sq > breakDistance2) return true;
  // obbDisjoint_check_B_axis
  {
    double s, t = 0;
    for (int k = 0; k < 3; ++k) {
      s = std::fabs(dot(B.col(k), T)) - dot(Bf.col(k), a) - b[k];
      if (s > 0) t += s * s;
    }
    sq = t;
  }
  if (sq > breakDistance2) return true;
  // Ai x Bj
  int ja = 1, ka = 2;
  for (int ia = 0; ia < 3; ++ia) {
    for (int ib = 0; ib < 3; ++ib) {
      const int jb = (ib + 1) % 3, kb = (ib + 2) % 3;
      const double sinus2 = 1 - Bf.m[ia][ib] * Bf.m[ia][ib];
      if (sinus2 < 1e-6) continue;
      const double s = T[ka] * B.m[ja][ib] - T[ja] * B.m[ka][ib];
      const double diff = std::fabs(s) - (a[ja] * Bf.m[ka][ib] + a[ka] * Bf.m[ja][ib] + b[jb] * Bf.m[ia][kb] +
                                          b[kb] * Bf.m[ia][jb]);
      if (diff > 0) {
        sq = diff * diff / sinus2;
        if (sq > breakDistance2) return true;
      }
    }
    ja = ka;
    ka = ia;
  }
  return false;
}

// overlap(R0, T0, b1, b2, request, sqrDistLowerBound), src/BV/OBB.cpp:475-483
bool obb_overlap(const M3& R0, const V3& T0, const hfcl_bvh_node& b1, const hfcl_bvh_node& b2, double security_margin,
                 double break_distance, double& sq) {
  const M3 A1 = axes_of(b1.obb_axes), A2 = axes_of(b2.obb_axes);
  const V3 Ttemp = tmul(R0, v3(b2.obb_To) - T0) - v3(b1.obb_To);
  const V3 T = tmul(A1, Ttemp);
  const M3 R = tmul(A1, tmul(R0, A2));  // b1.axes^T * R0^T * b2.axes
  return !obb_disjoint_lb(R, T, v3(b1.obb_extent), v3(b2.obb_extent), security_margin, break_distance, sq);
}

namespace {
struct Traversal {
  const MeshView& m1;
  const MeshView& m2;
  Tf tf1, tf2;
  M3 RT_R;
  V3 RT_T;
  const hfcl_collision_request& req;
  // CollisionResult state
  double distance_lower_bound = std::numeric_limits<double>::max();
  V3 np1 = nan3(), np2 = nan3(), normal = nan3();
  double record_distance = std::numeric_limits<double>::max();
  std::vector<hfcl_contact> contacts;
  uint32_t pair_index = 0;
  BvhStats stats;

  Traversal(const MeshView& a, const Tf& t1, const MeshView& b, const Tf& t2, const hfcl_collision_request& r)
      : m1(a), m2(b), tf1(t1), tf2(t2), req(r) {
    RT_R = tmul(tf1.R, tf2.R);  // traversal_node_setup.h:560-563
    RT_T = tmul(tf1.R, tf2.T - tf1.T);
  }
  bool can_stop() const { return !contacts.empty() && contacts.size() >= req.num_max_contacts; }
  static bool is_leaf(const hfcl_bvh_node& n) { return n.first_child < 0; }
  static double size_of(const hfcl_bvh_node& n) {  // OBBRSS::size() = obb.extent.squaredNorm(), OBBRSS.h:114
    return n.obb_extent[0] * n.obb_extent[0] + n.obb_extent[1] * n.obb_extent[1] + n.obb_extent[2] * n.obb_extent[2];
  }
  bool bv_disjoints(unsigned b1, unsigned b2) {  // traversal_node_bvhs.h:152-168
    ++stats.num_bv_tests;
    double sq;
    // NOTE the argument order of the reference: (RT.R, RT.T, model2.bv(b2), model1.bv(b1))
    const bool disjoint = !obb_overlap(RT_R, RT_T, m2.nodes[b2], m1.nodes[b1], req.security_margin, req.break_distance, sq);
    if (disjoint) {  // updateDistanceLowerBoundFromBV, collision_data.h:1177-1184
      if (!(distance_lower_bound <= 0)) {
        const double new_dlb = std::sqrt(sq);
        if (new_dlb < distance_lower_bound) {
          distance_lower_bound = new_dlb;
          record_distance = new_dlb + req.security_margin;
        }
      }
    }
    return disjoint;
  }
  void leaf_collides(unsigned b1, unsigned b2) {  // traversal_node_bvhs.h:184-233
    ++stats.num_leaf_tests;
    const int pid1 = -(m1.nodes[b1].first_child + 1), pid2 = -(m2.nodes[b2].first_child + 1);
    double t1[9], t2[9];
    for (int k = 0; k < 3; ++k)
      for (int c = 0; c < 3; ++c) {
        t1[3 * k + c] = m1.verts[3 * size_t(m1.tris[3 * pid1 + k]) + c];
        t2[3 * k + c] = m2.verts[3 * size_t(m2.tris[3 * pid2 + k]) + c];
      }
    Shape s1, s2;
    s1.kind = s2.kind = K_TRIANGLE;
    s1.verts = t1;
    s2.verts = t2;
    s1.nverts = s2.nverts = 3;
    GJKSolver solver;  // GJKSolver solver(this->request), traversal_node_bvhs.h:208
    solver.set(req);
    const bool compute_penetration = req.enable_contact || (req.security_margin < 0);
    double distance;
    V3 p1, p2, n;
    shape_shape_distance(s1, tf1, s2, tf2, solver, compute_penetration, distance, p1, p2, n);
    const double dtc = distance - req.security_margin;
    if (dtc < distance_lower_bound) {  // updateDistanceLowerBoundFromLeaf
      distance_lower_bound = dtc;
      record_distance = distance;
      np1 = p1;
      np2 = p2;
      normal = n;
    }
    if (dtc <= req.q.collision_distance_threshold) {
      if (contacts.size() < req.num_max_contacts) {
        hfcl_contact c;
        c.pair = pair_index;
        c.b1 = pid1;
        c.b2 = pid2;
        c._pad = 0;
        c.penetration_depth = distance;
        for (int k = 0; k < 3; ++k) {
          c.normal[k] = n[k];
          c.p1[k] = p1[k];
          c.p2[k] = p2[k];
        }
        contacts.push_back(c);
      }
    }
  }
  void recurse(unsigned b1, unsigned b2) {  // collisionRecurse, traversal_recurse.cpp:44-85
    const hfcl_bvh_node& n1 = m1.nodes[b1];
    const hfcl_bvh_node& n2 = m2.nodes[b2];
    const bool l1 = is_leaf(n1), l2 = is_leaf(n2);
    if (l1 && l2) {
      leaf_collides(b1, b2);
      return;
    }
    if (bv_disjoints(b1, b2)) return;
    // firstOverSecond, traversal_node_bvhs.h:89-98
    const bool first = l2 || (!l1 && (size_of(n1) > size_of(n2)));
    if (first) {
      const unsigned c1 = unsigned(n1.first_child), c2 = c1 + 1;
      recurse(c1, b2);
      if (can_stop()) return;
      recurse(c2, b2);
    } else {
      const unsigned c1 = unsigned(n2.first_child), c2 = c1 + 1;
      recurse(b1, c1);
      if (can_stop()) return;
      recurse(b1, c2);
    }
  }
};
}  // namespace

int bvh_collide_pair(const MeshView& m1, const Tf& tf1, const MeshView& m2, const Tf& tf2,
                     const hfcl_collision_request& req, hfcl_result& out, std::vector<hfcl_contact>* contacts,
                     uint32_t pair_index, BvhStats* stats) {
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
  Traversal t(m1, tf1, m2, tf2, req);
  t.pair_index = pair_index;
  t.recurse(0, 0);
  out.distance = t.record_distance;
  for (int k = 0; k < 3; ++k) {
    out.normal[k] = t.normal[k];
    out.p1[k] = t.np1[k];
    out.p2[k] = t.np2[k];
  }
  out.num_contacts = int(t.contacts.size());
  out.b1 = t.contacts.empty() ? -1 : t.contacts[0].b1;
  out.b2 = t.contacts.empty() ? -1 : t.contacts[0].b2;
  out.status = t.contacts.empty() ? 0u : 128u;
  if (contacts) contacts->insert(contacts->end(), t.contacts.begin(), t.contacts.end());
  if (stats) *stats = t.stats;
  return HFCL_OK;
}

}  // namespace orc
