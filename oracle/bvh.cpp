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

// =============================================================================================
// distance()
// =============================================================================================
namespace orc {

static inline void clip_to_range(double& val, double a, double b) {  // RSS.cpp:49-54
  if (val < a)
    val = a;
  else if (val > b)
    val = b;
}
static void seg_coords(double& t, double& u, double a, double b, double A_dot_B, double A_dot_T, double B_dot_T) {  // :67-88
  double denom = 1 - A_dot_B * A_dot_B;
  if (denom == 0)
    t = 0;
  else {
    t = (A_dot_T - B_dot_T * A_dot_B) / denom;
    clip_to_range(t, 0, a);
  }
  u = t * A_dot_B - B_dot_T;
  if (u < 0) {
    u = 0;
    t = A_dot_T;
    clip_to_range(t, 0, a);
  } else if (u > b) {
    u = b;
    t = u * A_dot_B + A_dot_T;
    clip_to_range(t, 0, a);
  }
}
static bool in_voronoi(double a, double b, double Anorm_dot_B, double Anorm_dot_T, double A_dot_B, double A_dot_T,
                       double B_dot_T) {  // :95-116
  if (std::fabs(Anorm_dot_B) < 1e-7) return false;
  double t, u, v;
  u = -Anorm_dot_T / Anorm_dot_B;
  clip_to_range(u, 0, b);
  t = u * A_dot_B + A_dot_T;
  clip_to_range(t, 0, a);
  v = t * A_dot_B - B_dot_T;
  if (Anorm_dot_B > 0) {
    if (v > (u + 1e-7)) return true;
  } else {
    if (v < (u - 1e-7)) return true;
  }
  return false;
}

// rectDistance, RSS.cpp:121-713.  The sixteen edge-pair blocks of the reference are written as
// sixteen calls of one local helper; every argument is the reference's expression for that block.
double rect_distance(const M3& Rab, const V3& Tab, const double a[2], const double b[2]) {
  const double A0_dot_B0 = Rab.m[0][0], A0_dot_B1 = Rab.m[0][1], A1_dot_B0 = Rab.m[1][0], A1_dot_B1 = Rab.m[1][1];
  const double aA0_dot_B0 = a[0] * A0_dot_B0, aA0_dot_B1 = a[0] * A0_dot_B1, aA1_dot_B0 = a[1] * A1_dot_B0,
               aA1_dot_B1 = a[1] * A1_dot_B1;
  const double bA0_dot_B0 = b[0] * A0_dot_B0, bA1_dot_B0 = b[0] * A1_dot_B0, bA0_dot_B1 = b[1] * A0_dot_B1,
               bA1_dot_B1 = b[1] * A1_dot_B1;
  const V3 Tba = tmul(Rab, Tab);
  double result = 0;
  // One edge-pair block.  pre1/pre2: the outer `if`; skip1/skip2: the left operands of the two `||`;
  // v1/v2: the inVoronoi argument lists; sg: the segCoords arguments; then S = base + col*u (+const) - (t on axis ta)
  auto block = [&](bool pre1, bool pre2, bool skip1, const double (&v1)[7], bool skip2, const double (&v2)[7],
                   const double (&sg)[5], const V3& S0, int ucol, int taxis) -> bool {
    if (!(pre1 && pre2)) return false;
    if (!((skip1 || in_voronoi(v1[0], v1[1], v1[2], v1[3], v1[4], v1[5], v1[6])) &&
          (skip2 || in_voronoi(v2[0], v2[1], v2[2], v2[3], v2[4], v2[5], v2[6]))))
      return false;
    double t, u;
    seg_coords(t, u, sg[0], sg[1], sg[2], sg[3], sg[4]);
    V3 S(S0.x + Rab.m[0][ucol] * u, S0.y + Rab.m[1][ucol] * u, S0.z + Rab.m[2][ucol] * u);
    S[taxis] -= t;
    result = norm(S);
    return true;
  };
  // S0 bases: Tab (+ Rab col * b) (- a on the fixed A axis)
  auto base = [&](int bcol, double bv, int aaxis, double av) {
    V3 s = Tab;
    if (bcol >= 0) s = V3(Tab.x + Rab.m[0][bcol] * bv, Tab.y + Rab.m[1][bcol] * bv, Tab.z + Rab.m[2][bcol] * bv);
    if (aaxis >= 0) s[aaxis] -= av;
    return s;
  };

  double ALL_x = -Tba[0], ALU_x = ALL_x + aA1_dot_B0, AUL_x = ALL_x + aA0_dot_B0, AUU_x = ALU_x + aA0_dot_B0;
  double LA1_lx, LA1_ux, UA1_lx, UA1_ux, LB1_lx, LB1_ux, UB1_lx, UB1_ux;
  if (ALL_x < ALU_x) { LA1_lx = ALL_x; LA1_ux = ALU_x; UA1_lx = AUL_x; UA1_ux = AUU_x; }
  else { LA1_lx = ALU_x; LA1_ux = ALL_x; UA1_lx = AUU_x; UA1_ux = AUL_x; }
  double BLL_x = Tab[0], BLU_x = BLL_x + bA0_dot_B1, BUL_x = BLL_x + bA0_dot_B0, BUU_x = BLU_x + bA0_dot_B0;
  if (BLL_x < BLU_x) { LB1_lx = BLL_x; LB1_ux = BLU_x; UB1_lx = BUL_x; UB1_ux = BUU_x; }
  else { LB1_lx = BLU_x; LB1_ux = BLL_x; UB1_lx = BUU_x; UB1_ux = BUL_x; }

  // UA1, UB1 / UA1, LB1 / LA1, UB1 / LA1, LB1
  if (block(UA1_ux > b[0], UB1_ux > a[0], UA1_lx > b[0],
            {b[1], a[1], A1_dot_B0, aA0_dot_B0 - b[0] - Tba[0], A1_dot_B1, aA0_dot_B1 - Tba[1], -Tab[1] - bA1_dot_B0},
            UB1_lx > a[0], {a[1], b[1], A0_dot_B1, Tab[0] + bA0_dot_B0 - a[0], A1_dot_B1, Tab[1] + bA1_dot_B0, Tba[1] - aA0_dot_B1},
            {a[1], b[1], A1_dot_B1, Tab[1] + bA1_dot_B0, Tba[1] - aA0_dot_B1}, base(0, b[0], 0, a[0]), 1, 1))
    return result;
  if (block(UA1_lx < 0, LB1_ux > a[0], UA1_ux < 0,
            {b[1], a[1], -A1_dot_B0, Tba[0] - aA0_dot_B0, A1_dot_B1, aA0_dot_B1 - Tba[1], -Tab[1]}, LB1_lx > a[0],
            {a[1], b[1], A0_dot_B1, Tab[0] - a[0], A1_dot_B1, Tab[1], Tba[1] - aA0_dot_B1},
            {a[1], b[1], A1_dot_B1, Tab[1], Tba[1] - aA0_dot_B1}, base(-1, 0, 0, a[0]), 1, 1))
    return result;
  if (block(LA1_ux > b[0], UB1_lx < 0, LA1_lx > b[0],
            {b[1], a[1], A1_dot_B0, -Tba[0] - b[0], A1_dot_B1, -Tba[1], -Tab[1] - bA1_dot_B0}, UB1_ux < 0,
            {a[1], b[1], -A0_dot_B1, -Tab[0] - bA0_dot_B0, A1_dot_B1, Tab[1] + bA1_dot_B0, Tba[1]},
            {a[1], b[1], A1_dot_B1, Tab[1] + bA1_dot_B0, Tba[1]}, base(0, b[0], -1, 0), 1, 1))
    return result;
  if (block(LA1_lx < 0, LB1_lx < 0, LA1_ux < 0, {b[1], a[1], -A1_dot_B0, Tba[0], A1_dot_B1, -Tba[1], -Tab[1]},
            LB1_ux < 0, {a[1], b[1], -A0_dot_B1, -Tab[0], A1_dot_B1, Tab[1], Tba[1]},
            {a[1], b[1], A1_dot_B1, Tab[1], Tba[1]}, base(-1, 0, -1, 0), 1, 1))
    return result;

  double ALL_y = -Tba[1], ALU_y = ALL_y + aA1_dot_B1, AUL_y = ALL_y + aA0_dot_B1, AUU_y = ALU_y + aA0_dot_B1;
  double LA1_ly, LA1_uy, UA1_ly, UA1_uy, LB0_lx, LB0_ux, UB0_lx, UB0_ux;
  if (ALL_y < ALU_y) { LA1_ly = ALL_y; LA1_uy = ALU_y; UA1_ly = AUL_y; UA1_uy = AUU_y; }
  else { LA1_ly = ALU_y; LA1_uy = ALL_y; UA1_ly = AUU_y; UA1_uy = AUL_y; }
  if (BLL_x < BUL_x) { LB0_lx = BLL_x; LB0_ux = BUL_x; UB0_lx = BLU_x; UB0_ux = BUU_x; }
  else { LB0_lx = BUL_x; LB0_ux = BLL_x; UB0_lx = BUU_x; UB0_ux = BLU_x; }

  // UA1, UB0 / UA1, LB0 / LA1, UB0 / LA1, LB0
  if (block(UA1_uy > b[1], UB0_ux > a[0], UA1_ly > b[1],
            {b[0], a[1], A1_dot_B1, aA0_dot_B1 - Tba[1] - b[1], A1_dot_B0, aA0_dot_B0 - Tba[0], -Tab[1] - bA1_dot_B1},
            UB0_lx > a[0], {a[1], b[0], A0_dot_B0, Tab[0] - a[0] + bA0_dot_B1, A1_dot_B0, Tab[1] + bA1_dot_B1, Tba[0] - aA0_dot_B0},
            {a[1], b[0], A1_dot_B0, Tab[1] + bA1_dot_B1, Tba[0] - aA0_dot_B0}, base(1, b[1], 0, a[0]), 0, 1))
    return result;
  if (block(UA1_ly < 0, LB0_ux > a[0], UA1_uy < 0,
            {b[0], a[1], -A1_dot_B1, Tba[1] - aA0_dot_B1, A1_dot_B0, aA0_dot_B0 - Tba[0], -Tab[1]}, LB0_lx > a[0],
            {a[1], b[0], A0_dot_B0, Tab[0] - a[0], A1_dot_B0, Tab[1], Tba[0] - aA0_dot_B0},
            {a[1], b[0], A1_dot_B0, Tab[1], Tba[0] - aA0_dot_B0}, base(-1, 0, 0, a[0]), 0, 1))
    return result;
  if (block(LA1_uy > b[1], UB0_lx < 0, LA1_ly > b[1],
            {b[0], a[1], A1_dot_B1, -Tba[1] - b[1], A1_dot_B0, -Tba[0], -Tab[1] - bA1_dot_B1}, UB0_ux < 0,
            {a[1], b[0], -A0_dot_B0, -Tab[0] - bA0_dot_B1, A1_dot_B0, Tab[1] + bA1_dot_B1, Tba[0]},
            {a[1], b[0], A1_dot_B0, Tab[1] + bA1_dot_B1, Tba[0]}, base(1, b[1], -1, 0), 0, 1))
    return result;
  if (block(LA1_ly < 0, LB0_lx < 0, LA1_uy < 0, {b[0], a[1], -A1_dot_B1, Tba[1], A1_dot_B0, -Tba[0], -Tab[1]},
            LB0_ux < 0, {a[1], b[0], -A0_dot_B0, -Tab[0], A1_dot_B0, Tab[1], Tba[0]},
            {a[1], b[0], A1_dot_B0, Tab[1], Tba[0]}, base(-1, 0, -1, 0), 0, 1))
    return result;

  double BLL_y = Tab[1], BLU_y = BLL_y + bA1_dot_B1, BUL_y = BLL_y + bA1_dot_B0, BUU_y = BLU_y + bA1_dot_B0;
  double LA0_lx, LA0_ux, UA0_lx, UA0_ux, LB1_ly, LB1_uy, UB1_ly, UB1_uy;
  if (ALL_x < AUL_x) { LA0_lx = ALL_x; LA0_ux = AUL_x; UA0_lx = ALU_x; UA0_ux = AUU_x; }
  else { LA0_lx = AUL_x; LA0_ux = ALL_x; UA0_lx = AUU_x; UA0_ux = ALU_x; }
  if (BLL_y < BLU_y) { LB1_ly = BLL_y; LB1_uy = BLU_y; UB1_ly = BUL_y; UB1_uy = BUU_y; }
  else { LB1_ly = BLU_y; LB1_uy = BLL_y; UB1_ly = BUU_y; UB1_uy = BUL_y; }

  // UA0, UB1 / UA0, LB1 / LA0, UB1 / LA0, LB1
  if (block(UA0_ux > b[0], UB1_uy > a[1], UA0_lx > b[0],
            {b[1], a[0], A0_dot_B0, aA1_dot_B0 - Tba[0] - b[0], A0_dot_B1, aA1_dot_B1 - Tba[1], -Tab[0] - bA0_dot_B0},
            UB1_ly > a[1], {a[0], b[1], A1_dot_B1, Tab[1] - a[1] + bA1_dot_B0, A0_dot_B1, Tab[0] + bA0_dot_B0, Tba[1] - aA1_dot_B1},
            {a[0], b[1], A0_dot_B1, Tab[0] + bA0_dot_B0, Tba[1] - aA1_dot_B1}, base(0, b[0], 1, a[1]), 1, 0))
    return result;
  if (block(UA0_lx < 0, LB1_uy > a[1], UA0_ux < 0,
            {b[1], a[0], -A0_dot_B0, Tba[0] - aA1_dot_B0, A0_dot_B1, aA1_dot_B1 - Tba[1], -Tab[0]}, LB1_ly > a[1],
            {a[0], b[1], A1_dot_B1, Tab[1] - a[1], A0_dot_B1, Tab[0], Tba[1] - aA1_dot_B1},
            {a[0], b[1], A0_dot_B1, Tab[0], Tba[1] - aA1_dot_B1}, base(-1, 0, 1, a[1]), 1, 0))
    return result;
  if (block(LA0_ux > b[0], UB1_ly < 0, LA0_lx > b[0],
            {b[1], a[0], A0_dot_B0, -b[0] - Tba[0], A0_dot_B1, -Tba[1], -bA0_dot_B0 - Tab[0]}, UB1_uy < 0,
            {a[0], b[1], -A1_dot_B1, -Tab[1] - bA1_dot_B0, A0_dot_B1, Tab[0] + bA0_dot_B0, Tba[1]},
            {a[0], b[1], A0_dot_B1, Tab[0] + bA0_dot_B0, Tba[1]}, base(0, b[0], -1, 0), 1, 0))
    return result;
  if (block(LA0_lx < 0, LB1_ly < 0, LA0_ux < 0, {b[1], a[0], -A0_dot_B0, Tba[0], A0_dot_B1, -Tba[1], -Tab[0]},
            LB1_uy < 0, {a[0], b[1], -A1_dot_B1, -Tab[1], A0_dot_B1, Tab[0], Tba[1]},
            {a[0], b[1], A0_dot_B1, Tab[0], Tba[1]}, base(-1, 0, -1, 0), 1, 0))
    return result;

  double LA0_ly, LA0_uy, UA0_ly, UA0_uy, LB0_ly, LB0_uy, UB0_ly, UB0_uy;
  if (ALL_y < AUL_y) { LA0_ly = ALL_y; LA0_uy = AUL_y; UA0_ly = ALU_y; UA0_uy = AUU_y; }
  else { LA0_ly = AUL_y; LA0_uy = ALL_y; UA0_ly = AUU_y; UA0_uy = ALU_y; }
  if (BLL_y < BUL_y) { LB0_ly = BLL_y; LB0_uy = BUL_y; UB0_ly = BLU_y; UB0_uy = BUU_y; }
  else { LB0_ly = BUL_y; LB0_uy = BLL_y; UB0_ly = BUU_y; UB0_uy = BLU_y; }

  // UA0, UB0 / UA0, LB0 / LA0, UB0 / LA0, LB0
  if (block(UA0_uy > b[1], UB0_uy > a[1], UA0_ly > b[1],
            {b[0], a[0], A0_dot_B1, aA1_dot_B1 - Tba[1] - b[1], A0_dot_B0, aA1_dot_B0 - Tba[0], -Tab[0] - bA0_dot_B1},
            UB0_ly > a[1], {a[0], b[0], A1_dot_B0, Tab[1] - a[1] + bA1_dot_B1, A0_dot_B0, Tab[0] + bA0_dot_B1, Tba[0] - aA1_dot_B0},
            {a[0], b[0], A0_dot_B0, Tab[0] + bA0_dot_B1, Tba[0] - aA1_dot_B0}, base(1, b[1], 1, a[1]), 0, 0))
    return result;
  if (block(UA0_ly < 0, LB0_uy > a[1], UA0_uy < 0,
            {b[0], a[0], -A0_dot_B1, Tba[1] - aA1_dot_B1, A0_dot_B0, aA1_dot_B0 - Tba[0], -Tab[0]}, LB0_ly > a[1],
            {a[0], b[0], A1_dot_B0, Tab[1] - a[1], A0_dot_B0, Tab[0], Tba[0] - aA1_dot_B0},
            {a[0], b[0], A0_dot_B0, Tab[0], Tba[0] - aA1_dot_B0}, base(-1, 0, 1, a[1]), 0, 0))
    return result;
  if (block(LA0_uy > b[1], UB0_ly < 0, LA0_ly > b[1],
            {b[0], a[0], A0_dot_B1, -Tba[1] - b[1], A0_dot_B0, -Tba[0], -Tab[0] - bA0_dot_B1}, UB0_uy < 0,
            {a[0], b[0], -A1_dot_B0, -Tab[1] - bA1_dot_B1, A0_dot_B0, Tab[0] + bA0_dot_B1, Tba[0]},
            {a[0], b[0], A0_dot_B0, Tab[0] + bA0_dot_B1, Tba[0]}, base(1, b[1], -1, 0), 0, 0))
    return result;
  if (block(LA0_ly < 0, LB0_ly < 0, LA0_uy < 0, {b[0], a[0], -A0_dot_B1, Tba[1], A0_dot_B0, -Tba[0], -Tab[0]},
            LB0_uy < 0, {a[0], b[0], -A1_dot_B0, -Tab[1], A0_dot_B0, Tab[0], Tba[0]},
            {a[0], b[0], A0_dot_B0, Tab[0], Tba[0]}, base(-1, 0, -1, 0), 0, 0))
    return result;

  // no edges passed: max separation along the face normals (:654-712)
  double sep1, sep2;
  if (Tab[2] > 0.0) {
    sep1 = Tab[2];
    if (Rab.m[2][0] < 0.0) sep1 += b[0] * Rab.m[2][0];
    if (Rab.m[2][1] < 0.0) sep1 += b[1] * Rab.m[2][1];
  } else {
    sep1 = -Tab[2];
    if (Rab.m[2][0] > 0.0) sep1 -= b[0] * Rab.m[2][0];
    if (Rab.m[2][1] > 0.0) sep1 -= b[1] * Rab.m[2][1];
  }
  if (Tba[2] < 0) {
    sep2 = -Tba[2];
    if (Rab.m[0][2] < 0.0) sep2 += a[0] * Rab.m[0][2];
    if (Rab.m[1][2] < 0.0) sep2 += a[1] * Rab.m[1][2];
  } else {
    sep2 = Tba[2];
    if (Rab.m[0][2] > 0.0) sep2 -= a[0] * Rab.m[0][2];
    if (Rab.m[1][2] > 0.0) sep2 -= a[1] * Rab.m[1][2];
  }
  const double sep = (sep1 > sep2 ? sep1 : sep2);
  return (sep > 0 ? sep : 0);
}

// distance(R0, T0, b1.rss, b2.rss), RSS.cpp:995-1005
double rss_distance(const M3& R0, const V3& T0, const hfcl_bvh_node& b1, const hfcl_bvh_node& b2) {
  const M3 A1 = axes_of(b1.rss_axes), A2 = axes_of(b2.rss_axes);
  const M3 R = tmul(A1, R0 * A2);
  const V3 Ttemp = R0 * v3(b2.rss_Tr) + T0 - v3(b1.rss_Tr);
  const V3 T = tmul(A1, Ttemp);
  double dist = rect_distance(R, T, b1.rss_length, b2.rss_length);
  dist -= (b1.rss_radius + b2.rss_radius);
  return (dist < 0.0) ? 0.0 : dist;
}

// TriangleDistance::segPoints, src/intersect.cpp:60-154
static void seg_points(const V3& P, const V3& A, const V3& Q, const V3& B, V3& VEC, V3& X, V3& Y) {
  V3 T = Q - P, TMP;
  const double A_dot_A = dot(A, A), B_dot_B = dot(B, B), A_dot_B = dot(A, B), A_dot_T = dot(A, T), B_dot_T = dot(B, T);
  double t, u;
  const double denom = A_dot_A * B_dot_B - A_dot_B * A_dot_B;
  t = (A_dot_T * B_dot_B - B_dot_T * A_dot_B) / denom;
  if ((t < 0) || std::isnan(t))
    t = 0;
  else if (t > 1)
    t = 1;
  u = (t * A_dot_B - B_dot_T) / B_dot_B;
  if ((u <= 0) || std::isnan(u)) {
    Y = Q;
    t = A_dot_T / A_dot_A;
    if ((t <= 0) || std::isnan(t)) {
      X = P;
      VEC = Q - P;
    } else if (t >= 1) {
      X = P + A;
      VEC = Q - X;
    } else {
      X = P + A * t;
      TMP = cross(T, A);
      VEC = cross(A, TMP);
    }
  } else if (u >= 1) {
    Y = Q + B;
    t = (A_dot_B + A_dot_T) / A_dot_A;
    if ((t <= 0) || std::isnan(t)) {
      X = P;
      VEC = Y - P;
    } else if (t >= 1) {
      X = P + A;
      VEC = Y - X;
    } else {
      X = P + A * t;
      T = Y - P;
      TMP = cross(T, A);
      VEC = cross(A, TMP);
    }
  } else {
    Y = Q + B * u;
    if ((t <= 0) || std::isnan(t)) {
      X = P;
      TMP = cross(T, B);
      VEC = cross(B, TMP);
    } else if (t >= 1) {
      X = P + A;
      T = Q - X;
      TMP = cross(T, B);
      VEC = cross(B, TMP);
    } else {
      X = P + A * t;
      VEC = cross(A, B);
      if (dot(VEC, T) < 0) VEC = VEC * (-1);
    }
  }
}

// TriangleDistance::sqrTriDistance, src/intersect.cpp:156-368
double sqr_tri_distance(const V3 S[3], const V3 T[3], V3& P, V3& Q) {
  V3 Sv[3] = {S[1] - S[0], S[2] - S[1], S[0] - S[2]};
  V3 Tv[3] = {T[1] - T[0], T[2] - T[1], T[0] - T[2]};
  V3 VEC, V, Z, minP, minQ;
  int shown_disjoint = 0;
  double mindd = sqnorm(S[0] - T[0]) + 1;
  for (int i = 0; i < 3; ++i) {
    for (int j = 0; j < 3; ++j) {
      seg_points(S[i], Sv[i], T[j], Tv[j], VEC, P, Q);
      V = Q - P;
      const double dd = dot(V, V);
      if (dd <= mindd) {
        minP = P;
        minQ = Q;
        mindd = dd;
        Z = S[(i + 2) % 3] - P;
        double a = dot(Z, VEC);
        Z = T[(j + 2) % 3] - Q;
        double b = dot(Z, VEC);
        if ((a <= 0) && (b >= 0)) return dd;
        const double p = dot(V, VEC);
        if (a < 0) a = 0;
        if (b > 0) b = 0;
        if ((p - a + b) > 0) shown_disjoint = 1;
      }
    }
  }
  const V3 Sn = cross(Sv[0], Sv[1]);
  const double Snl = dot(Sn, Sn);
  if (Snl > 1e-15) {
    double Tp[3] = {dot(S[0] - T[0], Sn), dot(S[0] - T[1], Sn), dot(S[0] - T[2], Sn)};
    int point = -1;
    if ((Tp[0] > 0) && (Tp[1] > 0) && (Tp[2] > 0)) {
      point = (Tp[0] < Tp[1]) ? 0 : 1;
      if (Tp[2] < Tp[point]) point = 2;
    } else if ((Tp[0] < 0) && (Tp[1] < 0) && (Tp[2] < 0)) {
      point = (Tp[0] > Tp[1]) ? 0 : 1;
      if (Tp[2] > Tp[point]) point = 2;
    }
    if (point >= 0) {
      shown_disjoint = 1;
      if (dot(T[point] - S[0], cross(Sn, Sv[0])) > 0 && dot(T[point] - S[1], cross(Sn, Sv[1])) > 0 &&
          dot(T[point] - S[2], cross(Sn, Sv[2])) > 0) {
        P = T[point] + Sn * (Tp[point] / Snl);
        Q = T[point];
        return sqnorm(P - Q);
      }
    }
  }
  const V3 Tn = cross(Tv[0], Tv[1]);
  const double Tnl = dot(Tn, Tn);
  if (Tnl > 1e-15) {
    double Sp[3] = {dot(T[0] - S[0], Tn), dot(T[0] - S[1], Tn), dot(T[0] - S[2], Tn)};
    int point = -1;
    if ((Sp[0] > 0) && (Sp[1] > 0) && (Sp[2] > 0)) {
      point = (Sp[0] < Sp[1]) ? 0 : 1;
      if (Sp[2] < Sp[point]) point = 2;
    } else if ((Sp[0] < 0) && (Sp[1] < 0) && (Sp[2] < 0)) {
      point = (Sp[0] > Sp[1]) ? 0 : 1;
      if (Sp[2] > Sp[point]) point = 2;
    }
    if (point >= 0) {
      shown_disjoint = 1;
      if (dot(S[point] - T[0], cross(Tn, Tv[0])) > 0 && dot(S[point] - T[1], cross(Tn, Tv[1])) > 0 &&
          dot(S[point] - T[2], cross(Tn, Tv[2])) > 0) {
        P = S[point];
        Q = S[point] + Tn * (Sp[point] / Tnl);
        return sqnorm(P - Q);
      }
    }
  }
  if (shown_disjoint) {
    P = minP;
    Q = minQ;
    return mindd;
  }
  return 0;
}

namespace {
struct DistTraversal {
  const MeshView& m1;
  const MeshView& m2;
  Tf tf1;
  M3 RT_R;
  V3 RT_T;
  double min_distance = std::numeric_limits<double>::max();
  int b1 = -1, b2 = -1;
  V3 np1 = nan3(), np2 = nan3();
  BvhStats stats;
  DistTraversal(const MeshView& a, const Tf& t1, const MeshView& b, const Tf& t2) : m1(a), m2(b), tf1(t1) {
    RT_R = tmul(t1.R, t2.R);  // relativeTransform, tools.h:91-99
    RT_T = tmul(t1.R, t2.T - t1.T);
  }
  void leaf(int pid1, int pid2) {  // leafComputeDistance / preprocess, traversal_node_bvhs.h:433-467,490-514
    V3 S[3], T[3];
    for (int k = 0; k < 3; ++k) {
      const double* p = m1.verts + 3 * size_t(m1.tris[3 * pid1 + k]);
      const double* q = m2.verts + 3 * size_t(m2.tris[3 * pid2 + k]);
      S[k] = V3(p[0], p[1], p[2]);
      T[k] = RT_R * V3(q[0], q[1], q[2]) + RT_T;
    }
    V3 P1, P2;
    const double d = std::sqrt(sqr_tri_distance(S, T, P1, P2));
    if (min_distance > d) {  // DistanceResult::update
      min_distance = d;
      b1 = pid1;
      b2 = pid2;
      np1 = P1;
      np2 = P2;
    }
  }
  // canStop with the node's rel_err = abs_err = 0: they are latched from a default-constructed
  // request in the node's constructor and never refreshed by initialize() (traversal_node_bvhs.h:409-410)
  bool can_stop(double c) const { return (c >= min_distance) && (c >= min_distance); }
  double lower_bound(unsigned a, unsigned b) {
    ++stats.num_bv_tests;
    return rss_distance(RT_R, RT_T, m1.nodes[a], m2.nodes[b]);
  }
  void recurse(unsigned n1i, unsigned n2i) {  // distanceRecurse, traversal_recurse.cpp:153-203
    const hfcl_bvh_node& n1 = m1.nodes[n1i];
    const hfcl_bvh_node& n2 = m2.nodes[n2i];
    const bool l1 = n1.first_child < 0, l2 = n2.first_child < 0;
    if (l1 && l2) {
      ++stats.num_leaf_tests;
      leaf(-(n1.first_child + 1), -(n2.first_child + 1));
      return;
    }
    const double s1 = n1.obb_extent[0] * n1.obb_extent[0] + n1.obb_extent[1] * n1.obb_extent[1] + n1.obb_extent[2] * n1.obb_extent[2];
    const double s2 = n2.obb_extent[0] * n2.obb_extent[0] + n2.obb_extent[1] * n2.obb_extent[1] + n2.obb_extent[2] * n2.obb_extent[2];
    unsigned a1, a2, c1, c2;
    if (l2 || (!l1 && (s1 > s2))) {
      a1 = unsigned(n1.first_child);
      a2 = n2i;
      c1 = a1 + 1;
      c2 = n2i;
    } else {
      a1 = n1i;
      a2 = unsigned(n2.first_child);
      c1 = n1i;
      c2 = a2 + 1;
    }
    const double d1 = lower_bound(a1, a2), d2 = lower_bound(c1, c2);
    if (d2 < d1) {
      if (!can_stop(d2)) recurse(c1, c2);
      if (!can_stop(d1)) recurse(a1, a2);
    } else {
      if (!can_stop(d1)) recurse(a1, a2);
      if (!can_stop(d2)) recurse(c1, c2);
    }
  }
};
}  // namespace

// the distance leafComputeDistance assigns to the triangle pair (pid1, pid2) of this query (for tests that enumerate ties)
double bvh_leaf_distance(const MeshView& m1, const Tf& tf1, const MeshView& m2, const Tf& tf2, int pid1, int pid2) {
  DistTraversal t(m1, tf1, m2, tf2);
  t.leaf(pid1, pid2);
  return t.min_distance;
}

int bvh_distance_pair(const MeshView& m1, const Tf& tf1, const MeshView& m2, const Tf& tf2, hfcl_result& out,
                      BvhStats* stats) {
  DistTraversal t(m1, tf1, m2, tf2);
  t.leaf(0, 0);  // preprocess(): seeds min_distance with triangle 0 x triangle 0
  t.recurse(0, 0);
  const double nanv = std::numeric_limits<double>::quiet_NaN();
  // postprocess(): nearest points from model-1 frame to world
  const V3 w1 = tf1.transform(t.np1), w2 = tf1.transform(t.np2);
  out.distance = t.min_distance;
  for (int k = 0; k < 3; ++k) {
    out.normal[k] = nanv;  // left uninitialised by the reference on this path (traversal_node_bvhs.h:454,465)
    out.p1[k] = w1[k];
    out.p2[k] = w2[k];
  }
  out.b1 = t.b1;
  out.b2 = t.b2;
  out.status = (t.min_distance <= 0) ? 128u : 0u;
  out.num_contacts = 0;
  if (stats) *stats = t.stats;
  return HFCL_OK;
}

}  // namespace orc
