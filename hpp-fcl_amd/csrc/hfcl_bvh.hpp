// hfcl_bvh.hpp -- BVHModel<OBBRSS> x BVHModel<OBBRSS> collide(): BV test and leaf test.
//
// Behavioural contract (reference file:line):
//   OBB SAT with lower bound   src/BV/OBB.cpp:290-393 (obbDisjointAndLowerBoundDistance), :475-483 (overlap)
//   leaf = triangle-triangle   include/hpp/fcl/internal/traversal_node_bvhs.h:184-233 (leafCollides)
//                              src/distance/triangle_triangle.cpp:46-105 (vanilla GJK on world-frame
//                              triangles, centroid guess, no EPA), src/narrowphase/details.h:699-709
//   triangle support           src/narrowphase/support_functions.cpp:110-134
#pragma once
// HFCL_LEAF_CONTRACT_OFF (the collide() part of the mesh unit, Makefile): the leaves' solvers -- GJK, EPA, the supports, everything of
// hfcl_pair.hpp and the leaf functions of hfcl_bvh_shape.hpp -- are compiled without contraction of a*b+c, the box tests around them
// with hipcc's default.  A leaf is called from several kernels (the lanes' walk, the waves' continuation, the task levels), each with
// its own inlined copy; contracted, two copies may fuse differently and a GJK run that stops an iteration apart reports a normal
// 1e-3 away (cylinder caps; profiles/r06_b) -- without, every copy is the same arithmetic, and the oracle's.
#if defined(HFCL_LEAF_CONTRACT_OFF)
#pragma clang fp contract(off)
#endif
#include "hfcl_pair.hpp"
#if defined(HFCL_LEAF_CONTRACT_OFF)
#pragma clang fp contract(fast)
#endif
#if !defined(__HIP_DEVICE_COMPILE__)
#include <algorithm>
#include <utility>
#include <vector>
#endif

namespace hfcl {

// RSS part of a node (RSS.h:54-150); its axes are the OBB's (BVFitter<OBBRSS>::fit sets
// rss.axes = obb.axes, BV_fitter.cpp:513).  fp32: 24 B, fp64: 48 B; read by the distance traversal only.
template <typename T>
struct DRss {
  V3<T> Tr;
  T l0, l1, r;
};

// Device node: what the collide traversal reads of BVNode<OBBRSS> (BV_node.h:52-148, OBB.h:52-126).
// fp32: 64 B, fp64: 128 B.
template <typename T>
struct DNode {
  int32_t first_child;  // >0 children at first_child, first_child+1 ; <0 leaf, primitive = -(first_child+1)
  int32_t pad_;
  M3<T> axes;           // rows of the OBB axes matrix (columns = axes)
  V3<T> To;
  V3<T> extent;
};

// What the distance() walk reads of a node, in ONE record (fp64: 128 B = one cache line, against a 128-B DNode of which it
// uses the axes and the child link plus a 48-B DRss from another array): the axes (RSS axes = OBB axes), the RSS origin,
// side lengths and radius, the child link, and the rank of obb.extent.squaredNorm() among all nodes of the library
// (obbf_size_ranks: equal sizes share a rank, so rank1 > rank2 <=> size1 > size2 -- the descent rule of distanceRecurse,
// traversal_recurse.cpp:165-166 via firstOverSecond, as an integer compare on values formed without contraction).
template <typename T>
struct alignas(16) DNodeD {
  M3<T> axes;
  V3<T> Tr;
  T l0, l1, r;
  int32_t first_child;
  uint32_t rank;
};

// obbDisjointAndLowerBoundDistance: B, T = pose of OBB 2 in the frame of OBB 1; a, b = extents.
template <typename T>
HFCL_HD bool obb_disjoint_lb(const M3<T>& B, const V3<T>& Tv, const V3<T>& a_, const V3<T>& b_, T security_margin,
                             T break_distance2, T& sq) {
  const V3<T> a = mk<T>(hmax(a_.x + security_margin / T(2), T(0)), hmax(a_.y + security_margin / T(2), T(0)),
                        hmax(a_.z + security_margin / T(2), T(0)));
  const V3<T> b = mk<T>(hmax(b_.x + security_margin / T(2), T(0)), hmax(b_.y + security_margin / T(2), T(0)),
                        hmax(b_.z + security_margin / T(2), T(0)));
  M3<T> Bf;
  Bf.r0 = mk<T>(habs(B.r0.x), habs(B.r0.y), habs(B.r0.z));
  Bf.r1 = mk<T>(habs(B.r1.x), habs(B.r1.y), habs(B.r1.z));
  Bf.r2 = mk<T>(habs(B.r2.x), habs(B.r2.y), habs(B.r2.z));
  {  // A axes
    const V3<T> Bfb = mul(Bf, b);
    const T cx = hmax(habs(Tv.x) - a.x - Bfb.x, T(0)), cy = hmax(habs(Tv.y) - a.y - Bfb.y, T(0)),
            cz = hmax(habs(Tv.z) - a.z - Bfb.z, T(0));
    sq = cx * cx + cy * cy + cz * cz;
  }
  if (sq > break_distance2) return true;
  {  // B axes
    T t = T(0), s;
    s = habs(dot(col(B, 0), Tv)) - dot(col(Bf, 0), a) - b.x;
    if (s > T(0)) t += s * s;
    s = habs(dot(col(B, 1), Tv)) - dot(col(Bf, 1), a) - b.y;
    if (s > T(0)) t += s * s;
    s = habs(dot(col(B, 2), Tv)) - dot(col(Bf, 2), a) - b.z;
    if (s > T(0)) t += s * s;
    sq = t;
  }
  if (sq > break_distance2) return true;
  // Ai x Bj, (ia, ja, ka) = (0,1,2), (1,2,0), (2,0,1)
  const T Tc[3] = {Tv.x, Tv.y, Tv.z};
  const T av[3] = {a.x, a.y, a.z};
  const T bv[3] = {b.x, b.y, b.z};
  const T Bm[3][3] = {{B.r0.x, B.r0.y, B.r0.z}, {B.r1.x, B.r1.y, B.r1.z}, {B.r2.x, B.r2.y, B.r2.z}};
#pragma unroll
  for (int ia = 0; ia < 3; ++ia) {
    const int ja = (ia + 1) % 3, ka = (ia + 2) % 3;
#pragma unroll
    for (int ib = 0; ib < 3; ++ib) {
      const int jb = (ib + 1) % 3, kb = (ib + 2) % 3;
      const T f_ia_ib = habs(Bm[ia][ib]);
      const T sinus2 = T(1) - f_ia_ib * f_ia_ib;
      if (!(sinus2 < T(1e-6))) {
        const T s = Tc[ka] * Bm[ja][ib] - Tc[ja] * Bm[ka][ib];
        const T diff = habs(s) - (av[ja] * habs(Bm[ka][ib]) + av[ka] * habs(Bm[ja][ib]) + bv[jb] * habs(Bm[ia][kb]) +
                                  bv[kb] * habs(Bm[ia][jb]));
        if (diff > T(0)) {
          sq = diff * diff / sinus2;
          if (sq > break_distance2) return true;
        }
      }
    }
  }
  return false;
}

// overlap(R0, T0, b1, b2, request, sq) of OBB.cpp:475-483; returns DISJOINT.
template <typename T>
HFCL_HD bool obb_disjoint(const M3<T>& R0, const V3<T>& T0, const DNode<T>& b1, const DNode<T>& b2, T security_margin,
                          T break_distance2, T& sq) {
  const V3<T> Ttemp = tmul(R0, b2.To - T0) - b1.To;
  const V3<T> Tv = tmul(b1.axes, Ttemp);
  const M3<T> R = tmul(b1.axes, tmul(R0, b2.axes));  // b1.axes^T * R0^T * b2.axes
  return obb_disjoint_lb(R, Tv, b1.extent, b2.extent, security_margin, break_distance2, sq);
}

// ---------------------------------------------------------------------------------------
// fp32 FILTER in front of the fp64 separating-axis test (mesh x mesh collide()).
//
// A step of the traversal is a chain of ~300 dependent fp64 instructions on two 128-byte node records.  Nearly all of
// those tests are decided far from any threshold, so they are evaluated in fp32 on 64-byte records with a rigorous bound
// on what rounding can have done to every tested quantity, and the fp64 test (obb_disjoint above: the reference's
// arithmetic) runs only where the fp32 result does not *prove* what the fp64 test would do:
//   OBBF_OVERLAP   the fp64 test returns "not disjoint" (no check point exceeds break_distance^2): push the children;
//   OBBF_DISJOINT  the fp64 test returns "disjoint" and the value it reports satisfies nd_lo <= sqrt(sq) <= nd_hi.  The
//                  caller skips the fp64 test if nd_lo >= its running lower bound (nothing would change); otherwise the pair
//                  is a candidate for the minimum, whose exact value is computed when something is compared with it;
//   OBBF_UNSURE    anything else (a tested quantity within the error bound of its threshold, nearly parallel edge axes,
//                  a node whose third axis is not the cross product of the first two): run the fp64 test.
// So every decision of the walk, and every number that reaches a record, is the fp64 test's.
//
// Error bound (u = 2^-24; S = |To1|_1 + |To2|_1 + |T0|_1, Eb = |a|_1 + |b|_1 + 3|margin|; rotations have unit rows):
//   stored axes: u per entry, third axis = +-(a0 x a1): <= 8u;   M = R0^T A2: <= 30u;   R = A1^T M: <= 72u per entry
//   Ttemp = R0^T (To2 - T0) - To1: <= 8u S;   Tv = A1^T Ttemp: <= 32u S
//   A-axis value |Tv_i| - a_i - sum_j |R_ij| b_j:  <= 40u S + 80u Eb
//   B-axis value |col_j(R) . Tv| - col_j(|R|) . a - b_j:  <= 190u S + 80u Eb
//   edge value |Tv_k R_ji - Tv_j R_ki| - (four extent x |R| products):  <= 320u S + 80u Eb;   1 - R_ij^2: <= 150u
// One bound is used for all of them: delta = 384u (S + Eb) (~2.3e-5 per unit of scene size: 20-100x what rounding
// actually does, tests/test_device_core_hostsim.py measures both), and OBBF_DELTA_SIN = 256u for the sine terms.
// ---------------------------------------------------------------------------------------
struct DNodeF {         // 64 bytes
  int32_t first_child;  // as DNode
  uint32_t rank;        // bits 0-29: rank of extent.squaredNorm() (fp64, ties share a rank) among ALL nodes of the library --
                        // firstOverSecond's size comparison (traversal_node_bvhs.h:160-170) as an exact integer compare;
                        // bit 31: third axis = -(a0 x a1);  bit 30: never trust the filter on this node
  float a0[3], a1[3];   // OBB axes 0 and 1 (columns of obb.axes)
  float To[3];
  float extent[3];
  float mag;            // >= |To|_1 + |extent|_1 (rounded up)
  float pad_;
};
constexpr uint32_t OBBF_RANK_MASK = 0x3FFFFFFFu, OBBF_UNSAFE = 0x40000000u, OBBF_LEFT = 0x80000000u;
enum { OBBF_OVERLAP = 0, OBBF_DISJOINT = 1, OBBF_UNSURE = 2 };
constexpr float OBBF_U = 5.9604645e-8f;  // 2^-24
constexpr float OBBF_DELTA = 384.f * OBBF_U, OBBF_DELTA_SIN = 256.f * OBBF_U;

// host side: the filter record of a reference node
inline DNodeF pack_fnode(const hfcl_bvh_node& n, uint32_t rank) {
  DNodeF f;
  f.first_child = n.first_child;
  const double* a = n.obb_axes;  // column-major: columns = axes
  const double c[3] = {a[1] * a[5] - a[2] * a[4], a[2] * a[3] - a[0] * a[5], a[0] * a[4] - a[1] * a[3]};  // a0 x a1
  const double dp = c[0] * a[6] + c[1] * a[7] + c[2] * a[8];
  const double sgn = dp < 0 ? -1.0 : 1.0;
  bool safe = true;
  for (int k = 0; k < 3; ++k) safe = safe && fabs(sgn * c[k] - a[6 + k]) <= 1e-9;  // orthonormal to well below fp32 rounding
  for (int k = 0; k < 9; ++k) safe = safe && fabs(a[k]) <= 1.0 + 1e-9;
  const double l0 = a[0] * a[0] + a[1] * a[1] + a[2] * a[2], l1 = a[3] * a[3] + a[4] * a[4] + a[5] * a[5];
  safe = safe && fabs(l0 - 1.0) <= 1e-9 && fabs(l1 - 1.0) <= 1e-9 && fabs(a[0] * a[3] + a[1] * a[4] + a[2] * a[5]) <= 1e-9;
  for (int k = 0; k < 3; ++k) safe = safe && n.obb_extent[k] >= 0;  // (the error bound was derived for non-negative extents)
  f.rank = (rank & OBBF_RANK_MASK) | (sgn < 0 ? OBBF_LEFT : 0u) | (safe ? 0u : OBBF_UNSAFE);
  double mag = 0;
  for (int k = 0; k < 3; ++k) {
    f.a0[k] = float(a[k]);
    f.a1[k] = float(a[3 + k]);
    f.To[k] = float(n.obb_To[k]);
    f.extent[k] = float(n.obb_extent[k]);
    mag += fabs(n.obb_To[k]) + fabs(n.obb_extent[k]);
  }
  if (!(mag < 1e30)) f.rank |= OBBF_UNSAFE;  // NaN / huge: the fp64 test decides
  f.mag = float(mag) * (1.f + 4.f * OBBF_U) + 1e-37f;
  f.pad_ = 0.f;
  return f;
}

// host side: rank of every node's size (extent.squaredNorm() in fp64, evaluated without contraction like the reference's
// default build) among all n nodes; equal sizes share a rank, so rank1 > rank2  <=>  size1 > size2
inline void obbf_size_ranks(const hfcl_bvh_node* nodes, size_t n, uint32_t* rank_out) {
  std::vector<std::pair<double, uint32_t>> key(n);
  for (size_t i = 0; i < n; ++i) {
    const volatile double x = nodes[i].obb_extent[0] * nodes[i].obb_extent[0], y = nodes[i].obb_extent[1] * nodes[i].obb_extent[1],
                          z = nodes[i].obb_extent[2] * nodes[i].obb_extent[2];  // (volatile: no fused multiply-add)
    const double sz = (x + y) + z;
    key[i] = std::make_pair(sz == sz ? sz : 1.7976931348623157e+308, uint32_t(i));  // (NaN would break the sort's ordering)
  }
  std::sort(key.begin(), key.end());
  uint32_t r = 0;
  for (size_t i = 0; i < n; ++i) {
    if (i > 0 && key[i].first != key[i - 1].first) ++r;
    rank_out[key[i].second] = r;
  }
}

// The filter.  Same argument roles as obb_disjoint: (R0, T0) = pose of OBB 2's frame in OBB 1's frame, b1 / b2 the nodes;
// t0mag >= |T0|_1; margin = request.security_margin (>= 0 expected; any value is handled), bd2 = break_distance^2.
HFCL_HD int obb_filter(const M3<float>& R0, const V3<float>& T0, float t0mag, const DNodeF& b1, const DNodeF& b2, float margin,
                       float bd2, float& nd_lo, float& nd_hi) {
  nd_lo = 0.f;
  nd_hi = 3.402823466e+38f;
  if ((b1.rank | b2.rank) & OBBF_UNSAFE) return OBBF_UNSURE;
  // axes as rows of A^T: A1[k] = axis k of b1 (a vector); third axis from the first two
  const V3<float> p0 = mk<float>(b1.a0[0], b1.a0[1], b1.a0[2]), p1 = mk<float>(b1.a1[0], b1.a1[1], b1.a1[2]);
  const V3<float> q0 = mk<float>(b2.a0[0], b2.a0[1], b2.a0[2]), q1 = mk<float>(b2.a1[0], b2.a1[1], b2.a1[2]);
  const V3<float> p2 = (b1.rank & OBBF_LEFT) ? cross(p1, p0) : cross(p0, p1);
  const V3<float> q2 = (b2.rank & OBBF_LEFT) ? cross(q1, q0) : cross(q0, q1);
  // M_k = R0^T q_k (axis k of b2 in OBB-1-parent frame); R_ij = p_i . M_j
  const V3<float> m0 = tmul(R0, q0), m1 = tmul(R0, q1), m2 = tmul(R0, q2);
  const float R[3][3] = {{dot(p0, m0), dot(p0, m1), dot(p0, m2)}, {dot(p1, m0), dot(p1, m1), dot(p1, m2)}, {dot(p2, m0), dot(p2, m1), dot(p2, m2)}};
  const V3<float> To2 = mk<float>(b2.To[0], b2.To[1], b2.To[2]), To1 = mk<float>(b1.To[0], b1.To[1], b1.To[2]);
  const V3<float> Ttemp = tmul(R0, To2 - T0) - To1;
  const float Tc[3] = {dot(p0, Ttemp), dot(p1, Ttemp), dot(p2, Ttemp)};
  const float hm = 0.5f * margin;
  const float av[3] = {hmax(b1.extent[0] + hm, 0.f), hmax(b1.extent[1] + hm, 0.f), hmax(b1.extent[2] + hm, 0.f)};
  const float bv[3] = {hmax(b2.extent[0] + hm, 0.f), hmax(b2.extent[1] + hm, 0.f), hmax(b2.extent[2] + hm, 0.f)};
  const float delta = OBBF_DELTA * (b1.mag + b2.mag + t0mag + 3.f * habs(margin));
  const float bd_hi = bd2 * (1.f + 16.f * OBBF_U), bd_lo = bd2 * (1.f - 16.f * OBBF_U);
  float F[3][3];
#pragma unroll
  for (int i = 0; i < 3; ++i)
#pragma unroll
    for (int j = 0; j < 3; ++j) F[i][j] = habs(R[i][j]);
  // ---- A axes: sq = sum_i max(|Tc_i| - a_i - sum_j F_ij b_j, 0)^2
  {
    float lo = 0.f, hi = 0.f;
#pragma unroll
    for (int i = 0; i < 3; ++i) {
      const float v = habs(Tc[i]) - av[i] - (F[i][0] * bv[0] + F[i][1] * bv[1] + F[i][2] * bv[2]);
      const float l = hmax(v - delta, 0.f), h = hmax(v + delta, 0.f);
      lo += l * l;
      hi += h * h;
    }
    if (lo > bd_hi) {
      nd_lo = hsqrt(lo) * (1.f - 8.f * OBBF_U);
      nd_hi = hsqrt(hi) * (1.f + 8.f * OBBF_U);
      return OBBF_DISJOINT;
    }
    if (!(hi <= bd_lo)) return OBBF_UNSURE;
  }
  // ---- B axes: sq = sum_j max(|col_j(R) . Tc| - col_j(F) . a - b_j, 0)^2
  {
    float lo = 0.f, hi = 0.f;
#pragma unroll
    for (int j = 0; j < 3; ++j) {
      const float v = habs(R[0][j] * Tc[0] + R[1][j] * Tc[1] + R[2][j] * Tc[2]) - (F[0][j] * av[0] + F[1][j] * av[1] + F[2][j] * av[2]) - bv[j];
      const float l = hmax(v - delta, 0.f), h = hmax(v + delta, 0.f);
      lo += l * l;
      hi += h * h;
    }
    if (lo > bd_hi) {
      nd_lo = hsqrt(lo) * (1.f - 8.f * OBBF_U);
      nd_hi = hsqrt(hi) * (1.f + 8.f * OBBF_U);
      return OBBF_DISJOINT;
    }
    if (!(hi <= bd_lo)) return OBBF_UNSURE;
  }
  // ---- edge axes A_ia x B_ib, in the reference's order; each decides on its own
#pragma unroll
  for (int ia = 0; ia < 3; ++ia) {
    const int ja = (ia + 1) % 3, ka = (ia + 2) % 3;
#pragma unroll
    for (int ib = 0; ib < 3; ++ib) {
      const int jb = (ib + 1) % 3, kb = (ib + 2) % 3;
      const float f = F[ia][ib];
      const float sin2 = 1.f - f * f;
      // fp64 skips the axis when sinus2 < 1e-6: only provable here when the fp32 value is clear of it on the upper side
      if (!(sin2 - OBBF_DELTA_SIN >= 1e-6f)) return OBBF_UNSURE;
      const float s = Tc[ka] * R[ja][ib] - Tc[ja] * R[ka][ib];
      const float diff = habs(s) - (av[ja] * F[ka][ib] + av[ka] * F[ja][ib] + bv[jb] * F[ia][kb] + bv[kb] * F[ia][jb]);
      const float dh = diff + delta;
      if (dh <= 0.f) continue;  // fp64: diff <= 0, the axis does not separate
      const float dl = diff - delta;
      const float s_hi = sin2 + OBBF_DELTA_SIN, s_lo = sin2 - OBBF_DELTA_SIN;
      if (dl > 0.f && dl * dl > bd_hi * s_hi) {
        nd_lo = dl / hsqrt(s_hi) * (1.f - 8.f * OBBF_U);
        nd_hi = dh / hsqrt(s_lo) * (1.f + 8.f * OBBF_U);
        return OBBF_DISJOINT;
      }
      if (!(dh * dh <= bd_lo * s_lo)) return OBBF_UNSURE;
    }
  }
  return OBBF_OVERLAP;
}

// getShapeSupport(TriangleP), support_functions.cpp:110-134
template <typename T>
HFCL_HD V3<T> tri_support(const V3<T>& a, const V3<T>& b, const V3<T>& c, const V3<T>& dir) {
  const T da = dot(dir, a), db = dot(dir, b), dc = dot(dir, c);
  if (da > db) return (dc > da) ? c : a;
  return (dc > db) ? c : b;
}

template <typename T>
struct TriSupport {
  V3<T> p1, p2, p3, q1, q2, q3;  // world frame
  HFCL_HD void operator()(const V3<T>& dir, V3<T>& w, V3<T>& w0) const {
    w0 = tri_support(p1, p2, p3, dir);
    w = w0 - tri_support(q1, q2, q3, -dir);
  }
};

// ShapeShapeDistance<TriangleP,TriangleP>: returns the distance, fills world-frame p1,p2,normal.
// PS: where the simplex keeps its witness payload (hfcl_pair.hpp: W0Regs; the traversal kernels park it in LDS).
template <typename T, class PS>
HFCL_HD T tri_tri_distance(const TriSupport<T>& tri, const GjkParams<T>& prm_in, bool cached_guess, const V3<T>& guess_c,
                           V3<T>& p1, V3<T>& p2, V3<T>& normal, int& status, int& iters, V3<T>* ray_out, const PS& ps) {
  GjkParams<T> prm = prm_in;  // fresh GJKSolver(request): DefaultGJK, Default/Relative criterion, no early stop
  prm.variant = VAR_DEFAULT;
  prm.crit = CRIT_DEFAULT;
  prm.crit_type = CRIT_RELATIVE;
  prm.distance_upper_bound = Lim<T>::max();
  const V3<T> guess = cached_guess ? guess_c : ((tri.p1 + tri.p2 + tri.p3 - tri.q1 - tri.q2 - tri.q3) / T(3));
  Gjk<T, typename PS::P> g;
  TriSupport<T> sup = tri;
  gjk_run(g, prm, guess, T(0), false, sup, ps);
  status = g.status;
  iters = g.iterations;
  if (ray_out) *ray_out = g.ray;  // solver->cached_guess = gjk.getGuessFromSimplex() (:80)
  // gjk.getWitnessPointsAndNormal for any rank (1..4); reference order: ref[i] = s[rank-1-i]
  typedef SimplexV<T, typename PS::P> SV;
  const int r = g.rank;
  const SV ref0 = svsel(r == 1, g.s0, svsel(r == 2, g.s1, svsel(r == 3, g.s2, g.s3)));
  const SV ref1 = svsel(r == 2, g.s0, svsel(r == 3, g.s1, g.s2));
  const SV ref2 = svsel(r == 3, g.s0, g.s1);
  const SV ref3 = g.s0;
  const V3<T> a0 = ps.get(ref0.p), b0 = r > 1 ? ps.get(ref1.p) : a0, c0 = r > 2 ? ps.get(ref2.p) : a0;
  if (r == 4) {
    const V3<T> d0 = ps.get(ref3.p);
    T prm4[4];
    project_tetra_origin(ref0.w, ref1.w, ref2.w, ref3.w, prm4);
    const V3<T> z = mk<T>(T(0), T(0), T(0));
    p1 = (((z + prm4[0] * a0) + prm4[1] * b0) + prm4[2] * c0) + prm4[3] * d0;
    p2 = (((z + prm4[0] * (a0 - ref0.w)) + prm4[1] * (b0 - ref1.w)) + prm4[2] * (c0 - ref2.w)) + prm4[3] * (d0 - ref3.w);
  } else {
    closest_points(r, ref0.w, ref1.w, ref2.w, a0, b0, c0, a0 - ref0.w, b0 - ref1.w, c0 - ref2.w, p1, p2);
  }
  gjk_witness_normal(g.ray, T(0), T(0), p1, p2, normal);
  T distance = g.distance;
  if (g.status == GJK_COLLISION) {  // computePenetration, details.h:699-709
    const V3<T> u = cross(tri.p2 - tri.p1, tri.p3 - tri.p1);
    normal = normalized(u);
    const T d1 = dot(tri.p1 - tri.q1, normal), d2 = dot(tri.p1 - tri.q2, normal), d3 = dot(tri.p1 - tri.q3, normal);
    distance = -hmax(d1, hmax(d2, d3));
  }
  return distance;
}

template <typename T>
HFCL_HD T tri_tri_distance(const TriSupport<T>& tri, const GjkParams<T>& prm_in, bool cached_guess, const V3<T>& guess_c,
                           V3<T>& p1, V3<T>& p2, V3<T>& normal, int& status, int& iters, V3<T>* ray_out = nullptr) {
  return tri_tri_distance(tri, prm_in, cached_guess, guess_c, p1, p2, normal, status, iters, ray_out, W0Regs<T>());
}

// ---------------------------------------------------------------------------------------
// distance(): RSS lower bound and triangle-triangle distance
//   rectDistance / segCoords / inVoronoi   src/BV/RSS.cpp:49-713
//   distance(R0,T0,rss1,rss2)              src/BV/RSS.cpp:995-1005
//   segPoints / sqrTriDistance             src/intersect.cpp:60-368
// ---------------------------------------------------------------------------------------
template <typename T> HFCL_HD T clipr(T v, T a, T b) { return v < a ? a : (v > b ? b : v); }

template <typename T>
HFCL_HD void seg_coords(T& t, T& u, T a, T b, T A_dot_B, T A_dot_T, T B_dot_T) {
  const T denom = T(1) - A_dot_B * A_dot_B;
  t = (denom == T(0)) ? T(0) : clipr((A_dot_T - B_dot_T * A_dot_B) / denom, T(0), a);
  u = t * A_dot_B - B_dot_T;
  if (u < T(0)) {
    u = T(0);
    t = clipr(A_dot_T, T(0), a);
  } else if (u > b) {
    u = b;
    t = clipr(u * A_dot_B + A_dot_T, T(0), a);
  }
}
template <typename T>
HFCL_HD bool in_voronoi(T a, T b, T Anorm_dot_B, T Anorm_dot_T, T A_dot_B, T A_dot_T, T B_dot_T) {
  if (habs(Anorm_dot_B) < T(1e-7)) return false;
  const T u = clipr(-Anorm_dot_T / Anorm_dot_B, T(0), b);
  const T t = clipr(u * A_dot_B + A_dot_T, T(0), a);
  const T v = t * A_dot_B - B_dot_T;
  return (Anorm_dot_B > T(0)) ? (v > (u + T(1e-7))) : (v < (u - T(1e-7)));
}

// rectDistance: distance between rectangle A = [0,a0]x[0,a1]x{0} and rectangle B with origin Tab,
// axes = columns 0,1 of Rab, side lengths b0,b1.  The reference enumerates 16 edge pairs as
// straight-line code; here the same tests are generated from the edge-pair geometry:
//   A edge: runs along axis ea at coordinate ua*a[oa] on the other axis oa = 1-ea,
//   B edge: runs along B's axis eb at B-coordinate ub*b[ob], ob = 1-eb,
// in the reference's order (ea,eb) = (1,1),(1,0),(0,1),(0,0), each with (U,U),(U,L),(L,U),(L,L).
// Two passes (round 3).  The reference's code tries the 16 pairs one after the other and returns from the first whose
// Voronoi tests pass; on a wavefront every lane passes in a different block, so all 16 ran nearly empty.  Pass 1
// evaluates only the cheap entry condition of each block (four comparisons on edge end points) for all 16 and keeps a
// bit mask; pass 2 takes the candidates of a lane in the reference's order and evaluates ONE parametrised block per trip
// (the edge-pair indices select the operands), so the lanes of a wave share the code of a trip whatever block each is in:
// as many trips as the lane with the most candidates needs (1-3 typically) instead of 16 sections.  Every expression is
// the one of the unrolled form (same operands, same order).
// The pieces of rectDistance as a state + steps, so that a kernel can run ONE candidate per lane and round
// (the candidates a lane still has to try travel as a mask: profiles/r04_f); rect_distance() below is their sequence.
template <typename T>
struct RectTest {
  T R[3][3], av[2], bv[2], Tabv[3], Tba[3];
  HFCL_HD void init(const M3<T>& Rab, const V3<T>& Tab, T a0, T a1, T b0, T b1) {
    R[0][0] = Rab.r0.x; R[0][1] = Rab.r0.y; R[0][2] = Rab.r0.z;
    R[1][0] = Rab.r1.x; R[1][1] = Rab.r1.y; R[1][2] = Rab.r1.z;
    R[2][0] = Rab.r2.x; R[2][1] = Rab.r2.y; R[2][2] = Rab.r2.z;
    av[0] = a0; av[1] = a1; bv[0] = b0; bv[1] = b1;
    const V3<T> Tba_v = tmul(Rab, Tab);
    Tabv[0] = Tab.x; Tabv[1] = Tab.y; Tabv[2] = Tab.z;
    Tba[0] = Tba_v.x; Tba[1] = Tba_v.y; Tba[2] = Tba_v.z;
  }
  // which blocks does the reference enter?  bit k = k-th block in its order
  HFCL_HD unsigned pass1() const {
    unsigned mask = 0u;
    {
      int k = 0;
#pragma unroll
      for (int ea = 1; ea >= 0; --ea) {
        const int oa = 1 - ea;
#pragma unroll
        for (int eb = 1; eb >= 0; --eb) {
          const int ob = 1 - eb;
          const T A_ll = -Tba[ob];
          const T A_e = av[ea] * R[ea][ob];
          const T A_o = av[oa] * R[oa][ob];
          const T B_ll = Tabv[oa];
          const T B_e = bv[eb] * R[oa][eb];
          const T B_o = bv[ob] * R[oa][ob];
#pragma unroll
          for (int ua = 1; ua >= 0; --ua) {
#pragma unroll
            for (int ub = 1; ub >= 0; --ub) {
              const T ea0 = A_ll + (ua ? A_o : T(0)), ea1 = ea0 + A_e;
              const T A_l = hmin(ea0, ea1), A_u = hmax(ea0, ea1);
              const T eb0 = B_ll + (ub ? B_o : T(0)), eb1 = eb0 + B_e;
              const T B_l = hmin(eb0, eb1), B_u = hmax(eb0, eb1);
              const bool pre1 = ub ? (A_u > bv[ob]) : (A_l < T(0));
              const bool pre2 = ua ? (B_u > av[oa]) : (B_l < T(0));
              if (pre1 && pre2) mask |= 1u << k;
              ++k;
            }
          }
        }
      }
    }
    return mask;
  }
  // what the block of a candidate selects of the state for its closest-point computation
  struct Sel {
    bool ea1_;
    T av_ea, bv_eb, A_dot_B, A_dot_T, B_dot_T, pa, pb;
    T Rc_eb[3], Rc_ob[3];
  };
  // the first candidate of `mask` (removed from it): true when both of its Voronoi tests pass (two divisions); its operands in `s`
  HFCL_HD bool decide(unsigned& mask, Sel& s) const {
    bool& ea1_ = s.ea1_;
    T &av_ea = s.av_ea, &bv_eb = s.bv_eb, &A_dot_B = s.A_dot_B, &A_dot_T = s.A_dot_T, &B_dot_T = s.B_dot_T, &pa = s.pa, &pb = s.pb;
    T* const Rc_eb = s.Rc_eb;
    T* const Rc_ob = s.Rc_ob;
    {
      const int k = __builtin_ctz(mask);
      mask &= mask - 1u;
      const bool eb1_ = !(k & 4), ua = !(k & 2), ub = !(k & 1);  // eb = 1 / upper A edge / upper B edge
      ea1_ = !(k & 8);                                            // ea = 1
      // operands selected by the edge-pair indices (ea, oa = 1 - ea index A's axes / rows of R; eb, ob index B's axes / columns)
      Rc_eb[0] = eb1_ ? R[0][1] : R[0][0]; Rc_eb[1] = eb1_ ? R[1][1] : R[1][0]; Rc_eb[2] = eb1_ ? R[2][1] : R[2][0];  // R[.][eb]
      Rc_ob[0] = eb1_ ? R[0][0] : R[0][1]; Rc_ob[1] = eb1_ ? R[1][0] : R[1][1]; Rc_ob[2] = eb1_ ? R[2][0] : R[2][1];  // R[.][ob]
      const T R_ea_eb = ea1_ ? Rc_eb[1] : Rc_eb[0], R_oa_eb = ea1_ ? Rc_eb[0] : Rc_eb[1];
      const T R_ea_ob = ea1_ ? Rc_ob[1] : Rc_ob[0], R_oa_ob = ea1_ ? Rc_ob[0] : Rc_ob[1];
      av_ea = ea1_ ? av[1] : av[0];
      const T av_oa = ea1_ ? av[0] : av[1];
      bv_eb = eb1_ ? bv[1] : bv[0];
      const T bv_ob = eb1_ ? bv[0] : bv[1];
      const T Tab_ea = ea1_ ? Tabv[1] : Tabv[0], Tab_oa = ea1_ ? Tabv[0] : Tabv[1];
      const T Tba_eb = eb1_ ? Tba[1] : Tba[0], Tba_ob = eb1_ ? Tba[0] : Tba[1];
      const T A_ll = -Tba_ob;
      const T A_e = av_ea * R_ea_ob;
      const T A_o = av_oa * R_oa_ob;
      const T B_ll = Tab_oa;
      const T B_e = bv_eb * R_oa_eb;
      const T B_o = bv_ob * R_oa_ob;
      const T ea0 = A_ll + (ua ? A_o : T(0)), ea1 = ea0 + A_e;  // A edge end points in B's ob coordinate
      const T A_l = hmin(ea0, ea1), A_u = hmax(ea0, ea1);
      const T eb0 = B_ll + (ub ? B_o : T(0)), eb1 = eb0 + B_e;  // B edge end points in A's oa coordinate
      const T B_l = hmin(eb0, eb1), B_u = hmax(eb0, eb1);
      pa = ua ? av_oa : T(0);
      pb = ub ? bv_ob : T(0);
      A_dot_B = R_ea_eb;
      A_dot_T = Tab_ea + pb * R_ea_ob;  // e_ea . (Pb - Pa)
      B_dot_T = Tba_eb - pa * R_oa_eb;  // B_eb . (Pb - Pa)
      const bool skip1 = ub ? (A_l > bv_ob) : (A_u < T(0));
      const T sgn_b = ub ? T(1) : T(-1);
      const bool v1 = skip1 || in_voronoi(bv_eb, av_ea, sgn_b * R_ea_ob, sgn_b * (pa * R_oa_ob - Tba_ob - pb), A_dot_B,
                                          pa * R_oa_eb - Tba_eb, -Tab_ea - pb * R_ea_ob);
      if (!v1) return false;
      const bool skip2 = ua ? (B_l > av_oa) : (B_u < T(0));
      const T sgn_a = ua ? T(1) : T(-1);
      const bool v2 = skip2 || in_voronoi(av_ea, bv_eb, sgn_a * R_oa_eb, sgn_a * (Tab_oa + pb * R_oa_ob - pa), A_dot_B, A_dot_T, B_dot_T);
      if (!v2) return false;
    }
    return true;
  }
  // the distance of the edge pair that passed: seg_coords (a third division) and the closest points
  HFCL_HD T finish(const Sel& s) const {
    const bool ea1_ = s.ea1_;
    const T av_ea = s.av_ea, bv_eb = s.bv_eb, A_dot_B = s.A_dot_B, A_dot_T = s.A_dot_T, B_dot_T = s.B_dot_T, pa = s.pa, pb = s.pb;
    const T* const Rc_eb = s.Rc_eb;
    const T* const Rc_ob = s.Rc_ob;
    T t, u;
    seg_coords(t, u, av_ea, bv_eb, A_dot_B, A_dot_T, B_dot_T);
    // S = (Pb + u B_eb) - (Pa + t A_ea); component oa loses pa, component ea loses t (third component: neither)
    T S0 = Tabv[0] + Rc_ob[0] * pb + Rc_eb[0] * u;
    T S1 = Tabv[1] + Rc_ob[1] * pb + Rc_eb[1] * u;
    const T S2 = Tabv[2] + Rc_ob[2] * pb + Rc_eb[2] * u;
    S0 -= ea1_ ? pa : t;  // ea = 1: oa = 0
    S1 -= ea1_ ? t : pa;
    return hsqrt(S0 * S0 + S1 * S1 + S2 * S2);
  }
  // no edge pair passed: the two face separations
  HFCL_HD T faces() const {
    const T a0 = av[0], a1 = av[1], b0 = bv[0], b1 = bv[1];
    T sep1, sep2;
    if (Tabv[2] > T(0)) {
      sep1 = Tabv[2];
      if (R[2][0] < T(0)) sep1 += b0 * R[2][0];
      if (R[2][1] < T(0)) sep1 += b1 * R[2][1];
    } else {
      sep1 = -Tabv[2];
      if (R[2][0] > T(0)) sep1 -= b0 * R[2][0];
      if (R[2][1] > T(0)) sep1 -= b1 * R[2][1];
    }
    if (Tba[2] < T(0)) {
      sep2 = -Tba[2];
      if (R[0][2] < T(0)) sep2 += a0 * R[0][2];
      if (R[1][2] < T(0)) sep2 += a1 * R[1][2];
    } else {
      sep2 = Tba[2];
      if (R[0][2] > T(0)) sep2 -= a0 * R[0][2];
      if (R[1][2] > T(0)) sep2 -= a1 * R[1][2];
    }
    const T sep = sep1 > sep2 ? sep1 : sep2;
    return sep > T(0) ? sep : T(0);
  }
};

template <typename T>
HFCL_HD T rect_distance(const M3<T>& Rab, const V3<T>& Tab, T a0, T a1, T b0, T b1) {
  RectTest<T> rt;
  rt.init(Rab, Tab, a0, a1, b0, b1);
  unsigned mask = rt.pass1();
  // The candidates in order, one parametrised block per trip.  The loop only DECIDES; the closest points of the edge pair that
  // passes are computed once behind it: on a wavefront the loop runs as many trips as its slowest lane needs (3-4 of 64
  // lanes' 1.2 on average), so what every trip carries is paid 3-4 times (round 4).
  typename RectTest<T>::Sel sel;
  bool found = false;
  while (mask)
    if (rt.decide(mask, sel)) {
      found = true;
      break;
    }
  return found ? rt.finish(sel) : rt.faces();
}

// the operands of distance(R0, T0, rss1, rss2) for RectTest, and the radii that come off the rectangle distance
template <typename T>
HFCL_HD T rss_rect_setup(RectTest<T>& rt, const M3<T>& R0, const V3<T>& T0, const DNodeD<T>& n1, const DNodeD<T>& n2) {
  const M3<T> R = tmul(n1.axes, mmul(R0, n2.axes));
  const V3<T> Ttemp = mul(R0, n2.Tr) + T0 - n1.Tr;
  const V3<T> Tv = tmul(n1.axes, Ttemp);
  rt.init(R, Tv, n1.l0, n1.l1, n2.l0, n2.l1);
  return n1.r + n2.r;
}

// distance(R0, T0, b1.rss, b2.rss): lower bound of the distance between the two node volumes
template <typename T>
HFCL_HD T rss_lower_bound(const M3<T>& R0, const V3<T>& T0, const DNode<T>& n1, const DRss<T>& r1, const DNode<T>& n2,
                          const DRss<T>& r2) {
  const M3<T> R = tmul(n1.axes, mmul(R0, n2.axes));
  const V3<T> Ttemp = mul(R0, r2.Tr) + T0 - r1.Tr;
  const V3<T> Tv = tmul(n1.axes, Ttemp);
  const T d = rect_distance(R, Tv, r1.l0, r1.l1, r2.l0, r2.l1) - (r1.r + r2.r);
  return d < T(0) ? T(0) : d;
}

// A cheap lower bound of rss_lower_bound(): the separation of the two swept rectangles along three directions -- the normal of rectangle 1, the
// normal of rectangle 2, the line through the two centres -- in the frame of rectangle 1 (R, Tv as rect_distance takes them: rectangle 1 spans
// [0, l0] x [0, l1] in the plane z = 0, rectangle 2 has its corner at Tv and its sides m0, m1 along the columns 0 and 1 of R), less a slack
// of 1e-9 of the lengths involved (fp64: seven orders above the rounding of either computation).  The distance of two convex sets is at least their
// separation along any direction, so a pair whose bound already exceeds the walk's minimum is one the reference prunes too (canStop), whatever
// the exact value: its rectDistance need not run.  On cfg4d's walks this decides 30 % of all tests, 70 % of those the exact value prunes at
// that moment, and never exceeds the exact distance (13 M tests, tools/bound_probe.cpp).
template <typename T>
HFCL_HD T rss_cheap_bound(const M3<T>& R, const V3<T>& Tv, T l0, T l1, T m0, T m1, T rsum) {
  const T h0 = T(0.5) * l0, h1 = T(0.5) * l1, g0 = T(0.5) * m0, g1 = T(0.5) * m1;
  const T cx = Tv.x + R.r0.x * g0 + R.r0.y * g1 - h0, cy = Tv.y + R.r1.x * g0 + R.r1.y * g1 - h1, cz = Tv.z + R.r2.x * g0 + R.r2.y * g1;
  const T gap1 = habs(cz) - (g0 * habs(R.r2.x) + g1 * habs(R.r2.y));
  const T gap2 = habs(cx * R.r0.z + cy * R.r1.z + cz * R.r2.z) - (h0 * habs(R.r0.z) + h1 * habs(R.r1.z));
  const T L2 = cx * cx + cy * cy + cz * cz;
  const T L = hsqrt(L2);
  const T e = h0 * habs(cx) + h1 * habs(cy) + g0 * habs(R.r0.x * cx + R.r1.x * cy + R.r2.x * cz) + g1 * habs(R.r0.y * cx + R.r1.y * cy + R.r2.y * cz);
  const T gap3 = L > T(0) ? L - e / L : T(-1);
  const T lb = hmax(gap1, hmax(gap2, gap3)) - rsum;
  return lb - T(sizeof(T) == 4 ? 1e-5 : 1e-9) * (L + h0 + h1 + g0 + g1 + rsum);  // (fp32: two orders above ITS rounding)
}
template <typename T>
HFCL_HD T rss_cheap_bound(const M3<T>& R0, const V3<T>& T0, const DNodeD<T>& n1, const DNodeD<T>& n2) {
  const M3<T> R = tmul(n1.axes, mmul(R0, n2.axes));
  const V3<T> Tv = tmul(n1.axes, mul(R0, n2.Tr) + T0 - n1.Tr);
  return rss_cheap_bound(R, Tv, n1.l0, n1.l1, n2.l0, n2.l1, n1.r + n2.r);
}

template <typename T>
HFCL_HD T rss_lower_bound(const M3<T>& R0, const V3<T>& T0, const DNodeD<T>& n1, const DNodeD<T>& n2) {
  const M3<T> R = tmul(n1.axes, mmul(R0, n2.axes));
  const V3<T> Ttemp = mul(R0, n2.Tr) + T0 - n1.Tr;
  const V3<T> Tv = tmul(n1.axes, Ttemp);
  const T d = rect_distance(R, Tv, n1.l0, n1.l1, n2.l0, n2.l1) - (n1.r + n2.r);
  return d < T(0) ? T(0) : d;
}

HFCL_HD bool hisnan(float x) { return !(x == x); }
HFCL_HD bool hisnan(double x) { return !(x == x); }

template <typename T>
HFCL_HD void seg_points(const V3<T>& P, const V3<T>& A, const V3<T>& Q, const V3<T>& B, V3<T>& VEC, V3<T>& X, V3<T>& Y) {
  V3<T> Tv = Q - P;
  const T A_dot_A = dot(A, A), B_dot_B = dot(B, B), A_dot_B = dot(A, B), A_dot_T = dot(A, Tv), B_dot_T = dot(B, Tv);
  const T denom = A_dot_A * B_dot_B - A_dot_B * A_dot_B;
  T t = (A_dot_T * B_dot_B - B_dot_T * A_dot_B) / denom;
  if ((t < T(0)) || hisnan(t))
    t = T(0);
  else if (t > T(1))
    t = T(1);
  const T u = (t * A_dot_B - B_dot_T) / B_dot_B;
  if ((u <= T(0)) || hisnan(u)) {
    Y = Q;
    t = A_dot_T / A_dot_A;
    if ((t <= T(0)) || hisnan(t)) {
      X = P;
      VEC = Q - P;
    } else if (t >= T(1)) {
      X = P + A;
      VEC = Q - X;
    } else {
      X = P + A * t;
      VEC = cross(A, cross(Tv, A));
    }
  } else if (u >= T(1)) {
    Y = Q + B;
    t = (A_dot_B + A_dot_T) / A_dot_A;
    if ((t <= T(0)) || hisnan(t)) {
      X = P;
      VEC = Y - P;
    } else if (t >= T(1)) {
      X = P + A;
      VEC = Y - X;
    } else {
      X = P + A * t;
      Tv = Y - P;
      VEC = cross(A, cross(Tv, A));
    }
  } else {
    Y = Q + B * u;
    if ((t <= T(0)) || hisnan(t)) {
      X = P;
      VEC = cross(B, cross(Tv, B));
    } else if (t >= T(1)) {
      X = P + A;
      Tv = Q - X;
      VEC = cross(B, cross(Tv, B));
    } else {
      X = P + A * t;
      VEC = cross(A, B);
      if (dot(VEC, Tv) < T(0)) VEC = -VEC;
    }
  }
}

// sqrTriDistance(S, T, P, Q): squared distance between triangles (0 when they overlap)
template <typename T>
HFCL_HD T sqr_tri_distance(const V3<T>& s0, const V3<T>& s1, const V3<T>& s2, const V3<T>& t0, const V3<T>& t1,
                           const V3<T>& t2, V3<T>& P, V3<T>& Q) {
  const V3<T> S[3] = {s0, s1, s2}, Tt[3] = {t0, t1, t2};
  const V3<T> Sv[3] = {s1 - s0, s2 - s1, s0 - s2}, Tv[3] = {t1 - t0, t2 - t1, t0 - t2};
  V3<T> minP = s0, minQ = t0;
  bool shown_disjoint = false;
  T mindd = sqnorm(s0 - t0) + T(1);
#pragma unroll
  for (int i = 0; i < 3; ++i) {
#pragma unroll
    for (int j = 0; j < 3; ++j) {
      V3<T> VEC;
      seg_points(S[i], Sv[i], Tt[j], Tv[j], VEC, P, Q);
      const V3<T> V = Q - P;
      const T dd = dot(V, V);
      if (dd <= mindd) {
        minP = P;
        minQ = Q;
        mindd = dd;
        T a = dot(S[(i + 2) % 3] - P, VEC);
        T b = dot(Tt[(j + 2) % 3] - Q, VEC);
        if ((a <= T(0)) && (b >= T(0))) return dd;
        const T p = dot(V, VEC);
        if (a < T(0)) a = T(0);
        if (b > T(0)) b = T(0);
        if ((p - a + b) > T(0)) shown_disjoint = true;
      }
    }
  }
  const V3<T> Sn = cross(Sv[0], Sv[1]);
  const T Snl = dot(Sn, Sn);
  if (Snl > T(1e-15)) {
    const T p0 = dot(s0 - t0, Sn), p1 = dot(s0 - t1, Sn), p2 = dot(s0 - t2, Sn);
    int point = -1;
    if ((p0 > T(0)) && (p1 > T(0)) && (p2 > T(0))) {
      point = (p0 < p1) ? 0 : 1;
      if (p2 < (point == 0 ? p0 : p1)) point = 2;
    } else if ((p0 < T(0)) && (p1 < T(0)) && (p2 < T(0))) {
      point = (p0 > p1) ? 0 : 1;
      if (p2 > (point == 0 ? p0 : p1)) point = 2;
    }
    if (point >= 0) {
      shown_disjoint = true;
      const V3<T> tp = point == 0 ? t0 : (point == 1 ? t1 : t2);
      const T pp = point == 0 ? p0 : (point == 1 ? p1 : p2);
      if (dot(tp - s0, cross(Sn, Sv[0])) > T(0) && dot(tp - s1, cross(Sn, Sv[1])) > T(0) &&
          dot(tp - s2, cross(Sn, Sv[2])) > T(0)) {
        P = tp + Sn * (pp / Snl);
        Q = tp;
        return sqnorm(P - Q);
      }
    }
  }
  const V3<T> Tn = cross(Tv[0], Tv[1]);
  const T Tnl = dot(Tn, Tn);
  if (Tnl > T(1e-15)) {
    const T p0 = dot(t0 - s0, Tn), p1 = dot(t0 - s1, Tn), p2 = dot(t0 - s2, Tn);
    int point = -1;
    if ((p0 > T(0)) && (p1 > T(0)) && (p2 > T(0))) {
      point = (p0 < p1) ? 0 : 1;
      if (p2 < (point == 0 ? p0 : p1)) point = 2;
    } else if ((p0 < T(0)) && (p1 < T(0)) && (p2 < T(0))) {
      point = (p0 > p1) ? 0 : 1;
      if (p2 > (point == 0 ? p0 : p1)) point = 2;
    }
    if (point >= 0) {
      shown_disjoint = true;
      const V3<T> sp = point == 0 ? s0 : (point == 1 ? s1 : s2);
      const T pp = point == 0 ? p0 : (point == 1 ? p1 : p2);
      if (dot(sp - t0, cross(Tn, Tv[0])) > T(0) && dot(sp - t1, cross(Tn, Tv[1])) > T(0) &&
          dot(sp - t2, cross(Tn, Tv[2])) > T(0)) {
        P = sp;
        Q = sp + Tn * (pp / Tnl);
        return sqnorm(P - Q);
      }
    }
  }
  if (shown_disjoint) {
    P = minP;
    Q = minQ;
    return mindd;
  }
  return T(0);
}

}  // namespace hfcl
