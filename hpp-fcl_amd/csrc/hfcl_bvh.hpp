// hfcl_bvh.hpp -- BVHModel<OBBRSS> x BVHModel<OBBRSS> collide(): BV test and leaf test.
//
// Behavioural contract (reference file:line):
//   OBB SAT with lower bound   src/BV/OBB.cpp:290-393 (obbDisjointAndLowerBoundDistance), :475-483 (overlap)
//   leaf = triangle-triangle   include/hpp/fcl/internal/traversal_node_bvhs.h:184-233 (leafCollides)
//                              src/distance/triangle_triangle.cpp:46-105 (vanilla GJK on world-frame
//                              triangles, centroid guess, no EPA), src/narrowphase/details.h:699-709
//   triangle support           src/narrowphase/support_functions.cpp:110-134
#pragma once
#include "hfcl_pair.hpp"

namespace hfcl {

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
template <typename T>
HFCL_HD T tri_tri_distance(const TriSupport<T>& tri, const GjkParams<T>& prm_in, bool cached_guess, const V3<T>& guess_c,
                           V3<T>& p1, V3<T>& p2, V3<T>& normal, int& status, int& iters) {
  GjkParams<T> prm = prm_in;  // fresh GJKSolver(request): DefaultGJK, Default/Relative criterion, no early stop
  prm.variant = VAR_DEFAULT;
  prm.crit = CRIT_DEFAULT;
  prm.crit_type = CRIT_RELATIVE;
  prm.distance_upper_bound = Lim<T>::max();
  const V3<T> guess = cached_guess ? guess_c : ((tri.p1 + tri.p2 + tri.p3 - tri.q1 - tri.q2 - tri.q3) / T(3));
  Gjk<T, PW0<T>> g;
  TriSupport<T> sup = tri;
  gjk_run(g, prm, guess, T(0), false, sup);
  status = g.status;
  iters = g.iterations;
  // gjk.getWitnessPointsAndNormal for any rank (1..4); reference order: ref[i] = s[rank-1-i]
  typedef SimplexV<T, PW0<T>> SV;
  const int r = g.rank;
  const SV ref0 = svsel(r == 1, g.s0, svsel(r == 2, g.s1, svsel(r == 3, g.s2, g.s3)));
  const SV ref1 = svsel(r == 2, g.s0, svsel(r == 3, g.s1, g.s2));
  const SV ref2 = svsel(r == 3, g.s0, g.s1);
  const SV ref3 = g.s0;
  if (r == 4) {
    T prm4[4];
    project_tetra_origin(ref0.w, ref1.w, ref2.w, ref3.w, prm4);
    const V3<T> z = mk<T>(T(0), T(0), T(0));
    p1 = (((z + prm4[0] * ref0.p.w0) + prm4[1] * ref1.p.w0) + prm4[2] * ref2.p.w0) + prm4[3] * ref3.p.w0;
    p2 = (((z + prm4[0] * (ref0.p.w0 - ref0.w)) + prm4[1] * (ref1.p.w0 - ref1.w)) + prm4[2] * (ref2.p.w0 - ref2.w)) +
         prm4[3] * (ref3.p.w0 - ref3.w);
  } else {
    closest_points(r, ref0.w, ref1.w, ref2.w, ref0.p.w0, ref1.p.w0, ref2.p.w0, ref0.p.w0 - ref0.w, ref1.p.w0 - ref1.w,
                   ref2.p.w0 - ref2.w, p1, p2);
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

}  // namespace hfcl
