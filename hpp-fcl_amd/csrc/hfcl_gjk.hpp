// hfcl_gjk.hpp -- GJK iteration core for the batched MI355X narrow phase.
//
// Behavioural contract: hpp-fcl's details::GJK::evaluate
// (/root/reference/src/narrowphase/gjk.cpp:188-370, checkConvergence :372-425, simplex
// projections :494-1010) -- same checks in the same order, same formulas, so statuses and
// iteration counts agree with the reference.  The implementation is re-designed for SIMT:
//   * one "trip" of the loop is split into begin() -> support -> end() so a kernel can keep
//     many pairs in lock-step and plug in any support evaluator (per-lane primitives,
//     lane-group-parallel convex hulls);
//   * the simplex lives in registers, newest vertex first (A,B,C,D = s0..s3): appending is a
//     register shift, projections leave A in place and only select s1/s2;
//   * the 12-predicate tetrahedron decision tree is evaluated branch-free: all predicates
//     are computed, packed into a 12-bit mask and looked up in a 4096-entry region table
//     generated at compile time from the tree.
#pragma once
#include "hfcl_math.hpp"

namespace hfcl {

enum {
  GJK_DID_NOT_RUN = 0, GJK_FAILED = 1, GJK_EARLY_STOPPED = 2, GJK_NO_COLLISION = 3,
  GJK_COLLISION_WITH_PEN = 4, GJK_COLLISION = 5
};
enum { VAR_DEFAULT = 0, VAR_POLYAK = 1, VAR_NESTEROV = 2 };
enum { CRIT_DEFAULT = 0, CRIT_DUALITY_GAP = 1, CRIT_HYBRID = 2 };
enum { CRIT_RELATIVE = 0, CRIT_ABSOLUTE = 1 };

// ---------------------------------------------------------------------------------------
// Tetrahedron Voronoi-region table (gjk.cpp:613-1010).  Predicate k (bit k-1 of the mask)
// is the reference's "a<k>" test; the tree below is that decision tree reduced to its leaves.
// ---------------------------------------------------------------------------------------
enum { REG_A = 0, REG_AB = 1, REG_AC = 2, REG_AD = 3, REG_ABC = 4, REG_ACD = 5, REG_ADB = 6, REG_INSIDE = 7 };

constexpr uint8_t tetra_region(unsigned m) {
  const bool a1 = m & 1u, a2 = m & 2u, a3 = m & 4u, a4 = m & 8u, a5 = m & 16u, a6 = m & 32u, a7 = m & 64u,
             a8 = m & 128u, a9 = m & 256u, a10 = m & 512u, a11 = m & 1024u, a12 = m & 2048u;
  if (a10) {
    if (a3) {
      if (a9) {
        if (a12) return a4 ? REG_ABC : REG_AB;
        if (a4) return a5 ? (a6 ? REG_ACD : REG_AC) : REG_ABC;
        return REG_AB;
      }
      if (a8) return REG_ADB;
      if (a6) return a7 ? REG_AD : REG_ACD;
      return a7 ? REG_AD : REG_AC;
    }
    if (a1) {
      if (a4) return a5 ? (a6 ? REG_ACD : REG_AC) : REG_ABC;
      return REG_AB;
    }
    if (a2) {
      if (a6) return a7 ? REG_AD : REG_ACD;
      return a11 ? REG_AC : REG_AD;
    }
    return REG_INSIDE;
  }
  if (a11) {
    if (a2) {
      if (a12) {
        if (a6) return a7 ? (a8 ? REG_ADB : REG_AD) : REG_ACD;
        return a5 ? REG_AC : REG_ABC;
      }
      if (a5) return a6 ? REG_ACD : REG_AC;
      return a1 ? REG_ABC : REG_ACD;
    }
    if (a1) return a5 ? REG_AC : REG_ABC;
    if (a3) return a8 ? REG_ADB : REG_AD;
    return REG_INSIDE;
  }
  if (a12) {
    if (a3) {
      if (a7) return a8 ? REG_ADB : REG_AD;
      return a2 ? REG_ACD : REG_ADB;
    }
    if (a2) return a7 ? REG_AD : REG_ACD;
    return REG_INSIDE;
  }
  return REG_A;
}

struct TetraLut {
  uint8_t v[4096];
  constexpr TetraLut() : v() {
    for (unsigned m = 0; m < 4096; ++m) v[m] = tetra_region(m);
  }
};
#if defined(__HIPCC__)
__device__ static const TetraLut g_tetra_lut_dev = TetraLut();
#endif
static constexpr TetraLut g_tetra_lut_host = TetraLut();
HFCL_HD unsigned tetra_lookup(unsigned mask) {
#if defined(__HIP_DEVICE_COMPILE__)
  return g_tetra_lut_dev.v[mask];
#else
  return g_tetra_lut_host.v[mask];
#endif
}

// ---------------------------------------------------------------------------------------
template <typename T>
struct GjkParams {
  T tolerance;
  T distance_upper_bound;
  unsigned max_iterations;
  int variant;
  int crit;
  int crit_type;
};

// P = per-vertex payload carried with every simplex vertex (witness data)
template <typename T, class P>
struct SimplexV {
  V3<T> w;
  P p;
};

template <typename T, class P>
struct Gjk {
  typedef SimplexV<T, P> SV;
  SV s0, s1, s2, s3;  // newest first: A = s0, B = s1, C = s2, D = s3
  int rank;
  V3<T> ray, dir, w;
  T rl, alpha, distance, ssr, upper_bound;
  int iterations, variant, status;
  bool done, normalize;
};

// Component-wise select of a simplex vertex.  (A `c ? a : b` on the structs themselves turns
// into a select of *addresses*, which keeps the whole simplex in scratch memory.)
template <typename T, class P>
HFCL_HD SimplexV<T, P> svsel(bool c, const SimplexV<T, P>& a, const SimplexV<T, P>& b) {
  SimplexV<T, P> r;
  r.w = sel(c, a.w, b.w);
  r.p = psel(c, a.p, b.p);
  return r;
}

template <typename T, class P>
HFCL_HD void gjk_init(Gjk<T, P>& g, const GjkParams<T>& prm, const V3<T>& guess, T ssr_sum, bool normalize_dir) {
  g.alpha = T(0);
  g.iterations = 0;
  g.ssr = ssr_sum;
  g.upper_bound = prm.distance_upper_bound + ssr_sum;
  g.status = GJK_NO_COLLISION;
  g.distance = T(0);
  g.rank = 0;
  g.done = false;
  g.normalize = normalize_dir;
  T rl = norm(guess);
  if (rl < prm.tolerance) {
    g.ray = mk<T>(T(-1), T(0), T(0));
    rl = T(1);
  } else
    g.ray = guess;
  g.rl = rl;
  g.variant = prm.variant;
  g.w = g.ray;
  g.dir = g.ray;
}

// First half of a trip: check A + support direction.  Returns false when the pair is finished.
template <typename T, class P>
HFCL_HD bool gjk_begin(Gjk<T, P>& g, const GjkParams<T>& prm, V3<T>& support_dir) {
  if (g.rl < prm.tolerance) {  // check A (gjk.cpp:228-243)
    g.status = GJK_COLLISION;
    g.distance = g.rl;
    g.done = true;
    return false;
  }
  if (g.variant == VAR_DEFAULT) {
    g.dir = g.ray;
  } else if (g.variant == VAR_NESTEROV) {  // :251-269
    if (g.normalize) {
      T momentum = (T(g.iterations) + T(2)) / (T(g.iterations) + T(3));
      V3<T> y = momentum * g.ray + (T(1) - momentum) * g.w;
      T y_norm = norm(y);
      g.dir = (momentum * g.dir) / norm(g.dir) + ((T(1) - momentum) * y) / y_norm;
    } else {
      T momentum = (T(g.iterations) + T(1)) / (T(g.iterations) + T(3));
      V3<T> y = momentum * g.ray + (T(1) - momentum) * g.w;
      g.dir = momentum * g.dir + (T(1) - momentum) * y;
    }
  } else {  // Polyak :271-274
    T momentum = T(1) / (T(g.iterations) + T(1));
    g.dir = momentum * g.dir + (T(1) - momentum) * g.ray;
  }
  support_dir = -g.dir;
  return true;
}

template <typename T, class P>
HFCL_HD void gjk_pop(Gjk<T, P>& g) {  // removeVertex: drop the newest vertex
  g.s0 = g.s1;
  g.s1 = g.s2;
  g.s2 = g.s3;
  --g.rank;
}

template <typename T, class P>
HFCL_HD bool gjk_check_convergence(Gjk<T, P>& g, const GjkParams<T>& prm, T omega) {  // :372-425
  const T tol = prm.tolerance;
  if (prm.crit == CRIT_DEFAULT) {
    g.alpha = hmax(g.alpha, omega);
    const T diff = g.rl - g.alpha;
    return (diff - (tol + tol * g.rl)) <= T(0);
  }
  T diff;
  if (prm.crit == CRIT_DUALITY_GAP) {
    diff = T(2) * dot(g.ray, g.ray - g.w);
  } else {
    g.alpha = hmax(g.alpha, omega);
    diff = g.rl * g.rl - g.alpha * g.alpha;
  }
  if (prm.crit_type == CRIT_ABSOLUTE) return (diff - tol) <= T(0);
  return ((diff / tol * g.rl) - tol * g.rl) <= T(0);
}

// originToSegment (gjk.cpp:502-515) for A and X; leaves [A, X]
template <typename T, class P>
HFCL_HD void gjk_to_segment(Gjk<T, P>& g, const SimplexV<T, P> X, const V3<T> AX, T AXdotAO) {
  const V3<T> A = g.s0.w;
  V3<T> r = dot(AX, X.w) * A + AXdotAO * X.w;
  g.ray = r / sqnorm(AX);
  g.s1 = X;
  g.rank = 2;
}
// originToTriangle (gjk.cpp:517-541) for (A, X, Y) with normal N and N.AO
template <typename T, class P>
HFCL_HD bool gjk_to_triangle(Gjk<T, P>& g, const SimplexV<T, P> X, const SimplexV<T, P> Y, const V3<T> N, T NdotAO) {
  g.rank = 3;
  const bool keep = (NdotAO >= T(0));  // ==0 and >0: next = [y, x, A]  -> newest-first [A, x, y]
  const SimplexV<T, P> n1 = svsel(keep, X, Y);
  const SimplexV<T, P> n2 = svsel(keep, Y, X);
  g.s1 = n1;
  g.s2 = n2;
  if (NdotAO == T(0)) {
    g.ray = mk<T>(T(0), T(0), T(0));
    return true;
  }
  g.ray = (-NdotAO / sqnorm(N)) * N;
  return false;
}

// The two small projections.  fp32: in SELECT form (round 5) -- every region's result is formed (the expressions are those of gjk_to_segment /
// gjk_to_triangle, term for term) and the region picks among them: a wave steps 16-32 pairs that sit in different regions, so as branches all
// regions ran anyway, one after the other under exec masks, each way out with its copy of the simplex (cfg3 -0.5 %).  fp64 keeps the branches: three
// fp64 divisions where one is needed cost more than the branches (cfg5 +2.2 % with selects; profiles/r05_h section 6).
template <typename T, class P>
HFCL_HD bool gjk_project_line(Gjk<T, P>& g) {  // :543-569
  const V3<T> A = g.s0.w, B = g.s1.w;
  const V3<T> AB = B - A;
  const T d = dot(AB, -A);
  if constexpr (sizeof(T) == 4) {
    const bool seg = !((d == T(0)) | (d < T(0)));  // else: the origin projects onto A (d == 0: A may be the origin itself)
    const V3<T> r_seg = (dot(AB, B) * A + d * B) / sqnorm(AB);  // originToSegment(A, B)
    g.ray = seg ? r_seg : A;
    g.rank = seg ? 2 : 1;
    return (d == T(0)) & is_zero(A);
  } else {
    if (d == T(0)) {
      g.ray = A;
      g.rank = 1;
      return is_zero(A);
    } else if (d < T(0)) {
      g.ray = A;
      g.rank = 1;
    } else {
      gjk_to_segment(g, g.s1, AB, d);
    }
    return false;
  }
}

template <typename T, class P>
HFCL_HD bool gjk_project_triangle(Gjk<T, P>& g) {  // :571-611
  const V3<T> A = g.s0.w, B = g.s1.w, C = g.s2.w;
  const V3<T> AB = B - A, AC = C - A, ABC = cross(AB, AC);
  const T edgeAC2o = dot(cross(ABC, AC), -A);
  if constexpr (sizeof(T) == 4) {
    const T towardsC = dot(AC, -A);
    const T edgeAB2o = dot(cross(AB, ABC), -A);
    const T towardsB = dot(AB, -A);
    const T NdotAO = dot(ABC, -A);
    const bool e_ac = edgeAC2o >= T(0);
    const bool segC = e_ac & (towardsC >= T(0));                      // originToSegment(A, C)
    const bool r45 = (e_ac & !(towardsC >= T(0))) | (!e_ac & (edgeAB2o >= T(0)));
    const bool vertA = r45 & (towardsB < T(0));                        // the vertex A
    const bool segB = r45 & !(towardsB < T(0));                        // originToSegment(A, B)
    const bool tri = !e_ac & !(edgeAB2o >= T(0));                      // originToTriangle(A, B, C)
    const V3<T> r_c = (dot(AC, C) * A + towardsC * C) / sqnorm(AC);
    const V3<T> r_b = (dot(AB, B) * A + towardsB * B) / sqnorm(AB);
    const V3<T> r_t = (NdotAO == T(0)) ? mk<T>(T(0), T(0), T(0)) : ((-NdotAO / sqnorm(ABC)) * ABC);
    const bool keep = NdotAO >= T(0);
    const SimplexV<T, P> n1 = svsel(keep, g.s1, g.s2), n2 = svsel(keep, g.s2, g.s1);
    g.ray = segC ? r_c : (vertA ? A : (segB ? r_b : r_t));
    g.rank = (segC | segB) ? 2 : (vertA ? 1 : 3);
    const SimplexV<T, P> s1 = svsel(segC, g.s2, svsel(tri, n1, g.s1));
    g.s2 = svsel(tri, n2, g.s2);
    g.s1 = s1;
    return tri & (NdotAO == T(0));
  } else {
    bool region45 = false;
    if (edgeAC2o >= T(0)) {
      const T towardsC = dot(AC, -A);
      if (towardsC >= T(0)) {
        gjk_to_segment(g, g.s2, AC, towardsC);
        return false;
      }
      region45 = true;
    } else {
      const T edgeAB2o = dot(cross(AB, ABC), -A);
      if (edgeAB2o >= T(0))
        region45 = true;
      else
        return gjk_to_triangle(g, g.s1, g.s2, ABC, dot(ABC, -A));
    }
    if (region45) {
      const T towardsB = dot(AB, -A);
      if (towardsB < T(0)) {
        g.ray = A;
        g.rank = 1;
      } else
        gjk_to_segment(g, g.s1, AB, towardsB);
    }
    return false;
  }
}

template <typename T, class P>
HFCL_HD bool gjk_project_tetra(Gjk<T, P>& g) {  // :613-1010
  const V3<T> A = g.s0.w, B = g.s1.w, C = g.s2.w, D = g.s3.w;
  const T aa = sqnorm(A);
  const T da = dot(D, A), db = dot(D, B), dc = dot(D, C), dd = dot(D, D);
  const T da_aa = da - aa;
  const T ca = dot(C, A), cb = dot(C, B), cc = dot(C, C);
  const T ca_aa = ca - aa;
  const T ba = dot(B, A), bb = dot(B, B);
  const T ba_aa = ba - aa, ba_ca = ba - ca, ca_da = ca - da, da_ba = da - ba;
  const V3<T> a_cross_b = cross(A, B);
  const V3<T> a_cross_c = cross(A, C);
  const T d_axb = dot(D, a_cross_b), c_axb = dot(C, a_cross_b), d_axc = dot(D, a_cross_c);

  unsigned m = 0;
  m |= (c_axb <= T(0)) ? 1u : 0u;                                        // a1
  m |= (d_axc <= T(0)) ? 2u : 0u;                                        // a2
  m |= (-d_axb <= T(0)) ? 4u : 0u;                                       // a3
  m |= (ba * ba_ca + bb * ca_aa - cb * ba_aa <= T(0)) ? 8u : 0u;         // a4
  m |= (ca * ba_ca + cb * ca_aa - cc * ba_aa <= T(0)) ? 16u : 0u;        // a5
  m |= (ca * ca_da + cc * da_aa - dc * ca_aa <= T(0)) ? 32u : 0u;        // a6
  m |= (da * ca_da + dc * da_aa - dd * ca_aa <= T(0)) ? 64u : 0u;        // a7
  m |= (da * da_ba + dd * ba_aa - db * da_aa <= T(0)) ? 128u : 0u;       // a8
  m |= (ba * da_ba + db * ba_aa - bb * da_aa <= T(0)) ? 256u : 0u;       // a9
  m |= (ba_aa <= T(0)) ? 512u : 0u;                                      // a10
  m |= (ca_aa <= T(0)) ? 1024u : 0u;                                     // a11
  m |= (da_aa <= T(0)) ? 2048u : 0u;                                     // a12
  const unsigned reg = tetra_lookup(m);

  if (reg == REG_INSIDE) {
    g.ray = mk<T>(T(0), T(0), T(0));
    return true;  // rank stays 4, order unchanged
  }
  if (reg == REG_A) {
    g.ray = A;
    g.rank = 1;
    return false;
  }
  if (reg <= REG_AD) {  // segment A-X
    const bool xb = (reg == REG_AB), xc = (reg == REG_AC);
    const SimplexV<T, P> X = svsel(xb, g.s1, svsel(xc, g.s2, g.s3));
    const T xa_aa = xb ? ba_aa : (xc ? ca_aa : da_aa);
    gjk_to_segment(g, X, X.w - A, -xa_aa);
    return false;
  }
  // triangle A-X-Y: ABC -> (B,C), ACD -> (C,D), ADB -> (D,B)
  const bool tb = (reg == REG_ABC), tc = (reg == REG_ACD);
  const SimplexV<T, P> X = svsel(tb, g.s1, svsel(tc, g.s2, g.s3));
  const SimplexV<T, P> Y = svsel(tb, g.s2, svsel(tc, g.s3, g.s1));
  const T ndotao = tb ? -c_axb : (tc ? -d_axc : d_axb);
  gjk_to_triangle(g, X, Y, cross(X.w - A, Y.w - A), ndotao);
  return false;
}

// Second half of a trip: the new support vertex `v` (for direction -dir) is appended and
// checks B, momentum removal, check C and the simplex projection run (gjk.cpp:281-365).
template <typename T, class P>
HFCL_HD void gjk_end(Gjk<T, P>& g, const GjkParams<T>& prm, const SimplexV<T, P>& v) {
  g.s3 = g.s2;
  g.s2 = g.s1;
  g.s1 = g.s0;
  g.s0 = v;
  ++g.rank;
  g.w = v.w;

  const T omega = dot(g.dir, g.w) / norm(g.dir);  // check B
  if (omega > g.upper_bound) {
    g.distance = omega - g.ssr;
    g.status = GJK_EARLY_STOPPED;
    g.done = true;
    return;
  }
  if (g.variant != VAR_DEFAULT) {  // :296-304
    const T gap = T(2) * dot(g.ray, g.ray - g.w);
    if (gap - prm.tolerance <= T(0)) {
      gjk_pop(g);
      g.variant = VAR_DEFAULT;
      return;  // `continue`: next trip, iterations unchanged
    }
  }
  bool cv = gjk_check_convergence(g, prm, omega);  // check C
  if (sizeof(T) == 4) {
    // fp32 only (the reference is fp64-only): when round-off keeps check C a hair above the
    // tolerance, the next support is a vertex the simplex already holds; the projection would
    // then divide by |AB|^2 = 0.  A repeated support vertex means no progress is possible:
    // treat it as converged (the classical GJK termination test the reference notes as
    // "check removed", gjk.cpp:283-284).
    const bool dup = ((g.rank > 1) & (g.w.x == g.s1.w.x) & (g.w.y == g.s1.w.y) & (g.w.z == g.s1.w.z)) |
                     ((g.rank > 2) & (g.w.x == g.s2.w.x) & (g.w.y == g.s2.w.y) & (g.w.z == g.s2.w.z)) |
                     ((g.rank > 3) & (g.w.x == g.s3.w.x) & (g.w.y == g.s3.w.y) & (g.w.z == g.s3.w.z));
    cv = cv | dup;
  }
  if (g.iterations > 0 && cv) {
    gjk_pop(g);
    if (g.variant != VAR_DEFAULT) {
      g.variant = VAR_DEFAULT;
      return;
    }
    g.distance = g.rl - g.ssr;
    g.status = (g.distance < prm.tolerance) ? GJK_COLLISION_WITH_PEN : GJK_NO_COLLISION;
    g.done = true;
    return;
  }
  bool inside = false;
  if (g.rank == 1) {
    g.ray = g.w;
  } else if (g.rank == 2) {
    inside = gjk_project_line(g);
  } else if (g.rank == 3) {
    inside = gjk_project_triangle(g);
  } else {
    inside = gjk_project_tetra(g);
  }
  g.rl = norm(g.ray);
  if (sizeof(T) == 4 && !(g.rl == g.rl)) {  // fp32 safety net: never iterate on a NaN ray
    g.status = GJK_FAILED;
    g.done = true;
    return;
  }
  if (inside || g.rl == T(0)) {
    g.status = GJK_COLLISION;
    g.distance = g.rl;
    g.done = true;
    return;
  }
  ++g.iterations;
  if (!(unsigned(g.iterations) < prm.max_iterations)) {
    g.status = GJK_FAILED;
    g.done = true;
  }
}

// ---------------------------------------------------------------------------------------
// Witness points: Project::project{Line,Triangle}Origin (src/intersect.cpp:570-646) and
// details::getClosestPoints (gjk.cpp:94-151).  Vertices are passed in the reference's order
// (oldest first): v[i] = simplex.vertex[i].
// ---------------------------------------------------------------------------------------
template <typename T>
HFCL_HD void project_line_origin(const V3<T>& a, const V3<T>& b, T& p0, T& p1, T& sqd) {
  const V3<T> d = b - a;
  const T l = sqnorm(d);
  p0 = T(0);
  p1 = T(0);
  sqd = T(-1);
  if (l > T(0)) {
    const T t = -dot(a, d);
    p1 = (t >= l) ? T(1) : ((t <= T(0)) ? T(0) : (t / l));
    p0 = T(1) - p1;
    if (t >= l)
      sqd = sqnorm(b);
    else if (t <= T(0))
      sqd = sqnorm(a);
    else
      sqd = sqnorm(a + d * p1);
  }
}

template <typename T>
HFCL_HD T project_triangle_origin(const V3<T>& a, const V3<T>& b, const V3<T>& c, T prm[3]) {
  prm[0] = prm[1] = prm[2] = T(0);
  const V3<T> dl0 = a - b, dl1 = b - c, dl2 = c - a;
  const V3<T> n = cross(dl0, dl1);
  const T l = sqnorm(n);
  if (!(l > T(0))) return T(-1);
  T mindist = T(-1);
  // edge i = (vt[i], vt[i+1]) ; unrolled for i = 0,1,2 (nexti = {1,2,0})
  {
    if (dot(a, cross(dl0, n)) > T(0)) {
      T q0, q1, sq;
      project_line_origin(a, b, q0, q1, sq);
      if (mindist < T(0) || sq < mindist) {
        mindist = sq;
        prm[0] = q0;
        prm[1] = q1;
        prm[2] = T(0);
      }
    }
    if (dot(b, cross(dl1, n)) > T(0)) {
      T q0, q1, sq;
      project_line_origin(b, c, q0, q1, sq);
      if (mindist < T(0) || sq < mindist) {
        mindist = sq;
        prm[1] = q0;
        prm[2] = q1;
        prm[0] = T(0);
      }
    }
    if (dot(c, cross(dl2, n)) > T(0)) {
      T q0, q1, sq;
      project_line_origin(c, a, q0, q1, sq);
      if (mindist < T(0) || sq < mindist) {
        mindist = sq;
        prm[2] = q0;
        prm[0] = q1;
        prm[1] = T(0);
      }
    }
  }
  if (mindist < T(0)) {
    const T d = dot(a, n);
    const T s = hsqrt(l);
    const V3<T> o = n * (d / l);
    mindist = sqnorm(o);
    prm[0] = norm(cross(dl1, b - o)) / s;
    prm[1] = norm(cross(dl2, c - o)) / s;
    prm[2] = T(1) - prm[0] - prm[1];
  }
  return mindist;
}

// Project::projectTetrahedraOrigin (src/intersect.cpp:648-705); only the parameterisation.
// The loop over the three faces (a,b,d), (b,c,d), (c,a,d) is written out so that no array is
// indexed dynamically (registers only).
template <typename T>
HFCL_HD void project_tetra_origin(const V3<T>& a, const V3<T>& b, const V3<T>& c, const V3<T>& d, T prm[4]) {
  prm[0] = prm[1] = prm[2] = prm[3] = T(0);
  const V3<T> dl0 = a - d, dl1 = b - d, dl2 = c - d;
  const T vl = triple(dl0, dl1, dl2);
  const bool ng = (vl * dot(a, cross(b - c, a - b))) <= T(0);
  if (ng && habs(vl) > T(0)) {
    T mindist = T(-1);
    if (vl * dot(d, cross(dl0, dl1)) > T(0)) {  // i = 0, j = 1
      T q[3];
      const T sq = project_triangle_origin(a, b, d, q);
      if (mindist < T(0) || sq < mindist) {
        mindist = sq;
        prm[0] = q[0]; prm[1] = q[1]; prm[2] = T(0); prm[3] = q[2];
      }
    }
    if (vl * dot(d, cross(dl1, dl2)) > T(0)) {  // i = 1, j = 2
      T q[3];
      const T sq = project_triangle_origin(b, c, d, q);
      if (mindist < T(0) || sq < mindist) {
        mindist = sq;
        prm[1] = q[0]; prm[2] = q[1]; prm[0] = T(0); prm[3] = q[2];
      }
    }
    if (vl * dot(d, cross(dl2, dl0)) > T(0)) {  // i = 2, j = 0
      T q[3];
      const T sq = project_triangle_origin(c, a, d, q);
      if (mindist < T(0) || sq < mindist) {
        mindist = sq;
        prm[2] = q[0]; prm[0] = q[1]; prm[1] = T(0); prm[3] = q[2];
      }
    }
    if (mindist < T(0)) {
      prm[0] = triple(c, b, d) / vl;
      prm[1] = triple(a, c, d) / vl;
      prm[2] = triple(b, a, d) / vl;
      prm[3] = T(1) - (prm[0] + prm[1] + prm[2]);
    }
  } else if (!ng) {
    T q[3];
    project_triangle_origin(a, b, c, q);
    prm[0] = q[0];
    prm[1] = q[1];
    prm[2] = q[2];
    prm[3] = T(0);
  }
}

// getClosestPoints for rank 1..3.  (va,vb,vc) = simplex.vertex[0..2] (oldest first), each given
// as w / w0 / w1.  Plain V3 arguments (no arrays) so everything stays in registers.
template <typename T>
HFCL_HD void closest_points(int rank, const V3<T>& aw, const V3<T>& bw, const V3<T>& cw, const V3<T>& a0, const V3<T>& b0,
                            const V3<T>& c0, const V3<T>& a1, const V3<T>& b1, const V3<T>& c1, V3<T>& w0, V3<T>& w1) {
  if (rank == 1) {
    w0 = a0;
    w1 = a1;
    return;
  }
  if (rank == 2) {
    const V3<T> N = bw - aw;
    T la = dot(N, -aw);
    if (la <= T(0)) {
      w0 = a0;
      w1 = a1;
    } else {
      T lb = sqnorm(N);
      if (la > lb) {
        w0 = b0;
        w1 = b1;
      } else {
        lb = la / lb;
        la = T(1) - lb;
        w0 = la * a0 + lb * b0;
        w1 = la * a1 + lb * b1;
      }
    }
    return;
  }
  T prm[3];
  project_triangle_origin(aw, bw, cw, prm);
  const V3<T> z = mk<T>(T(0), T(0), T(0));
  w0 = ((z + prm[0] * a0) + prm[1] * b0) + prm[2] * c0;
  w1 = ((z + prm[0] * a1) + prm[1] * b1) + prm[2] * c1;
}

// GJK::getWitnessPointsAndNormal (gjk.cpp:177-186) + details::inflate (:158-173), shape-0 frame
template <typename T>
HFCL_HD void gjk_witness_normal(const V3<T>& ray, T r0, T r1, V3<T>& w0, V3<T>& w1, V3<T>& normal) {
  const V3<T> d = w1 - w0;
  if (norm(d) > Lim<T>::dummy())
    normal = normalized(d);
  else
    normal = -normalized(ray);
  if (r0 > T(0)) w0 = w0 + r0 * normal;
  if (r1 > T(0)) w1 = w1 - r1 * normal;
}

// GJKExtractWitnessPointsAndNormal / EPAExtract... (narrowphase.h:610-636, 658-711): to world frame
template <typename T>
HFCL_HD void to_world(const Pose<T>& tf1, T distance, V3<T>& p1, V3<T>& p2, V3<T>& normal) {
  const V3<T> p = xform(tf1, T(0.5) * (p1 + p2));
  normal = mul(tf1.R, normal);
  p1 = p - (T(0.5) * distance) * normal;
  p2 = p + (T(0.5) * distance) * normal;
}

}  // namespace hfcl
