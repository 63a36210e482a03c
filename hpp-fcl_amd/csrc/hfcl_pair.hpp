// hfcl_pair.hpp -- per-pair driver shared by every kernel: the GJK loop, the status switch of
// GJKSolver::runGJKAndEPA (/root/reference/include/hpp/fcl/narrowphase/narrowphase.h:420-723)
// and the result-record semantics of ShapeShapeDistancer::run / ShapeShapeCollider::run
// (include/hpp/fcl/internal/shape_shape_func.h:53-70, 134-163).
//
// Everything is templated on the scalar T and on a support evaluator `Sup` with
//     void operator()(const V3<T>& dir, V3<T>& w, V3<T>& w0)
// (MinkowskiDiff::support, minkowski_difference.cpp:47-63: w0 = support of shape 0 along dir,
// w = w0 - support of shape 1 along -dir, both in shape 0's frame).  The HIP kernels plug in
// register/lane-group evaluators; tests/hostsim plugs in a serial one to validate this exact
// code on the CPU.
#pragma once
#include "hfcl_epa.hpp"
#include "hfcl_gjk.hpp"
#include "hfcl_shapes.hpp"

namespace hfcl {

template <typename T>
struct QParams {
  GjkParams<T> gjk;
  T epa_tolerance;
  int epa_max_iterations;
  int compute_penetration;
  int mode;  // 0 = distance(), 1 = collide()
  T security_margin;
  T collision_distance_threshold;
  int guess_mode;  // HFCL_GUESS_* (0 default, 1 cached)
  T guess[3];
};

// MinkowskiDiff::set (minkowski_difference.cpp:269-285)
template <typename T>
struct MDiff {
  M3<T> oR1;
  V3<T> ot1;
  bool identity;
};
template <typename T>
HFCL_HD MDiff<T> make_mdiff(const Pose<T>& tf0, const Pose<T>& tf1) {
  MDiff<T> m;
  m.oR1 = tmul(tf0.R, tf1.R);
  m.ot1 = tmul(tf0.R, tf1.t - tf0.t);
  m.identity = is_identity(m.oR1) && is_zero(m.ot1);
  return m;
}

// GJKInitialGuess::BoundingVolumeGuess (narrowphase.h:366-378): centre of shape 0's local AABB minus the centre of
// shape 1's, in the frame of shape 0.  The local AABBs of the primitives are symmetric (centre 0); a ConvexBase
// carries the centre of its vertices' box in p0..p2 (filled by hfcl_lib_create).
template <typename T>
HFCL_HD V3<T> bv_center(const DShape<T>& s) {
  return s.kind == K_CONVEX ? mk<T>(s.p0, s.p1, s.p2) : mk<T>(T(0), T(0), T(0));
}
template <typename T>
HFCL_HD V3<T> start_guess(const QParams<T>& q, const DShape<T>& a, const DShape<T>& b, const MDiff<T>& md,
                                             const V3<T>& cached0) {
  if (q.guess_mode != HFCL_GUESS_BOUNDING_VOLUME) return cached0;
  return bv_center(a) - (mul(md.oR1, bv_center(b)) + md.ot1);
}

// What one finished query reports (world frame).
template <typename T>
struct PairOut {
  T distance;
  V3<T> normal, p1, p2;
  V3<T> cached_guess;
  int gjk_status, epa_status, gjk_iters, epa_iters;
};

// GJK's final simplex handed to EPA (reference order: oldest vertex first).
template <typename T>
struct EpaSeed {
  uint32_t pair;
  int32_t rank;
  V3<T> w[4], w0[4];
  V3<T> guess;
  uint32_t gjk_iters;
};

// Where the simplex vertices keep their witness data (the support point on shape 0, needed only after the loop):
// W0Regs carries it with the vertex in registers (payload PW0); the HIP kernels can plug in a policy that parks
// it outside the register file and carries a slot number instead (hfcl_dev.hpp: W0Lds).
//   P put(g, w0)  payload for the vertex about to be appended to g (g.rank vertices are live at that point)
//   V3 get(p)     the support point of a live vertex
template <typename T>
struct W0Regs {
  typedef PW0<T> P;
  template <class G> HFCL_HD P put(const G&, const V3<T>& w0) const { return P{w0}; }
  HFCL_HD V3<T> get(const P& p) const { return p.w0; }
};

template <typename T, class P, class Sup, class PS>
HFCL_HD void gjk_run(Gjk<T, P>& g, const GjkParams<T>& prm, const V3<T>& guess, T ssr_sum, bool normalize, Sup& sup, const PS& ps) {
  gjk_init(g, prm, guess, ssr_sum, normalize);
  while (!g.done) {
    V3<T> sd;
    if (gjk_begin(g, prm, sd)) {
      SimplexV<T, P> v;
      V3<T> w0;
      sup(sd, v.w, w0);
      v.p = ps.put(g, w0);
      gjk_end(g, prm, v);
    }
  }
}
template <typename T, class Sup>
HFCL_HD void gjk_run(Gjk<T, PW0<T>>& g, const GjkParams<T>& prm, const V3<T>& guess, T ssr_sum, bool normalize, Sup& sup) {
  gjk_run(g, prm, guess, ssr_sum, normalize, sup, W0Regs<T>());
}

// The pose of shape 1 is needed once, for the record: a caller may pass the pose itself or something that produces it then
// (a batch kernel re-reads it instead of carrying 12 scalars through the whole GJK / EPA loop).
template <typename T>
HFCL_HD const Pose<T>& pose_now(const Pose<T>& p) { return p; }
template <typename T, class F>
HFCL_HD auto pose_now(const F& f) -> decltype(f()) { return f(); }

// Returns true when the pair must go through EPA (seed filled); otherwise `out` is final.
template <typename T, class P, class PS, class TF>
HFCL_HD bool gjk_finish(const Gjk<T, P>& g, const QParams<T>& q, const TF& tf1, T r0, T r1,
                        const V3<T>& guess0, PairOut<T>& out, EpaSeed<T>& seed, const PS& ps) {
  typedef SimplexV<T, P> SV;
  const T nanv = Lim<T>::nan();
  const V3<T> nan3 = mk<T>(nanv, nanv, nanv);
  const int st = g.status;
  const int r = g.rank;
  out.gjk_status = st;
  out.epa_status = EPA_DID_NOT_RUN;
  out.gjk_iters = g.iterations;
  out.epa_iters = 0;
  if (st == GJK_COLLISION && q.compute_penetration) {
    // newest-first registers -> reference order: ref[i] = s[rank-1-i]
    const SV ref0 = svsel(r == 1, g.s0, svsel(r == 2, g.s1, svsel(r == 3, g.s2, g.s3)));
    const SV ref1 = svsel(r == 2, g.s0, svsel(r == 3, g.s1, g.s2));
    const SV ref2 = svsel(r == 3, g.s0, g.s1);
    seed.rank = r;
    seed.w[0] = ref0.w; seed.w0[0] = ps.get(ref0.p);
    seed.w[1] = ref1.w; seed.w0[1] = ps.get(ref1.p);
    seed.w[2] = ref2.w; seed.w0[2] = ps.get(ref2.p);
    seed.w[3] = g.s0.w; seed.w0[3] = ps.get(g.s0.p);
    seed.guess = guess0;
    seed.gjk_iters = uint32_t(g.iterations);
    return true;
  }
  if (st == GJK_EARLY_STOPPED || st == GJK_COLLISION) {
    // early stop: lower bound only (narrowphase.h:589-608); Collision w/o penetration: :638-656
    out.distance = g.distance;
    out.normal = out.p1 = out.p2 = nan3;
    out.cached_guess = (st == GJK_COLLISION) ? guess0 : g.ray;
    return false;
  }
  // NoCollision / CollisionWithPenetrationInformation / Failed: GJKExtractWitnessPointsAndNormal :610-636
  const SV ref0 = svsel(r == 1, g.s0, svsel(r == 2, g.s1, g.s2));
  const SV ref1 = svsel(r == 2, g.s0, g.s1);
  const V3<T> a0 = ps.get(ref0.p), b0 = ps.get(ref1.p), c0 = ps.get(g.s0.p);
  V3<T> p1, p2, n;
  closest_points(r, ref0.w, ref1.w, g.s0.w, a0, b0, c0, a0 - ref0.w, b0 - ref1.w, c0 - g.s0.w, p1, p2);
  gjk_witness_normal(g.ray, r0, r1, p1, p2, n);
  to_world(pose_now<T>(tf1), g.distance, p1, p2, n);
  out.distance = g.distance;
  out.normal = n;
  out.p1 = p1;
  out.p2 = p2;
  out.cached_guess = g.ray;
  return false;
}
template <typename T>
HFCL_HD bool gjk_finish(const Gjk<T, PW0<T>>& g, const QParams<T>& q, const Pose<T>& tf1, T r0, T r1,
                        const V3<T>& guess0, PairOut<T>& out, EpaSeed<T>& seed) {
  return gjk_finish(g, q, tf1, r0, r1, guess0, out, seed, W0Regs<T>());
}

// EPAExtractWitnessPointsAndNormal / EPAFailedExtractWitnessPointsAndNormal (narrowphase.h:658-723)
template <typename T>
HFCL_HD void epa_finish(const EpaResult<T>& res, uint32_t gjk_iters, const Pose<T>& tf1, T r0, T r1, PairOut<T>& out);
template <typename T>
HFCL_HD void epa_finish(const EpaResult<T>& res, const EpaSeed<T>& seed, const Pose<T>& tf1, T r0, T r1, PairOut<T>& out) {
  epa_finish(res, seed.gjk_iters, tf1, r0, r1, out);
}

// EPA branch of runGJKAndEPA (narrowphase.h:505-584) for one seed.  Returns 1 when `out` is final; 0 when
// the polytope outgrew the CAP-sized scratch block and must be redone by the full-capacity kernel; 2 when
// it outgrew the block at an iteration boundary: the scratch block (incl. its hdr) then describes it
// completely and the full-capacity kernel can continue it (epa_resume).
template <typename T, class Grp, int CAP, class Sup, int V0M, class TF>
HFCL_HD int epa_run(EpaScratch<T, CAP, V0M>* scratch, const EpaSeed<T>& seed, const QParams<T>& q, const TF& tf1, T r0,
                    T r1, Sup& sup, PairOut<T>& out, Quad<T>* v0_ext = nullptr) {
  static_assert(V0M != V0_TAG, "the batch form keeps coordinates");
  Epa<T, Grp, CAP, V0M> epa;
  epa.reset(scratch, q.epa_max_iterations, q.epa_tolerance, v0_ext);
  // all four slots are written (slots >= rank are scratch that encloseOrigin overwrites): constant
  // indices keep the seed in registers
  epa.set_vert(0, seed.w[0], seed.w0[0]);
  epa.set_vert(1, seed.w[1], seed.w0[1]);
  epa.set_vert(2, seed.w[2], seed.w0[2]);
  epa.set_vert(3, seed.w[3], seed.w0[3]);
  Grp::sync();
  EpaResult<T> res;
  epa.evaluate(seed.rank, -seed.guess, r0 + r1, sup, res);
  if (epa.overflow) return epa.resumable ? 2 : 0;
  epa_finish(res, seed, pose_now<T>(tf1), r0, r1, out);
  return 1;
}

// Continue a polytope a CAP_SRC-tier saved (blob) in a CAP-sized block.  CAP must be the reference capacity.
template <typename T, class Grp, int CAP_SRC, int CAP, class Sup, int V0M, class TF>
HFCL_HD void epa_resume(EpaScratch<T, CAP, V0M>* scratch, const EpaSaved<T, CAP_SRC>* blob, const EpaSeed<T>& seed,
                        const QParams<T>& q, const TF& tf1, T r0, T r1, Sup& sup, PairOut<T>& out, Quad<T>* v0_ext = nullptr) {
  Epa<T, Grp, CAP, V0M> epa;
  epa.reset(scratch, q.epa_max_iterations, q.epa_tolerance, v0_ext);
  const EpaHeader h = epa.template load<CAP_SRC>(blob);
  EpaResult<T> res;
  epa.run_loop(h.closest, h.iterations, h.pass, r0 + r1, sup, res);
  epa_finish(res, seed, pose_now<T>(tf1), r0, r1, out);
}

template <typename T>
HFCL_HD void epa_finish(const EpaResult<T>& res, uint32_t gjk_iters, const Pose<T>& tf1, T r0, T r1, PairOut<T>& out) {
  out.gjk_status = GJK_COLLISION;
  out.gjk_iters = int(gjk_iters);
  out.epa_status = res.status;
  out.epa_iters = res.iterations;
  if (res.status == EPA_FALLBACK) {  // EPAFailedExtractWitnessPointsAndNormal :713-723
    const T nanv = Lim<T>::nan();
    out.distance = -Lim<T>::max();
    out.normal = out.p1 = out.p2 = mk<T>(nanv, nanv, nanv);
    out.cached_guess = mk<T>(T(1), T(0), T(0));
    return;
  }
  // EPAExtractWitnessPointsAndNormal :658-711
  out.cached_guess = -(res.depth * res.normal);
  const T d = hmin(T(0), -res.depth);
  V3<T> p1, p2, n;
  epa_witness_normal(res, r0, r1, p1, p2, n);
  to_world(tf1, d, p1, p2, n);
  out.distance = d;
  out.normal = n;
  out.p1 = p1;
  out.p2 = p2;
}

// Record semantics on a fresh result object.  Returns the contact flag; for collide() the
// witness data are NaN when the lower bound was not updated (collision_data.h:1186-1197).
template <typename T>
HFCL_HD bool apply_query_semantics(const QParams<T>& q, PairOut<T>& o, int& num_contacts) {
  num_contacts = 0;
  if (q.mode == 1) {
    const T dtc = o.distance - q.security_margin;
    if (!(dtc < Lim<T>::max())) {
      const T x = Lim<T>::nan();
      o.normal = o.p1 = o.p2 = mk<T>(x, x, x);
    }
    const bool contact = dtc <= q.collision_distance_threshold;
    num_contacts = contact ? 1 : 0;
    return contact;
  }
  return o.distance <= T(0);
}

HFCL_HD uint32_t pack_status(int gjk, int epa, bool contact, int gi, int ei) {
  return (uint32_t(gjk) & 7u) | ((uint32_t(epa) & 15u) << 3) | (contact ? 128u : 0u) |
         (uint32_t(gi > 255 ? 255 : gi) << 8) | (uint32_t(ei > 127 ? 127 : ei) << 16);
}

// Serial support evaluator over DShape records (any kinds).  Used by the per-lane primitive
// kernel (no vertices involved) and by tests/hostsim (convex hulls scanned serially,
// getShapeSupportLinear support_functions.cpp:400-421).
template <typename T>
struct SerialSupport {
  DShape<T> a, b;
  const T* va;
  const T* vb;
  MDiff<T> md;
  HFCL_HD V3<T> one(const DShape<T>& s, const T* verts, const V3<T>& dir) const {
    if (s.kind != K_CONVEX) return prim_support(s, dir);
    int best = 0;
    T bd = verts[0] * dir.x + verts[1] * dir.y + verts[2] * dir.z;
    for (uint32_t i = 1; i < s.num_points; ++i) {
      const T d = verts[3 * i] * dir.x + verts[3 * i + 1] * dir.y + verts[3 * i + 2] * dir.z;
      if (d > bd) {
        bd = d;
        best = int(i);
      }
    }
    return mk<T>(verts[3 * best], verts[3 * best + 1], verts[3 * best + 2]);
  }
  HFCL_HD void operator()(const V3<T>& dir, V3<T>& w, V3<T>& w0) const {
    w0 = one(a, va, dir);
    const V3<T> d1 = md.identity ? -dir : -tmul(md.oR1, dir);
    V3<T> s1 = one(b, vb, d1);
    s1 = md.identity ? s1 : (mul(md.oR1, s1) + md.ot1);
    w = w0 - s1;
  }
};

}  // namespace hfcl
