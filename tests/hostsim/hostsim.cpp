// tests/hostsim -- CPU-side validation build of the *device* per-pair code.
//
// TEST INFRASTRUCTURE ONLY.  This compiles the very headers the HIP kernels are made of
// (hpp-fcl_amd/csrc/hfcl_{math,gjk,epa,shapes,pair}.hpp) with plain g++ and runs them one pair
// at a time with a serial support evaluator, so the GJK/EPA core that ships in the kernels can
// be checked against the fp64 oracle in this GPU-less container (fp64 and fp32 instantiations).
// It is NOT a fallback: the product library (csrc/libhppfcl_amd.so) neither links nor loads
// this file, and the C ABI fails with HFCL_ERR_NO_DEVICE when there is no GPU.
#include <algorithm>
#include <cstring>
#include <vector>

#include "../../include/hppfcl_amd.h"
#include "../../hpp-fcl_amd/csrc/hfcl_bvh.hpp"
#include "../../hpp-fcl_amd/csrc/hfcl_bvh_shape.hpp"
#include "../../hpp-fcl_amd/csrc/hfcl_pair.hpp"

using namespace hfcl;

template <typename T>
static DShape<T> to_dshape(const hfcl_shape& s) {
  DShape<T> d;
  d.kind = s.type;
  d.num_points = s.num_points;
  d.vertex_offset = s.vertex_offset;
  d.bvh_index = s.bvh_index;
  d.p0 = T(s.params[0]);
  d.p1 = T(s.params[1]);
  d.p2 = T(s.params[2]);
  d.p3 = T(s.params[3]);
  d.ssr = T(s.swept_sphere_radius);
  return d;
}
// as hfcl_lib_create: a ConvexBase carries the centre of its vertices' box in p0..p2 (BoundingVolumeGuess)
template <typename T>
static DShape<T> to_dshape(const hfcl_shape& s, const double* vertices) {
  DShape<T> d = to_dshape<T>(s);
  if (s.type == HFCL_GEOM_CONVEX && s.num_points > 0 && vertices) {
    double mn[3], mx[3];
    const double* v = vertices + 3 * size_t(s.vertex_offset);
    for (int k = 0; k < 3; ++k) mn[k] = mx[k] = v[k];
    for (uint32_t j = 1; j < s.num_points; ++j)
      for (int k = 0; k < 3; ++k) {
        mn[k] = std::min(mn[k], v[3 * size_t(j) + k]);
        mx[k] = std::max(mx[k], v[3 * size_t(j) + k]);
      }
    d.p0 = T((mn[0] + mx[0]) * 0.5);
    d.p1 = T((mn[1] + mx[1]) * 0.5);
    d.p2 = T((mn[2] + mx[2]) * 0.5);
  }
  return d;
}

template <typename T>
static void fill_q(QParams<T>& q, const hfcl_query_request& r) {
  q.gjk.tolerance = T(r.gjk_tolerance);
  q.gjk.max_iterations = r.gjk_max_iterations;
  q.gjk.variant = r.gjk_variant;
  q.gjk.crit = r.gjk_convergence_criterion;
  q.gjk.crit_type = r.gjk_convergence_criterion_type;
  q.epa_tolerance = T(r.epa_tolerance);
  q.epa_max_iterations = int(r.epa_max_iterations);
  q.collision_distance_threshold = T(r.collision_distance_threshold);
  q.guess_mode = r.gjk_initial_guess;
  for (int k = 0; k < 3; ++k) q.guess[k] = T(r.cached_gjk_guess[k]);
}

template <typename T>
struct TriPairSolid {  // support of the non-triangle shape of a top-level TriangleP pair, in its own frame
  DShape<T> s;
  const T* verts;
  V3<T> operator()(const V3<T>& d) const {
    SerialSupport<T> ss;
    return ss.one(s, verts + 3 * size_t(s.vertex_offset), d);
  }
};

template <typename T>
static void one_pair(const DShape<T>& a, const DShape<T>& b, const T* verts, const Pose<T>& tf1, const Pose<T>& tf2,
                     const QParams<T>& q, const V3<T>& guess0, PairOut<T>& o, bool& contact, int& nc, bool& skipped) {
  skipped = false;
  if (q.mode != 1 && (a.kind == K_TRIANGLE || b.kind == K_TRIANGLE)) {  // no TriangleP in the distance matrix
    skipped = true;
    return;
  }
  const int cls = pair_class(a.kind, b.kind);
  if (cls == CLS_CLOSED) {
    o.distance = closed_form_distance(a, tf1, b, tf2, verts, o.p1, o.p2, o.normal);
    o.gjk_status = GJK_DID_NOT_RUN;
    o.epa_status = EPA_DID_NOT_RUN;
    o.gjk_iters = o.epa_iters = 0;
    o.cached_guess = guess0;
  } else if (cls == CLS_PRIM_GJK || cls == CLS_CONVEX) {
    SerialSupport<T> sup;
    sup.a = a;
    sup.b = b;
    sup.va = verts + 3 * size_t(a.vertex_offset);
    sup.vb = verts + 3 * size_t(b.vertex_offset);
    sup.md = make_mdiff(tf1, tf2);
    const T r0 = swept_radius(a), r1 = swept_radius(b);
    Gjk<T, PW0<T>> g;
    gjk_run(g, q.gjk, start_guess(q, a, b, sup.md, guess0), r0 + r1, a.kind == K_CONVEX && b.kind == K_CONVEX, sup);
    EpaSeed<T> seed;
    if (gjk_finish(g, q, tf1, r0, r1, guess0, o, seed)) {
      // same two-tier scheme as the kernels: small-capacity block first; on overflow the polytope is saved
      // and continued in the full-capacity block (or redone there when it is not at an iteration boundary)
      static thread_local EpaScratch<T, 20> small;
      static thread_local EpaSaved<T, 20> saved;
      static thread_local EpaScratch<T, EPA_MAX_ITER> full;
      const int rc = epa_run<T, SerialGroup<1>, 20>(&small, seed, q, tf1, r0, r1, sup, o);
      if (rc == 2) {
        epa_save_block<T, SerialGroup<1>, 20>(&small, &saved);
        epa_resume<T, SerialGroup<1>, 20, EPA_MAX_ITER>(&full, &saved, seed, q, tf1, r0, r1, sup, o);
      } else if (rc == 0) {
        epa_run<T, SerialGroup<1>, EPA_MAX_ITER>(&full, seed, q, tf1, r0, r1, sup, o);
      }
    }
  } else if ((a.kind == K_TRIANGLE || b.kind == K_TRIANGLE) && cls != CLS_BVH) {
    const DShape<T>& other = (a.kind == K_TRIANGLE) ? b : a;
    if (!(other.kind == K_TRIANGLE || kind_is_prim(other.kind) || other.kind == K_CONVEX)) {
      skipped = true;
      return;
    }
    static thread_local EpaScratch<T, EPA_MAX_ITER> full;
    TriPairSolid<T> hs{other, verts};
    triangle_pair<T, SerialGroup<1>>(a, b, verts, tf1, tf2, hs, q, guess0, &full, o);
  } else {
    skipped = true;
    return;
  }
  contact = apply_query_semantics(q, o, nc);
}

extern "C" {

// fp64: same signature family as the C ABI's host entry points (no library object).
int sim_batch_f64(const hfcl_shape* shapes, size_t n_shapes, const double* vertices, const uint32_t* s1,
                  const uint32_t* s2, const double* tf1, const double* tf2, size_t n, const hfcl_collision_request* creq,
                  const hfcl_distance_request* dreq, hfcl_result* out, const hfcl_guess* gin, hfcl_guess* gout) {
  std::vector<DShape<double>> lib(n_shapes);
  for (size_t i = 0; i < n_shapes; ++i) lib[i] = to_dshape<double>(shapes[i], vertices);
  QParams<double> q;
  if (creq) {
    fill_q(q, creq->q);
    q.mode = 1;
    q.compute_penetration = (creq->enable_contact || creq->security_margin < 0) ? 1 : 0;
    q.security_margin = creq->security_margin;
    double ub = creq->distance_upper_bound > creq->security_margin ? creq->distance_upper_bound : creq->security_margin;
    q.gjk.distance_upper_bound = ub < 0 ? 0 : ub;
  } else {
    fill_q(q, dreq->q);
    q.mode = 0;
    q.compute_penetration = dreq->enable_signed_distance ? 1 : 0;
    q.security_margin = 0;
    q.gjk.distance_upper_bound = Lim<double>::max();
  }
  for (size_t i = 0; i < n; ++i) {
    V3<double> g0 = mk<double>(1, 0, 0);
    if (q.guess_mode == HFCL_GUESS_BOUNDING_VOLUME) g0 = mk<double>(q.guess[0], q.guess[1], q.guess[2]);
    if (q.guess_mode == HFCL_GUESS_CACHED)
      g0 = gin ? mk<double>(gin[i].gjk_guess[0], gin[i].gjk_guess[1], gin[i].gjk_guess[2])
               : mk<double>(q.guess[0], q.guess[1], q.guess[2]);
    PairOut<double> o;
    bool contact = false, skipped = false;
    int nc = 0;
    one_pair<double>(lib[s1[i]], lib[s2[i]], vertices, pose_from_abi<double>(tf1 + 12 * i),
                     pose_from_abi<double>(tf2 + 12 * i), q, g0, o, contact, nc, skipped);
    hfcl_result& r = out[i];
    std::memset(&r, 0, sizeof(r));
    if (skipped) {
      r.status = 0x80000000u;
      continue;
    }
    r.distance = o.distance;
    r.normal[0] = o.normal.x; r.normal[1] = o.normal.y; r.normal[2] = o.normal.z;
    r.p1[0] = o.p1.x; r.p1[1] = o.p1.y; r.p1[2] = o.p1.z;
    r.p2[0] = o.p2.x; r.p2[1] = o.p2.y; r.p2[2] = o.p2.z;
    r.b1 = r.b2 = -1;
    r.status = pack_status(o.gjk_status, o.epa_status, contact, o.gjk_iters, o.epa_iters);
    r.num_contacts = nc;
    if (gout) {
      gout[i].gjk_guess[0] = o.cached_guess.x;
      gout[i].gjk_guess[1] = o.cached_guess.y;
      gout[i].gjk_guess[2] = o.cached_guess.z;
      gout[i].support_guess[0] = gout[i].support_guess[1] = 0;
    }
  }
  return 0;
}

// fp32: 7-float poses (quat wxyz + translation), 44-byte records, as hfcl_*_batch_device_f32.
int sim_batch_f32(const hfcl_shape* shapes, size_t n_shapes, const double* vertices, size_t n_vertices,
                  const uint32_t* s1, const uint32_t* s2, const float* pose1, const float* pose2, size_t n,
                  const hfcl_collision_request* creq, const hfcl_distance_request* dreq, hfcl_result_f32* out) {
  std::vector<DShape<float>> lib(n_shapes);
  for (size_t i = 0; i < n_shapes; ++i) lib[i] = to_dshape<float>(shapes[i]);
  std::vector<float> v32(3 * n_vertices + 3);
  for (size_t i = 0; i < 3 * n_vertices; ++i) v32[i] = float(vertices[i]);
  QParams<float> q;
  if (creq) {
    fill_q(q, creq->q);
    q.mode = 1;
    q.compute_penetration = (creq->enable_contact || creq->security_margin < 0) ? 1 : 0;
    q.security_margin = float(creq->security_margin);
    double ub = creq->distance_upper_bound > creq->security_margin ? creq->distance_upper_bound : creq->security_margin;
    if (ub < 0) ub = 0;
    q.gjk.distance_upper_bound = ub >= double(Lim<float>::max()) ? Lim<float>::max() : float(ub);
  } else {
    fill_q(q, dreq->q);
    q.mode = 0;
    q.compute_penetration = dreq->enable_signed_distance ? 1 : 0;
    q.security_margin = 0;
    q.gjk.distance_upper_bound = Lim<float>::max();
  }
  for (size_t i = 0; i < n; ++i) {
    V3<float> g0 = mk<float>(1, 0, 0);
    if (q.guess_mode == HFCL_GUESS_CACHED) g0 = mk<float>(q.guess[0], q.guess[1], q.guess[2]);
    PairOut<float> o;
    bool contact = false, skipped = false;
    int nc = 0;
    one_pair<float>(lib[s1[i]], lib[s2[i]], v32.data(), pose_from_quat<float>(pose1 + 7 * i),
                    pose_from_quat<float>(pose2 + 7 * i), q, g0, o, contact, nc, skipped);
    hfcl_result_f32& r = out[i];
    std::memset(&r, 0, sizeof(r));
    if (skipped) {
      r.status = 0x80000000u;
      continue;
    }
    r.distance = o.distance;
    r.p1[0] = o.p1.x; r.p1[1] = o.p1.y; r.p1[2] = o.p1.z;
    r.p2[0] = o.p2.x; r.p2[1] = o.p2.y; r.p2[2] = o.p2.z;
    r.normal[0] = o.normal.x; r.normal[1] = o.normal.y; r.normal[2] = o.normal.z;
    r.status = pack_status(o.gjk_status, o.epa_status, contact, o.gjk_iters, o.epa_iters);
  }
  return 0;
}

// The staged convex x convex fast tier (hfcl_epa.hpp: EpaReady): for every pair whose GJK ends in a seed of rank 4,
// epa_prepare_tetrahedron + Epa::install must leave the scratch block and the solver fields exactly as Epa::begin does
// (or both fall back).  Returns the number of seeds checked; *mismatches counts the ones that differ in any byte.
namespace {
struct NoSupportTagged {  // encloseOrigin evaluates no support for a seed of rank 4
  void operator()(const V3<float>&, V3<float>&, V3<float>&, int&) const {}
};
struct ZeroTags {
  V3<float> operator()(int) const { return mk<float>(0.f, 0.f, 0.f); }
};
}  // namespace
static bool epa_prepare_check_one(const EpaSeed<float>& seed, const QParams<float>& q, long* fallbacks, float* feat) {
  constexpr int CAP = 17;
  typedef Epa<float, SerialGroup<1>, CAP, V0_TAG> E;
  static EpaScratch<float, CAP, V0_TAG> blk_a, blk_b;
  std::memset(&blk_a, 0xA5, sizeof(blk_a));
  std::memset(&blk_b, 0xA5, sizeof(blk_b));
  E ea;
  ea.reset(&blk_a, q.epa_max_iterations, q.epa_tolerance);
  for (int k = 0; k < 4; ++k) ea.set_vert(k, seed.w[k], seed.w0[k], -1 - k);
  NoSupportTagged ns;
  EpaResult<float> res;
  const int closest_a = ea.begin(seed.rank, -seed.guess, ns, res, ZeroTags());
  EpaReady<float> rb;
  int flags[4], closest_b = 0;
  const bool live = epa_prepare_tetrahedron(seed.w, q.epa_tolerance, rb.vw, rb.fn, flags, closest_b);
  if (!live || closest_a == EPA_NULL) {
    ++*fallbacks;
    return live == (closest_a != EPA_NULL);
  }
  if (feat) {  // what k_epa_prepare knows about the polytope before its loop (tools: predictors of the length of the loop)
    float dmin = Lim<float>::max(), dmax = 0.f;
    for (int f = 0; f < 4; ++f)
      if (!(flags[f] & 2)) {
        dmin = std::min(dmin, rb.fn[f].w);
        dmax = std::max(dmax, rb.fn[f].w);
      }
    feat[0] = dmin;
    feat[1] = dmax;
    feat[2] = habs(triple(seed.w[0] - seed.w[3], seed.w[1] - seed.w[3], seed.w[2] - seed.w[3]));
    feat[3] = float(seed.gjk_iters);
    float e = 0.f;
    for (int a2 = 0; a2 < 4; ++a2)
      for (int b2 = a2 + 1; b2 < 4; ++b2) e = std::max(e, norm(seed.w[a2] - seed.w[b2]));
    feat[4] = e;
    feat[5] = float((flags[0] >> 1) + (flags[1] >> 1) + (flags[2] >> 1) + (flags[3] >> 1));
  }
  rb.packed = uint32_t(closest_b) << 12;
  for (int f = 0; f < 4; ++f) rb.packed |= uint32_t((flags[f] >> 1) & 1) << (14 + f);
  E eb;
  eb.reset(&blk_b, q.epa_max_iterations, q.epa_tolerance);
  const int closest_i = eb.install(&rb, rb.packed);
  bool same = closest_i == closest_a && ea.status == eb.status && ea.num_vertices == eb.num_vertices && ea.hull_count == eb.hull_count &&
              ea.stock_top == eb.stock_top && ea.stamp == eb.stamp && ea.hw == eb.hw && ea.pending_release == eb.pending_release;
  same = same && std::memcmp(blk_a.vw, blk_b.vw, 4 * sizeof(Quad<float>)) == 0 && std::memcmp(blk_a.fn, blk_b.fn, 4 * sizeof(Quad<float>)) == 0 &&
         std::memcmp(blk_a.ft, blk_b.ft, 4 * sizeof(FaceTopo)) == 0;
  // the part of the stock that is still in use, and the flags of the unused faces
  same = same && std::memcmp(blk_a.stock, blk_b.stock, size_t(ea.stock_top)) == 0;
  for (int f = 4; f < 2 * CAP + 4; ++f) same = same && blk_a.ft[f].flag() == blk_b.ft[f].flag();
  return same;
}
long sim_epa_prepare_selftest(const hfcl_shape* shapes, size_t n_shapes, const double* vertices, size_t n_vertices, const uint32_t* s1,
                              const uint32_t* s2, const float* pose1, const float* pose2, size_t n, const hfcl_distance_request* dreq,
                              long* mismatches, long* fallbacks, float* features /* optional: 6 per pair */) {
  std::vector<DShape<float>> lib(n_shapes);
  for (size_t i = 0; i < n_shapes; ++i) lib[i] = to_dshape<float>(shapes[i]);
  std::vector<float> v32(3 * n_vertices + 3);
  for (size_t i = 0; i < 3 * n_vertices; ++i) v32[i] = float(vertices[i]);
  QParams<float> q;
  fill_q(q, dreq->q);
  q.mode = 0;
  q.compute_penetration = 1;
  q.security_margin = 0;
  q.gjk.distance_upper_bound = Lim<float>::max();
  long checked = 0;
  *mismatches = 0;
  *fallbacks = 0;
  for (size_t i = 0; i < n; ++i) {
    const DShape<float>&a = lib[s1[i]], &b = lib[s2[i]];
    SerialSupport<float> sup;
    sup.a = a;
    sup.b = b;
    sup.va = v32.data() + 3 * size_t(a.vertex_offset);
    sup.vb = v32.data() + 3 * size_t(b.vertex_offset);
    const Pose<float> tf1 = pose_from_quat<float>(pose1 + 7 * i), tf2 = pose_from_quat<float>(pose2 + 7 * i);
    sup.md = make_mdiff(tf1, tf2);
    const V3<float> g0 = mk<float>(1, 0, 0);
    Gjk<float, PW0<float>> g;
    gjk_run(g, q.gjk, g0, 0.f, true, sup);
    PairOut<float> o;
    EpaSeed<float> seed;
    if (features)
      for (int k = 0; k < 6; ++k) features[6 * i + k] = Lim<float>::nan();
    if (!gjk_finish(g, q, tf1, 0.f, 0.f, g0, o, seed) || seed.rank != 4) continue;
    ++checked;
    if (!epa_prepare_check_one(seed, q, fallbacks, features ? features + 6 * i : nullptr)) ++*mismatches;
  }
  return checked;
}
// ... on tetrahedra given as they are (w: 12 floats each): flat, inverted, origin outside -- what GJK's seeds rarely are
long sim_epa_prepare_selftest_tetrahedra(const float* w, size_t n, float tolerance, long* mismatches, long* fallbacks) {
  QParams<float> q;
  std::memset(&q, 0, sizeof(q));
  q.epa_tolerance = tolerance;
  q.epa_max_iterations = 64;
  *mismatches = 0;
  *fallbacks = 0;
  for (size_t i = 0; i < n; ++i) {
    EpaSeed<float> seed;
    seed.pair = uint32_t(i);
    seed.rank = 4;
    for (int k = 0; k < 4; ++k) {
      seed.w[k] = mk<float>(w[12 * i + 3 * k], w[12 * i + 3 * k + 1], w[12 * i + 3 * k + 2]);
      seed.w0[k] = seed.w[k];
    }
    seed.guess = mk<float>(1.f, 0.f, 0.f);
    seed.gjk_iters = 0;
    if (!epa_prepare_check_one(seed, q, fallbacks, nullptr)) ++*mismatches;
  }
  return long(n);
}

}  // extern "C"

// BVHModel<OBBRSS> collide through the device headers' BV test / leaf test with a serial DFS
// (same push order and stop rule as k_bvh_collide).  mesh_table: (node_off, n_nodes, vert_off, tri_off).
// fnodes != nullptr: the walk of k_bvh_collide<double, ., FILT>: the fp32 filter (hfcl_bvh.hpp: obb_filter) decides where it
// can prove the fp64 test's outcome, the fp64 test runs elsewhere; sizes are compared through the library-wide ranks.
// fstats (7 counters): BV tests, filter says overlap, filter says disjoint and the bound makes the value irrelevant,
// disjoint but the exact value is needed, unsure, UNSAFE verdicts (filter contradicted by the fp64 test: must stay 0),
// rank comparisons that differ from the fp64 size comparison (must stay 0).
template <typename T>
static void bvh_pair(const std::vector<DNode<T>>& nodes, const std::vector<T>& verts, const uint32_t* tris,
                     const uint64_t* m1, const uint64_t* m2, const Pose<T>& tf1, const Pose<T>& tf2, const QParams<T>& q,
                     uint32_t num_max_contacts, T break_distance2, hfcl_result& r, std::vector<hfcl_contact>* contacts,
                     uint32_t pair, const DNodeF* fnodes = nullptr, uint64_t* fstats = nullptr) {
  const M3<T> RT_R = tmul(tf1.R, tf2.R);
  const V3<T> RT_T = tmul(tf1.R, tf2.t - tf1.t);
  M3<float> RT_Rf;
  RT_Rf.r0 = mk<float>(float(RT_R.r0.x), float(RT_R.r0.y), float(RT_R.r0.z));
  RT_Rf.r1 = mk<float>(float(RT_R.r1.x), float(RT_R.r1.y), float(RT_R.r1.z));
  RT_Rf.r2 = mk<float>(float(RT_R.r2.x), float(RT_R.r2.y), float(RT_R.r2.z));
  const V3<float> RT_Tf = mk<float>(float(RT_T.x), float(RT_T.y), float(RT_T.z));
  const float t0mag = (fabsf(RT_Tf.x) + fabsf(RT_Tf.y) + fabsf(RT_Tf.z)) * (1.f + 4.f * OBBF_U);
  std::vector<uint32_t> stack;
  stack.push_back(0);
  uint32_t nc = 0;
  T dlb = Lim<T>::max(), rec = Lim<T>::max();
  const T nanv = Lim<T>::nan();
  V3<T> np1 = mk<T>(nanv, nanv, nanv), np2 = np1, nn = np1;
  int fb1 = -1, fb2 = -1;
  bool has_cand = false;
  uint32_t cand = 0;
  float cand_lo = 0.f, cand_hi = 0.f;
  auto resolve_cand = [&]() {  // the fp64 value of the candidate enters the bound
    if (!has_cand) return;
    has_cand = false;
    const uint32_t c1 = cand & 0xFFFFu, c2 = cand >> 16;
    T sqc;
    const bool dj = obb_disjoint(RT_R, RT_T, nodes[m2[0] + c2], nodes[m1[0] + c1], q.security_margin, break_distance2, sqc);
    ++fstats[3];
    if (!dj) ++fstats[5];
    if (!(dlb <= T(0))) {
      const T nd = hsqrt(sqc);
      if (nd < dlb) { dlb = nd; rec = nd + q.security_margin; }
    }
  };
  while (!stack.empty()) {
    const uint32_t e = stack.back();
    stack.pop_back();
    const uint32_t b1 = e & 0xFFFFu, b2 = e >> 16;
    const DNode<T>& n1 = nodes[m1[0] + b1];
    const DNode<T>& n2 = nodes[m2[0] + b2];
    const bool l1 = n1.first_child < 0, l2 = n2.first_child < 0;
    if (l1 && l2) {
      if (fnodes) resolve_cand();  // the leaf's distance is compared with the bound: the bound must be exact
      const uint32_t p1i = uint32_t(-(n1.first_child + 1)), p2i = uint32_t(-(n2.first_child + 1));
      const uint32_t* t1 = tris + 3 * (m1[3] + p1i);
      const uint32_t* t2 = tris + 3 * (m2[3] + p2i);
      auto vtx = [&](uint64_t off, uint32_t i) { return mk<T>(verts[3 * (off + i)], verts[3 * (off + i) + 1], verts[3 * (off + i) + 2]); };
      TriSupport<T> tri;
      tri.p1 = xform(tf1, vtx(m1[2], t1[0])); tri.p2 = xform(tf1, vtx(m1[2], t1[1])); tri.p3 = xform(tf1, vtx(m1[2], t1[2]));
      tri.q1 = xform(tf2, vtx(m2[2], t2[0])); tri.q2 = xform(tf2, vtx(m2[2], t2[1])); tri.q3 = xform(tf2, vtx(m2[2], t2[2]));
      V3<T> p1, p2, n;
      int gst, git;
      const T d = tri_tri_distance(tri, q.gjk, q.guess_mode == HFCL_GUESS_CACHED, mk<T>(q.guess[0], q.guess[1], q.guess[2]),
                                   p1, p2, n, gst, git);
      const T dtc = d - q.security_margin;
      if (dtc < dlb) { dlb = dtc; rec = d; np1 = p1; np2 = p2; nn = n; }
      if (dtc <= q.collision_distance_threshold) {
        if (nc < num_max_contacts) {
          if (nc == 0) { fb1 = int(p1i); fb2 = int(p2i); }
          ++nc;
          if (contacts) {
            hfcl_contact c;
            c.pair = pair; c.b1 = int(p1i); c.b2 = int(p2i); c._pad = 0; c.penetration_depth = d;
            c.normal[0] = n.x; c.normal[1] = n.y; c.normal[2] = n.z;
            c.p1[0] = p1.x; c.p1[1] = p1.y; c.p1[2] = p1.z; c.p2[0] = p2.x; c.p2[1] = p2.y; c.p2[2] = p2.z;
            contacts->push_back(c);
          }
        }
        if (nc >= num_max_contacts) break;
      }
      continue;
    }
    T sq;
    bool disjoint;
    if (fnodes) {
      // the state machine of k_bvh_collide<double, ., true>: a disjoint pair whose value could be the new minimum of the
      // bound becomes THE candidate (cand, [cand_lo, cand_hi]); its exact value is computed only when something has to be
      // compared with it (a leaf test, a second candidate it cannot be told apart from, the end of a contact-free walk)
      float nd_lo, nd_hi;
      const int v = obb_filter(RT_Rf, RT_Tf, t0mag, fnodes[m2[0] + b2], fnodes[m1[0] + b1], float(q.security_margin),
                               float(break_distance2), nd_lo, nd_hi);
      const bool dj64 = obb_disjoint(RT_R, RT_T, n2, n1, q.security_margin, break_distance2, sq);  // (the checker)
      ++fstats[0];
      if (v == OBBF_OVERLAP) {
        ++fstats[1];
        if (dj64) ++fstats[5];
        disjoint = false;
      } else if (v == OBBF_DISJOINT) {
        if (!dj64 || !(double(hsqrt(sq)) >= double(nd_lo)) || !(double(hsqrt(sq)) <= double(nd_hi))) ++fstats[5];
        if (dlb <= T(0) || T(nd_lo) >= dlb || (has_cand && nd_lo >= cand_hi)) {
          ++fstats[2];  // nothing would change: the value is not needed
        } else if (!has_cand || nd_hi < cand_lo) {
          has_cand = true;  // (a candidate that is certainly above this one is dropped)
          cand = e;
          cand_lo = nd_lo;
          cand_hi = nd_hi;
        } else {  // two candidates that cannot be told apart: the old one is resolved, this pair is looked at again
          resolve_cand();
          stack.push_back(e);
        }
        continue;
      } else {
        ++fstats[4];
        disjoint = dj64;
      }
    } else {
      disjoint = obb_disjoint(RT_R, RT_T, n2, n1, q.security_margin, break_distance2, sq);
    }
    if (disjoint) {
      if (!(dlb <= T(0))) {
        const T nd = hsqrt(sq);
        if (nd < dlb) { dlb = nd; rec = nd + q.security_margin; }
      }
      continue;
    }
    bool first = l2 || (!l1 && (sqnorm(n1.extent) > sqnorm(n2.extent)));
    if (fnodes) {
      const bool first_r = l2 || (!l1 && ((fnodes[m1[0] + b1].rank & OBBF_RANK_MASK) > (fnodes[m2[0] + b2].rank & OBBF_RANK_MASK)));
      if (first_r != first) ++fstats[6];
      first = first_r;
    }
    if (first) {
      const uint32_t c1 = uint32_t(n1.first_child);
      stack.push_back((c1 + 1) | (b2 << 16));
      stack.push_back(c1 | (b2 << 16));
    } else {
      const uint32_t c1 = uint32_t(n2.first_child);
      stack.push_back(b1 | ((c1 + 1) << 16));
      stack.push_back(b1 | (c1 << 16));
    }
  }
  if (fnodes && nc == 0) resolve_cand();  // a contact-free walk reports its bound
  r.distance = rec;
  r.normal[0] = nn.x; r.normal[1] = nn.y; r.normal[2] = nn.z;
  r.p1[0] = np1.x; r.p1[1] = np1.y; r.p1[2] = np1.z;
  r.p2[0] = np2.x; r.p2[1] = np2.y; r.p2[2] = np2.z;
  r.b1 = fb1; r.b2 = fb2;
  r.status = nc ? 128u : 0u;
  r.num_contacts = int(nc);
}

// BVHModel<OBBRSS> distance(): distanceRecurse order (nearer child first, prune on the RSS bound)
template <typename T>
static void bvh_distance_pair(const std::vector<DNode<T>>& nodes, const std::vector<DRss<T>>& rss, const std::vector<T>& verts,
                              const uint32_t* tris, const uint64_t* m1, const uint64_t* m2, const Pose<T>& tf1,
                              const Pose<T>& tf2, hfcl_result& r) {
  const M3<T> RT_R = tmul(tf1.R, tf2.R);
  const V3<T> RT_T = tmul(tf1.R, tf2.t - tf1.t);
  T mind = Lim<T>::max();
  int fb1 = -1, fb2 = -1;
  const T nanv = Lim<T>::nan();
  V3<T> np1 = mk<T>(nanv, nanv, nanv), np2 = np1;
  auto vtx = [&](uint64_t off, uint32_t i) { return mk<T>(verts[3 * (off + i)], verts[3 * (off + i) + 1], verts[3 * (off + i) + 2]); };
  auto leaf = [&](uint32_t p1i, uint32_t p2i) {
    const uint32_t* t1 = tris + 3 * (m1[3] + p1i);
    const uint32_t* t2 = tris + 3 * (m2[3] + p2i);
    V3<T> P, Q;
    const T d2 = sqr_tri_distance(vtx(m1[2], t1[0]), vtx(m1[2], t1[1]), vtx(m1[2], t1[2]),
                                  mul(RT_R, vtx(m2[2], t2[0])) + RT_T, mul(RT_R, vtx(m2[2], t2[1])) + RT_T,
                                  mul(RT_R, vtx(m2[2], t2[2])) + RT_T, P, Q);
    const T d = hsqrt(d2);
    if (mind > d) { mind = d; fb1 = int(p1i); fb2 = int(p2i); np1 = P; np2 = Q; }
  };
  leaf(0, 0);
  struct E { uint32_t b1, b2; T d; };
  std::vector<E> stack;
  stack.push_back({0, 0, T(-1)});
  while (!stack.empty()) {
    const E e = stack.back();
    stack.pop_back();
    if (e.d >= T(0) && e.d >= mind) continue;  // canStop(d) evaluated when the child is about to be visited
    const DNode<T>& n1 = nodes[m1[0] + e.b1];
    const DNode<T>& n2 = nodes[m2[0] + e.b2];
    const bool l1 = n1.first_child < 0, l2 = n2.first_child < 0;
    if (l1 && l2) { leaf(uint32_t(-(n1.first_child + 1)), uint32_t(-(n2.first_child + 1))); continue; }
    uint32_t a1, a2, c1, c2;
    if (l2 || (!l1 && (sqnorm(n1.extent) > sqnorm(n2.extent)))) { a1 = uint32_t(n1.first_child); a2 = e.b2; c1 = a1 + 1; c2 = e.b2; }
    else { a1 = e.b1; a2 = uint32_t(n2.first_child); c1 = e.b1; c2 = a2 + 1; }
    const T d1 = rss_lower_bound(RT_R, RT_T, nodes[m1[0] + a1], rss[m1[0] + a1], nodes[m2[0] + a2], rss[m2[0] + a2]);
    const T d2 = rss_lower_bound(RT_R, RT_T, nodes[m1[0] + c1], rss[m1[0] + c1], nodes[m2[0] + c2], rss[m2[0] + c2]);
    if (d2 < d1) { stack.push_back({a1, a2, d1}); stack.push_back({c1, c2, d2}); }
    else { stack.push_back({c1, c2, d2}); stack.push_back({a1, a2, d1}); }
  }
  const V3<T> w1 = xform(tf1, np1), w2 = xform(tf1, np2);
  r.distance = mind;
  r.normal[0] = r.normal[1] = r.normal[2] = nanv;
  r.p1[0] = w1.x; r.p1[1] = w1.y; r.p1[2] = w1.z;
  r.p2[0] = w2.x; r.p2[1] = w2.y; r.p2[2] = w2.z;
  r.b1 = fb1; r.b2 = fb2;
  r.status = (mind <= T(0)) ? 128u : 0u;
  r.num_contacts = 0;
}

extern "C" {

double sim_rect_distance(const double* Rab, const double* Tab, const double* a, const double* b) {
  M3<double> R;
  R.r0 = mk<double>(Rab[0], Rab[1], Rab[2]);
  R.r1 = mk<double>(Rab[3], Rab[4], Rab[5]);
  R.r2 = mk<double>(Rab[6], Rab[7], Rab[8]);
  return rect_distance(R, mk<double>(Tab[0], Tab[1], Tab[2]), a[0], a[1], b[0], b[1]);
}
double sim_sqr_tri_distance(const double* S, const double* T, double* out) {
  V3<double> P, Q;
  const double d2 = sqr_tri_distance(mk<double>(S[0], S[1], S[2]), mk<double>(S[3], S[4], S[5]), mk<double>(S[6], S[7], S[8]),
                                     mk<double>(T[0], T[1], T[2]), mk<double>(T[3], T[4], T[5]), mk<double>(T[6], T[7], T[8]), P, Q);
  out[0] = P.x; out[1] = P.y; out[2] = P.z; out[3] = Q.x; out[4] = Q.y; out[5] = Q.z;
  return d2;
}

int sim_bvh_distance_f64(const hfcl_bvh_node* nodes, size_t n_nodes, const double* verts, size_t n_verts,
                         const uint32_t* tris, const uint64_t* mesh_table, const uint32_t* m1, const uint32_t* m2,
                         const double* tf1, const double* tf2, size_t n, hfcl_result* out) {
  std::vector<DNode<double>> dn(n_nodes);
  std::vector<DRss<double>> dr(n_nodes);
  for (size_t i = 0; i < n_nodes; ++i) {
    const double* a = nodes[i].obb_axes;
    dn[i].first_child = nodes[i].first_child;
    dn[i].axes.r0 = mk<double>(a[0], a[3], a[6]);
    dn[i].axes.r1 = mk<double>(a[1], a[4], a[7]);
    dn[i].axes.r2 = mk<double>(a[2], a[5], a[8]);
    dn[i].To = mk<double>(nodes[i].obb_To[0], nodes[i].obb_To[1], nodes[i].obb_To[2]);
    dn[i].extent = mk<double>(nodes[i].obb_extent[0], nodes[i].obb_extent[1], nodes[i].obb_extent[2]);
    dr[i].Tr = mk<double>(nodes[i].rss_Tr[0], nodes[i].rss_Tr[1], nodes[i].rss_Tr[2]);
    dr[i].l0 = nodes[i].rss_length[0];
    dr[i].l1 = nodes[i].rss_length[1];
    dr[i].r = nodes[i].rss_radius;
  }
  std::vector<double> v(verts, verts + 3 * n_verts);
  for (size_t i = 0; i < n; ++i)
    bvh_distance_pair<double>(dn, dr, v, tris, mesh_table + 4 * m1[i], mesh_table + 4 * m2[i], pose_from_abi<double>(tf1 + 12 * i),
                              pose_from_abi<double>(tf2 + 12 * i), out[i]);
  return 0;
}

// BVHModel<OBBRSS> x convex shape (either operand order): the device header's mesh_shape_collide on one lane.
// nodes popped / triangles tested by each query of the last lane-form batch (tools/mesh_solid_steps.py: the tail of the walks)
static std::vector<uint32_t> g_steps, g_leaves;
const uint32_t* sim_shape_walk_steps() { return g_steps.data(); }
const uint32_t* sim_shape_walk_leaves() { return g_leaves.data(); }
static int g_shape_lane = 0;  // 1: the one-query-per-lane forms where the request admits them
void sim_set_shape_lane(int on) { g_shape_lane = on; }
struct HostSolid {
  DShape<double> s;
  const double* verts;
  V3<double> operator()(const V3<double>& d) const {
    SerialSupport<double> ss;
    return ss.one(s, verts + 3 * size_t(s.vertex_offset), d);
  }
};
int sim_mesh_shape_collide_f64(const hfcl_shape* shapes, size_t n_shapes, const double* shape_verts, const hfcl_bvh_node* nodes,
                               size_t n_nodes, const double* verts, size_t n_verts, const uint32_t* tris,
                               const uint64_t* mesh_table, const uint32_t* s1, const uint32_t* s2, const double* tf1,
                               const double* tf2, size_t n, const hfcl_collision_request* creq, hfcl_result* out,
                               hfcl_guess* gout, hfcl_contact* contacts, size_t max_contacts, size_t* n_contacts) {
  std::vector<DNode<double>> dn(n_nodes);
  for (size_t i = 0; i < n_nodes; ++i) {
    const double* a = nodes[i].obb_axes;
    dn[i].first_child = nodes[i].first_child;
    dn[i].axes.r0 = mk<double>(a[0], a[3], a[6]);
    dn[i].axes.r1 = mk<double>(a[1], a[4], a[7]);
    dn[i].axes.r2 = mk<double>(a[2], a[5], a[8]);
    dn[i].To = mk<double>(nodes[i].obb_To[0], nodes[i].obb_To[1], nodes[i].obb_To[2]);
    dn[i].extent = mk<double>(nodes[i].obb_extent[0], nodes[i].obb_extent[1], nodes[i].obb_extent[2]);
  }
  (void)n_verts;
  QParams<double> q;
  fill_q(q, creq->q);
  q.mode = 1;
  q.compute_penetration = (creq->enable_contact || creq->security_margin < 0) ? 1 : 0;
  q.security_margin = creq->security_margin;
  const double ub = creq->distance_upper_bound > creq->security_margin ? creq->distance_upper_bound : creq->security_margin;
  q.gjk.distance_upper_bound = ub < 0 ? 0 : ub;
  std::vector<hfcl_contact> cl;
  static thread_local EpaScratch<double, EPA_MAX_ITER> scratch;
  uint32_t stack[128];
  for (size_t i = 0; i < n; ++i) {
    const hfcl_shape &a = shapes[s1[i]], &b = shapes[s2[i]];
    const bool swapped = a.type != HFCL_BV_OBBRSS;
    const hfcl_shape& ms = swapped ? b : a;
    const DShape<double> solid = to_dshape<double>(swapped ? a : b);
    const Pose<double> tfm = pose_from_abi<double>((swapped ? tf2 : tf1) + 12 * i);
    const Pose<double> tfs = pose_from_abi<double>((swapped ? tf1 : tf2) + 12 * i);
    const uint64_t* mt = mesh_table + 4 * ms.bvh_index;
    HostSolid hs{solid, shape_verts};
    MeshShapeState<double> st;
    auto on_contact = [&](int prim, double distance, const V3<double>& p1, const V3<double>& p2, const V3<double>& nn) {
      if (!contacts) return;
      hfcl_contact c;
      c.pair = uint32_t(i);
      c.b1 = swapped ? -1 : prim;
      c.b2 = swapped ? prim : -1;
      c._pad = 0;
      c.penetration_depth = distance;
      const V3<double> a1 = swapped ? p2 : p1, a2 = swapped ? p1 : p2, an = swapped ? -nn : nn;
      c.normal[0] = an.x; c.normal[1] = an.y; c.normal[2] = an.z;
      c.p1[0] = a1.x; c.p1[1] = a1.y; c.p1[2] = a1.z;
      c.p2[0] = a2.x; c.p2[1] = a2.y; c.p2[2] = a2.z;
      cl.push_back(c);
    };
    if (g_shape_lane && mesh_shape_lane_request(q, creq->num_max_contacts)) {
      // the one-query-per-lane form (k_bvh_shape_lane + k_bvh_shape_finish), in the order the kernels take the steps
      const double nanv = Lim<double>::nan();
      const DNode<double>* nodes1 = dn.data() + mt[0];
      const double* mverts = verts + 3 * mt[2];
      const uint32_t* mtris = tris + 3 * mt[3];
      st.guess = mk<double>(q.guess[0], q.guess[1], q.guess[2]);
      st.dlb = st.rec_dist = Lim<double>::max();
      st.np1 = st.np2 = st.nn = mk<double>(nanv, nanv, nanv);
      st.ncontacts = 0;
      st.first_prim = -1;
      st.overflow = st.unsupported = false;
      DNode<double> bv2;
      if (!shape_obb(solid, shape_verts, tfs, bv2)) {
        st.unsupported = true;
      } else {
        const ObbQuery<double> oq = make_obb_query(tfm, bv2);
        const MDiff<double> sMt = make_mdiff(tfs, tfm);
        const double r1 = swept_radius(solid), bd2 = creq->break_distance * creq->break_distance;
        int sp = 1;
        stack[0] = 0;
        if (g_steps.size() < n) { g_steps.assign(n, 0); g_leaves.assign(n, 0); }
        g_steps[i] = g_leaves[i] = 0;
        while (sp > 0) {
          const DNode<double> n1 = nodes1[stack[--sp]];
          ++g_steps[i];
          if (n1.first_child < 0) ++g_leaves[i];
          if (n1.first_child >= 0) {
            double sq;
            if (obb_disjoint_q(oq, n1, q.security_margin, bd2, sq)) {
              mesh_shape_bv_bound(sq, q, st.dlb, st.rec_dist);
            } else if (sp + 2 > 128) {
              st.overflow = true;
              sp = 0;
            } else {
              stack[sp] = uint32_t(n1.first_child + 1);
              stack[sp + 1] = uint32_t(n1.first_child);
              sp += 2;
            }
            continue;
          }
          const uint32_t prim = uint32_t(-(n1.first_child + 1));
          const uint32_t* t3 = mtris + 3 * size_t(prim);
          auto vtx = [&](uint32_t k) { return mk<double>(mverts[3 * size_t(k)], mverts[3 * size_t(k) + 1], mverts[3 * size_t(k) + 2]); };
          auto tfm_of = [&]() { return tfm; };
          auto tfs_of = [&]() { return tfs; };
          double distance;
          V3<double> p1, p2, nn;
          ShapeDeferItem<double> item;
          if (mesh_shape_leaf_lane(vtx(t3[0]), vtx(t3[1]), vtx(t3[2]), sMt, tfm_of, tfs_of, solid, hs, r1, q, st.guess, W0Regs<double>(),
                                   distance, p1, p2, nn, item)) {
            item.seed.pair = uint32_t(i);
            item.bound = st.dlb;
            item.rec = st.rec_dist;
            item.prim = prim;
            distance = mesh_shape_leaf_finish<double, SerialGroup<1>>(item, tfs, hs, r1, q, &scratch, p1, p2, nn, st.guess);
            st.dlb = item.bound;
            st.rec_dist = item.rec;
            sp = 0;  // the walk ended at this leaf
            bool lowered;
            const bool contact = mesh_shape_leaf_bound(distance, q, st.dlb, st.rec_dist, lowered);
            if (lowered) { st.np1 = p1; st.np2 = p2; st.nn = nn; }
            if (contact) {
              st.ncontacts = 1;
              st.first_prim = int(prim);
              on_contact(int(prim), distance, p1, p2, nn);
            } else {
              st.overflow = true;
            }
          } else {
            bool lowered;
            const bool contact = mesh_shape_leaf_bound(distance, q, st.dlb, st.rec_dist, lowered);
            if (lowered) { st.np1 = p1; st.np2 = p2; st.nn = nn; }
            if (contact) {
              st.ncontacts = 1;
              st.first_prim = int(prim);
              on_contact(int(prim), distance, p1, p2, nn);
              sp = 0;
            }
          }
        }
      }
    } else
    mesh_shape_collide<double, SerialGroup<1>>(dn.data() + mt[0], verts + 3 * mt[2], tris + 3 * mt[3], tfm, solid, shape_verts, tfs, hs,
                                              q, creq->num_max_contacts, creq->break_distance * creq->break_distance, stack, 128,
                                              &scratch, mk<double>(q.guess[0], q.guess[1], q.guess[2]), on_contact, st);
    hfcl_result& r = out[i];
    std::memset(&r, 0, sizeof(r));
    if (st.unsupported) {
      r.status = 0x80000000u;
      continue;
    }
    const V3<double> p1 = swapped ? st.np2 : st.np1, p2 = swapped ? st.np1 : st.np2, nn = swapped ? -st.nn : st.nn;
    r.distance = st.rec_dist;
    r.normal[0] = nn.x; r.normal[1] = nn.y; r.normal[2] = nn.z;
    r.p1[0] = p1.x; r.p1[1] = p1.y; r.p1[2] = p1.z;
    r.p2[0] = p2.x; r.p2[1] = p2.y; r.p2[2] = p2.z;
    r.b1 = swapped ? -1 : st.first_prim;
    r.b2 = swapped ? st.first_prim : -1;
    r.status = (st.ncontacts ? 128u : 0u) | (st.overflow ? 0xC0000000u : 0u);
    r.num_contacts = int(st.ncontacts);
    if (gout) {
      gout[i].gjk_guess[0] = st.guess.x; gout[i].gjk_guess[1] = st.guess.y; gout[i].gjk_guess[2] = st.guess.z;
      gout[i].support_guess[0] = gout[i].support_guess[1] = 0;
    }
  }
  (void)n_shapes;
  if (contacts) {
    size_t k = 0;
    for (auto& c : cl)
      if (k < max_contacts) contacts[k++] = c;
    *n_contacts = cl.size();
  }
  return 0;
}

int sim_mesh_shape_distance_f64(const hfcl_shape* shapes, size_t n_shapes, const double* shape_verts, const hfcl_bvh_node* nodes,
                                size_t n_nodes, const double* verts, size_t n_verts, const uint32_t* tris,
                                const uint64_t* mesh_table, const uint32_t* s1, const uint32_t* s2, const double* tf1,
                                const double* tf2, size_t n, const hfcl_distance_request* dreq, hfcl_result* out,
                                hfcl_guess* gout) {
  std::vector<DNode<double>> dn(n_nodes);
  std::vector<DRss<double>> dr(n_nodes);
  for (size_t i = 0; i < n_nodes; ++i) {
    const double* a = nodes[i].obb_axes;
    dn[i].first_child = nodes[i].first_child;
    dn[i].axes.r0 = mk<double>(a[0], a[3], a[6]);
    dn[i].axes.r1 = mk<double>(a[1], a[4], a[7]);
    dn[i].axes.r2 = mk<double>(a[2], a[5], a[8]);
    dn[i].To = mk<double>(nodes[i].obb_To[0], nodes[i].obb_To[1], nodes[i].obb_To[2]);
    dn[i].extent = mk<double>(nodes[i].obb_extent[0], nodes[i].obb_extent[1], nodes[i].obb_extent[2]);
    dr[i].Tr = mk<double>(nodes[i].rss_Tr[0], nodes[i].rss_Tr[1], nodes[i].rss_Tr[2]);
    dr[i].l0 = nodes[i].rss_length[0];
    dr[i].l1 = nodes[i].rss_length[1];
    dr[i].r = nodes[i].rss_radius;
  }
  (void)n_verts;
  (void)n_shapes;
  QParams<double> q;
  fill_q(q, dreq->q);
  q.mode = 0;
  q.compute_penetration = dreq->enable_signed_distance ? 1 : 0;
  q.security_margin = 0;
  q.gjk.distance_upper_bound = Lim<double>::max();
  static thread_local EpaScratch<double, EPA_MAX_ITER> scratch;
  uint32_t stack_n[128];
  double stack_d[128];
  for (size_t i = 0; i < n; ++i) {
    const hfcl_shape &a = shapes[s1[i]], &b = shapes[s2[i]];
    const bool swapped = a.type != HFCL_BV_OBBRSS;
    const hfcl_shape& ms = swapped ? b : a;
    const DShape<double> solid = to_dshape<double>(swapped ? a : b);
    const Pose<double> tfm = pose_from_abi<double>((swapped ? tf2 : tf1) + 12 * i);
    const Pose<double> tfs = pose_from_abi<double>((swapped ? tf1 : tf2) + 12 * i);
    const uint64_t* mt = mesh_table + 4 * ms.bvh_index;
    HostSolid hs{solid, shape_verts};
    MeshShapeDist<double> st;
    mesh_shape_distance<double, SerialGroup<1>>(dn.data() + mt[0], dr.data() + mt[0], verts + 3 * mt[2], tris + 3 * mt[3], tfm, solid,
                                               shape_verts, tfs, hs, q, stack_n, stack_d, 128, &scratch,
                                               mk<double>(q.guess[0], q.guess[1], q.guess[2]), st);
    hfcl_result& r = out[i];
    std::memset(&r, 0, sizeof(r));
    if (st.unsupported) {
      r.status = 0x80000000u;
      continue;
    }
    const V3<double> p1 = swapped ? st.np2 : st.np1, p2 = swapped ? st.np1 : st.np2, nn = swapped ? -st.nn : st.nn;
    r.distance = st.min_distance;
    r.normal[0] = nn.x; r.normal[1] = nn.y; r.normal[2] = nn.z;
    r.p1[0] = p1.x; r.p1[1] = p1.y; r.p1[2] = p1.z;
    r.p2[0] = p2.x; r.p2[1] = p2.y; r.p2[2] = p2.z;
    r.b1 = st.prim;  // distance.cpp:84-88 does not swap b1 / b2
    r.b2 = -1;
    r.status = (st.min_distance <= 0 ? 128u : 0u) | (st.overflow ? 0xC0000000u : 0u);
    r.num_contacts = 0;
    if (gout) {
      gout[i].gjk_guess[0] = st.guess.x; gout[i].gjk_guess[1] = st.guess.y; gout[i].gjk_guess[2] = st.guess.z;
      gout[i].support_guess[0] = gout[i].support_guess[1] = 0;
    }
  }
  return 0;
}

// library-wide ranks of the node sizes (extent.squaredNorm(), fp64, ties share a rank) -> filter records; what
// hfcl_host.hip: upload_bvh builds (the same function is used there)
static std::vector<DNodeF> make_fnodes(const hfcl_bvh_node* nodes, size_t n_nodes) {
  std::vector<uint32_t> rank(n_nodes);
  obbf_size_ranks(nodes, n_nodes, rank.data());
  std::vector<DNodeF> f(n_nodes);
  for (size_t i = 0; i < n_nodes; ++i) f[i] = pack_fnode(nodes[i], rank[i]);
  return f;
}

static int sim_bvh_collide_impl(const hfcl_bvh_node* nodes, size_t n_nodes, const double* verts, size_t n_verts,
                        const uint32_t* tris, const uint64_t* mesh_table, const uint32_t* m1, const uint32_t* m2,
                        const double* tf1, const double* tf2, size_t n, const hfcl_collision_request* creq, hfcl_result* out,
                        hfcl_contact* contacts, size_t max_contacts, size_t* n_contacts, uint64_t* fstats);

int sim_bvh_collide_f64(const hfcl_bvh_node* nodes, size_t n_nodes, const double* verts, size_t n_verts,
                        const uint32_t* tris, const uint64_t* mesh_table, const uint32_t* m1, const uint32_t* m2,
                        const double* tf1, const double* tf2, size_t n, const hfcl_collision_request* creq, hfcl_result* out,
                        hfcl_contact* contacts, size_t max_contacts, size_t* n_contacts) {
  return sim_bvh_collide_impl(nodes, n_nodes, verts, n_verts, tris, mesh_table, m1, m2, tf1, tf2, n, creq, out, contacts,
                              max_contacts, n_contacts, nullptr);
}
// the same walk with the fp32 filter in front of the fp64 separating-axis test; fstats: 7 counters (bvh_pair)
int sim_bvh_collide_filtered_f64(const hfcl_bvh_node* nodes, size_t n_nodes, const double* verts, size_t n_verts,
                        const uint32_t* tris, const uint64_t* mesh_table, const uint32_t* m1, const uint32_t* m2,
                        const double* tf1, const double* tf2, size_t n, const hfcl_collision_request* creq, hfcl_result* out,
                        uint64_t* fstats) {
  for (int k = 0; k < 7; ++k) fstats[k] = 0;
  return sim_bvh_collide_impl(nodes, n_nodes, verts, n_verts, tris, mesh_table, m1, m2, tf1, tf2, n, creq, out, nullptr, 0,
                              nullptr, fstats);
}

static int sim_bvh_collide_impl(const hfcl_bvh_node* nodes, size_t n_nodes, const double* verts, size_t n_verts,
                        const uint32_t* tris, const uint64_t* mesh_table, const uint32_t* m1, const uint32_t* m2,
                        const double* tf1, const double* tf2, size_t n, const hfcl_collision_request* creq, hfcl_result* out,
                        hfcl_contact* contacts, size_t max_contacts, size_t* n_contacts, uint64_t* fstats) {
  std::vector<DNodeF> fn;
  if (fstats) fn = make_fnodes(nodes, n_nodes);
  std::vector<DNode<double>> dn(n_nodes);
  for (size_t i = 0; i < n_nodes; ++i) {
    const double* a = nodes[i].obb_axes;
    dn[i].first_child = nodes[i].first_child;
    dn[i].axes.r0 = mk<double>(a[0], a[3], a[6]);
    dn[i].axes.r1 = mk<double>(a[1], a[4], a[7]);
    dn[i].axes.r2 = mk<double>(a[2], a[5], a[8]);
    dn[i].To = mk<double>(nodes[i].obb_To[0], nodes[i].obb_To[1], nodes[i].obb_To[2]);
    dn[i].extent = mk<double>(nodes[i].obb_extent[0], nodes[i].obb_extent[1], nodes[i].obb_extent[2]);
  }
  std::vector<double> v(verts, verts + 3 * n_verts);
  QParams<double> q;
  fill_q(q, creq->q);
  q.mode = 1;
  q.compute_penetration = (creq->enable_contact || creq->security_margin < 0) ? 1 : 0;
  q.security_margin = creq->security_margin;
  q.gjk.distance_upper_bound = Lim<double>::max();
  std::vector<hfcl_contact> cl;
  for (size_t i = 0; i < n; ++i)
    bvh_pair<double>(dn, v, tris, mesh_table + 4 * m1[i], mesh_table + 4 * m2[i], pose_from_abi<double>(tf1 + 12 * i),
                     pose_from_abi<double>(tf2 + 12 * i), q, creq->num_max_contacts,
                     creq->break_distance * creq->break_distance, out[i], contacts ? &cl : nullptr, uint32_t(i),
                     fstats ? fn.data() : nullptr, fstats);
  if (contacts) {
    size_t k = 0;
    for (auto& c : cl)
      if (k < max_contacts) contacts[k++] = c;
    *n_contacts = cl.size();
  }
  return 0;
}

}  // extern "C"
