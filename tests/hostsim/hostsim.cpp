// tests/hostsim -- CPU-side validation build of the *device* per-pair code.
//
// TEST INFRASTRUCTURE ONLY.  This compiles the very headers the HIP kernels are made of
// (hpp-fcl_amd/csrc/hfcl_{math,gjk,epa,shapes,pair}.hpp) with plain g++ and runs them one pair
// at a time with a serial support evaluator, so the GJK/EPA core that ships in the kernels can
// be checked against the fp64 oracle in this GPU-less container (fp64 and fp32 instantiations).
// It is NOT a fallback: the product library (csrc/libhppfcl_amd.so) neither links nor loads
// this file, and the C ABI fails with HFCL_ERR_NO_DEVICE when there is no GPU.
#include <cstring>
#include <vector>

#include "../../include/hppfcl_amd.h"
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
  d.ssr = T(s.swept_sphere_radius);
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
static void one_pair(const DShape<T>& a, const DShape<T>& b, const T* verts, const Pose<T>& tf1, const Pose<T>& tf2,
                     const QParams<T>& q, const V3<T>& guess0, PairOut<T>& o, bool& contact, int& nc, bool& skipped) {
  skipped = false;
  const int cls = pair_class(a.kind, b.kind);
  if (cls == CLS_CLOSED) {
    o.distance = closed_form_distance(a, tf1, b, tf2, o.p1, o.p2, o.normal);
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
    gjk_run(g, q.gjk, guess0, r0 + r1, a.kind == K_CONVEX && b.kind == K_CONVEX, sup);
    EpaSeed<T> seed;
    if (gjk_finish(g, q, tf1, r0, r1, guess0, o, seed)) {
      static thread_local EpaScratch<T> scratch;
      epa_run<T, SerialGroup<1>>(&scratch, seed, q, tf1, r0, r1, sup, o);
    }
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
  for (size_t i = 0; i < n_shapes; ++i) lib[i] = to_dshape<double>(shapes[i]);
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

}  // extern "C"
