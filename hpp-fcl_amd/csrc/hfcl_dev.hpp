// hfcl_dev.hpp -- what the kernel translation units (hfcl_k_gjk.hip, hfcl_k_epa.hip, hfcl_k_bvh.hip) and the host side
// (hfcl_host.hip) share: bucket ids, kernel parameter blocks, result-record writers, lane-group primitives, the
// register-resident hull, tier constants of the EPA kernels and the BVH views.  gfx950 / wave64 only.
#pragma once
#include <hip/hip_runtime.h>

#include <cstdint>
#include <cstring>
#include <type_traits>

#include "../../include/hppfcl_amd.h"
#include "hfcl_bvh.hpp"
#include "hfcl_bvh_shape.hpp"
#include "hfcl_pair.hpp"

using namespace hfcl;

// minimum waves per SIMD the register allocator must allow for (A/B-tuned, see profiles/)
#ifndef HFCL_WPE_GJK_W2
#define HFCL_WPE_GJK_W2 2  // 2-lane groups hold 16 vertices of each hull per lane (96 VGPRs)
#endif
#ifndef HFCL_WPE_GJK
#define HFCL_WPE_GJK 3
#endif
// k_epa's occupancy attribute follows from its LDS block (hfcl_k_epa.hip: epa_waves_per_simd): the fp64 tiers are sized
// for two waves per SIMD.  (Round 1 found a forced two-wave fp64 build to miscompute -- profiles/r01_c_waves_per_eu_ab.txt;
// that was the hand-over save reading unwritten vertex records, fixed in round 2: profiles/r02_u.)
#ifndef HFCL_WPE_GJK64
#define HFCL_WPE_GJK64 2
#endif
#ifndef HFCL_WPE_PRIM
#define HFCL_WPE_PRIM 2
#endif
#ifndef HFCL_WPE_EPA32
#define HFCL_WPE_EPA32 2  // fp32 EPA: the LDS block allows 2 waves/SIMD, keep the registers within that
#endif
#ifndef HFCL_WPE_EPA64
#define HFCL_WPE_EPA64 1  // (only where an fp64 EPA block is too large for two waves per SIMD: A/B builds)
#endif
#ifndef HFCL_WPE_BVH
#define HFCL_WPE_BVH 1
#endif

// ---------------------------------------------------------------------------------------
// bucket ids (finer than hfcl_shapes.hpp's pair_class: the convex bucket is split by which
// side carries vertices so the kernel is specialised at compile time)
// ---------------------------------------------------------------------------------------
enum { B_CLOSED = 0, B_PRIM = 1, B_CC = 2, B_PC = 3, B_CP = 4, B_BVH = 5, B_UNSUPPORTED = 6, B_LARGE = 7, B_BVHSHAPE = 8, B_TRI = 9, B_COUNT = 10 };
// Pairs with a curved shape (Ellipsoid, Cone, Cylinder) take two to three times the GJK iterations of the others (their
// supports are not vertices: cfg5 with the oracle, 10 against 4.5 on average, p99 23 against 8), and a GJK kernel steps
// the 32 / 64 pairs of a wave in lockstep.  k_classify therefore files them from the top end of their bucket's list
// (population in counts[B_CURVED0 + bucket]); a kernel walks the list's bottom part, then its top part (BucketList), and
// the pairs of a wave are of one class but for one wave per bucket.
constexpr int B_CURVED0 = B_COUNT + 4;
// + the two counters of the one-query-per-lane mesh x solid form (its ticket, the length of its EPA queue)
constexpr int CTR_SHAPE_TICKET = 2 * B_COUNT + 4, CTR_SHAPE_DEFER = 2 * B_COUNT + 5;
constexpr int CTR_DIST_SUSP = 2 * B_COUNT + 6;  // suspended mesh x mesh distance() walks (DistSusp records)
constexpr int CTR_SHAPE_DIST_SUSP = 2 * B_COUNT + 7;  // ... mesh x solid (ShapeDistSusp records)
constexpr int CTR_DIST_TICKET = 2 * B_COUNT + 8;  // ticket of k_bvh_distance_pool: next DistSusp record to take
constexpr int CTR_SHAPE_DIST_TICKET = 2 * B_COUNT + 9;  // ticket of k_bvh_shape_distance_pool
constexpr int CTR_EPA_CC_OVER = 2 * B_COUNT + 10;  // convex x convex polytopes k_epa_loop saved for k_epa_resume_cc
constexpr int CTR_SHAPE_FINISH_OVER = 2 * B_COUNT + 11;  // [+0, +1] mesh x solid EPA leaves that outgrew k_bvh_shape_finish's fast block (the two halves of Work::shape_finish_over)
constexpr int CTR_SHAPE_DEFER_MARK = 2 * B_COUNT + 13;   // CTR_SHAPE_DEFER as the first launch of k_bvh_shape_coop left it: the items of whole walks (k_bvh_level_mark)
// walks the pooled distance() continuations walked again in the reference's order (BvhSpill::rerun_count): mesh x mesh, mesh x solid
constexpr int CTR_DIST_RERUN = 2 * B_COUNT + 14, CTR_SHAPE_DIST_RERUN = 2 * B_COUNT + 15;
constexpr int N_COUNTERS = 2 * B_COUNT + 16;  // bucket populations + the four counters of Work::counts + curved populations + those

// Classification-only kind code of a ConvexBase with more than 32 vertices (the reference switches
// support algorithm there, minkowski_difference.cpp:136-151): GJK pairs with such a hull go to
// the B_LARGE bucket whose kernel scans the vertices from memory instead of holding them in registers.
constexpr int K_CONVEX_LARGE = 21;

// distance_mode: the reference's *distance* function matrix has no TriangleP row or column at all
// (src/distance_func_matrix.cpp:283-560; only collide() knows GEOM_TRIANGLE, collision_func_matrix.cpp:295-469):
// distance() on such a pair throws there and is reported as unsupported here.
__host__ __device__ inline int bucket_of(int k1, int k2, bool distance_mode = false) {
  if (distance_mode && (k1 == K_TRIANGLE || k2 == K_TRIANGLE)) return B_UNSUPPORTED;
  const bool large = (k1 == K_CONVEX_LARGE) || (k2 == K_CONVEX_LARGE);
  if (k1 == K_CONVEX_LARGE) k1 = K_CONVEX;
  if (k2 == K_CONVEX_LARGE) k2 = K_CONVEX;
  if ((k1 == K_BVH) != (k2 == K_BVH)) {  // BVHModel x convex solid, either operand order (k_bvh_shape)
    const int o = (k1 == K_BVH) ? k2 : k1;
    return (kind_is_prim(o) || o == K_CONVEX || kind_is_flat(o)) ? B_BVHSHAPE : B_UNSUPPORTED;
  }
  if ((k1 == K_TRIANGLE || k2 == K_TRIANGLE) && !kind_is_flat(k1) && !kind_is_flat(k2)) {
    // top-level TriangleP rows of the table (collision_func_matrix.cpp:295-469): k_triangle
    const int o = (k1 == K_TRIANGLE) ? k2 : k1;
    return (o == K_TRIANGLE || kind_is_prim(o) || o == K_CONVEX) ? B_TRI : B_UNSUPPORTED;
  }
  const int c = pair_class(k1, k2);
  if (large && c == CLS_CONVEX) return B_LARGE;
  if (c == CLS_CLOSED) return B_CLOSED;
  if (c == CLS_PRIM_GJK) return B_PRIM;
  if (c == CLS_BVH) return B_BVH;
  if (c == CLS_CONVEX) {
    if (k1 == K_CONVEX && k2 == K_CONVEX) return B_CC;
    return (k1 == K_CONVEX) ? B_CP : B_PC;
  }
  return B_UNSUPPORTED;
}

// ---------------------------------------------------------------------------------------
// kernel parameter blocks
// ---------------------------------------------------------------------------------------
template <typename T>
struct NbrEntry;
// one neighbour of a hull vertex: its coordinates (in the precision of the vertex table, same rounding) and its index
template <> struct alignas(16) NbrEntry<float> { float x, y, z; uint32_t id; };
template <> struct alignas(32) NbrEntry<double> { double x, y, z; uint32_t id; uint32_t pad_; };

template <typename T>
struct LibView {
  const DShape<T>* shapes;
  const T* verts;
  const uint8_t* kinds;
  uint32_t n_shapes;
  // vertex adjacency of the convex shapes that registered one (hfcl_lib_set_convex_neighbors): graph_base[shape] is the
  // index in graph_off of the shape's num_points+1 offsets into graph_ent (HFCL_NO_GRAPH: none; the 14 warm-start
  // vertices of the shape sit just before its offsets); hulls of >= climb_min vertices with a graph hill-climb
  const uint32_t* graph_base;
  const uint32_t* graph_off;
  const NbrEntry<T>* graph_ent;
  uint32_t climb_min;
};
static constexpr uint32_t HFCL_NO_GRAPH = 0xFFFFFFFFu;
#ifndef HFCL_CLIMB_MIN
#define HFCL_CLIMB_MIN 512u  // climb_support overtakes scan_support between 256 and 1024 vertices (profiles/r02_p)
#endif
static constexpr int HFCL_WARM_STARTS = 14;  // ConvexBase::num_support_warm_starts (shape/geometric_shapes.h)
template <typename T>
struct HullGraph {
  const uint32_t* off;  // nullptr: no graph, scan the vertices
  const NbrEntry<T>* ent;
};
template <typename T>
__device__ __forceinline__ HullGraph<T> hull_graph(const LibView<T>& lib, uint32_t shape_id, uint32_t num_points) {
  HullGraph<T> g{nullptr, nullptr};
  if (lib.graph_base == nullptr || num_points < lib.climb_min) return g;
  const uint32_t base = lib.graph_base[shape_id];
  if (base == HFCL_NO_GRAPH) return g;
  g.off = lib.graph_off + base;
  g.ent = lib.graph_ent;
  return g;
}

template <typename T> struct IO;
template <> struct IO<double> {
  const double* tf1;
  const double* tf2;
  hfcl_result* out;
  const hfcl_guess* gin;
  hfcl_guess* gout;
};
template <> struct IO<float> {
  const float* tf1;
  const float* tf2;
  hfcl_result_f32* out;
  const hfcl_guess* gin;  // unused
  hfcl_guess* gout;       // unused
};

template <typename T> using EpaItem = EpaSeed<T>;

struct Work {
  const uint32_t* shape1;
  const uint32_t* shape2;
  uint32_t n;
  uint32_t* lists;   // B_COUNT lists of capacity n each
  uint32_t* counts;  // B_COUNT counters + [B_COUNT] = epa queue length + [B_COUNT+1] = overflow queue length
                     // + [B_COUNT+2] = ticket counter of the streaming BVH kernel + [B_COUNT+3] = length of the
                     // second EPA queue (the top end of epa_queue, filled downwards from slot n-1: fp32 convex x convex,
                     // fp64 pairs with a curved shape) + [B_CURVED0 + b] = curved pairs of bucket b (top end of its list)
  void* epa_queue;
  void* epa_queue2;  // polytopes that outgrew the fast EPA kernel's scratch block
  void* epa_v0;      // shape-0 support points of the polytopes in flight in the full-capacity EPA kernel
  void* epa_resume;  // saved polytopes (EpaSaved, epa_resume_stride<T> bytes apart) of the first `resume_cap` slots of epa_queue2
  uint32_t resume_cap;
  void* shape_defer;  // ShapeDeferItem<T>[shape_defer_cap]: mesh x solid leaves waiting for EPA (k_bvh_collide<SOLID> -> k_bvh_shape_finish); nullptr: group kernels
  uint32_t shape_defer_cap;  // a unit (query, or task of a split walk) queues at most one item: sized by the host for every unit a batch can make
  uint32_t* shape_finish_over;  // [2 * shape_defer_cap] indices of shape_defer: the items whose polytope outgrew the fast block of k_bvh_shape_finish (nullptr: one tier)
  void* shape_oq;     // ObbQuery<T>[n], by pair: the solid's fitted OBB against the mesh pose (k_shape_obb)
  void* epa_ready;    // EpaReady<T>[n]: convex x convex polytopes between k_epa_prepare, k_epa_loop and k_epa_records (nullptr: the one-kernel form)
  void* epa_ready_g;  // EpaReadyG<T>[n]: polytopes of any pair kinds between k_epa_prepare_general / k_epa_loop_general / k_epa_records_general (nullptr: the lockstep tiers)
  uint32_t* epa_cc_over;  // blocks of epa_ready whose polytope k_epa_loop saved for k_epa_resume_cc, slot i <-> slot cc_resume_base + i of epa_resume
  uint32_t cc_resume_base, cc_resume_cap;
};
// a pair with a shape whose support is not a vertex
__host__ __device__ inline bool curved_pair(int k1, int k2) {
  return k1 == K_ELLIPSOID || k1 == K_CONE || k1 == K_CYLINDER || k2 == K_ELLIPSOID || k2 == K_CONE || k2 == K_CYLINDER;
}
// bucket `b` of a batch in processing order: the bottom part of its list, then the curved pairs from the top end
struct BucketList {
  const uint32_t* base;
  uint32_t n, c0, cnt;
  __device__ __forceinline__ uint32_t operator[](uint32_t i) const { return i < c0 ? base[i] : base[n - 1u - (i - c0)]; }
};
__device__ __forceinline__ BucketList bucket_list(const Work& wk, int b) {
  const uint32_t c0 = wk.counts[b];
  return BucketList{wk.lists + size_t(b) * wk.n, wk.n, c0, c0 + wk.counts[B_CURVED0 + b]};
}
constexpr int32_t EPA_RESUME_FLAG = 0x100;   // EpaSeed::rank bit: "continue the saved polytope of this slot"
constexpr int32_t EPA_RESUME_SMALL = 0x200;  // ... "saved by the fast tier of the polytope class" (EpaSaved<T, epa_small_cap<T>>)

__device__ __forceinline__ Pose<double> load_pose(const double* base, uint32_t i) { return pose_from_abi<double>(base + 12 * size_t(i)); }
__device__ __forceinline__ void put_record(hfcl_result* dst, const hfcl_result& r) { *dst = r; }
__device__ __forceinline__ Pose<float> load_pose(const float* base, uint32_t i) { return pose_from_quat<float>(base + 7 * size_t(i)); }

// One finished query -> result record (tail of ShapeShapeDistancer::run / ShapeShapeCollider::run).
__device__ __forceinline__ void store_record(const IO<double>& io, uint32_t pair, const PairOut<double>& o, bool contact,
                                             int nc) {
  hfcl_result r;
  r.distance = o.distance;
  r.normal[0] = o.normal.x; r.normal[1] = o.normal.y; r.normal[2] = o.normal.z;
  r.p1[0] = o.p1.x; r.p1[1] = o.p1.y; r.p1[2] = o.p1.z;
  r.p2[0] = o.p2.x; r.p2[1] = o.p2.y; r.p2[2] = o.p2.z;
  r.b1 = -1;
  r.b2 = -1;
  r.status = pack_status(o.gjk_status, o.epa_status, contact, o.gjk_iters, o.epa_iters);
  r.num_contacts = nc;
  put_record(&io.out[pair], r);
}
__device__ __forceinline__ void store_record(const IO<float>& io, uint32_t pair, const PairOut<float>& o, bool contact,
                                             int) {
  hfcl_result_f32 r;
  r.distance = o.distance;
  r.p1[0] = o.p1.x; r.p1[1] = o.p1.y; r.p1[2] = o.p1.z;
  r.p2[0] = o.p2.x; r.p2[1] = o.p2.y; r.p2[2] = o.p2.z;
  r.normal[0] = o.normal.x; r.normal[1] = o.normal.y; r.normal[2] = o.normal.z;
  r.status = pack_status(o.gjk_status, o.epa_status, contact, o.gjk_iters, o.epa_iters);
  io.out[pair] = r;
}
template <typename T>
__device__ __forceinline__ void write_out(const IO<T>& io, const QParams<T>& q, uint32_t pair, PairOut<T> o) {
  int nc;
  const bool contact = apply_query_semantics(q, o, nc);
  store_record(io, pair, o, contact, nc);
}

// BVH pair record: distance = distance_lower_bound + margin (= the first contact's penetration depth
// when num_max_contacts == 1), p1/p2/normal = CollisionResult::nearest_points/normal, b1/b2 = first contact.
__device__ __forceinline__ void store_bvh_record(const IO<double>& io, uint32_t pair, const PairOut<double>& o, uint32_t nc,
                                                 int b1, int b2, bool overflow) {
  hfcl_result r;
  r.distance = o.distance;
  r.normal[0] = o.normal.x; r.normal[1] = o.normal.y; r.normal[2] = o.normal.z;
  r.p1[0] = o.p1.x; r.p1[1] = o.p1.y; r.p1[2] = o.p1.z;
  r.p2[0] = o.p2.x; r.p2[1] = o.p2.y; r.p2[2] = o.p2.z;
  r.b1 = b1;
  r.b2 = b2;
  // nc: number of contacts (collide) ; bit 31 set = "distance(): contact flag only, no Contact object"
  r.status = (nc ? 128u : 0u) | (overflow ? 0xC0000000u : 0u);
  r.num_contacts = int(nc & 0x7FFFFFFFu);
  io.out[pair] = r;
}
__device__ __forceinline__ void store_bvh_record(const IO<float>& io, uint32_t pair, const PairOut<float>& o, uint32_t nc,
                                                 int, int, bool overflow) {
  hfcl_result_f32 r;
  r.distance = o.distance;
  r.p1[0] = o.p1.x; r.p1[1] = o.p1.y; r.p1[2] = o.p1.z;
  r.p2[0] = o.p2.x; r.p2[1] = o.p2.y; r.p2[2] = o.p2.z;
  r.normal[0] = o.normal.x; r.normal[1] = o.normal.y; r.normal[2] = o.normal.z;
  r.status = (nc ? 128u : 0u) | (overflow ? 0xC0000000u : 0u);
  io.out[pair] = r;
}

// The witness part of a BVH pair record (p1, p2, normal) and the rest of it, written separately: the mesh x mesh
// traversal updates the witness in place whenever a leaf lowers its bound (a handful of times per query) instead of
// carrying 9 values in registers for the whole walk (k_bvh_collide runs on its register budget: two waves per SIMD).
__device__ __forceinline__ void store_witness(const IO<double>& io, uint32_t pair, const V3<double>& p1, const V3<double>& p2, const V3<double>& n) {
  hfcl_result* r = &io.out[pair];
  r->normal[0] = n.x; r->normal[1] = n.y; r->normal[2] = n.z;
  r->p1[0] = p1.x; r->p1[1] = p1.y; r->p1[2] = p1.z;
  r->p2[0] = p2.x; r->p2[1] = p2.y; r->p2[2] = p2.z;
}
__device__ __forceinline__ void store_witness(const IO<float>& io, uint32_t pair, const V3<float>& p1, const V3<float>& p2, const V3<float>& n) {
  hfcl_result_f32* r = &io.out[pair];
  r->p1[0] = p1.x; r->p1[1] = p1.y; r->p1[2] = p1.z;
  r->p2[0] = p2.x; r->p2[1] = p2.y; r->p2[2] = p2.z;
  r->normal[0] = n.x; r->normal[1] = n.y; r->normal[2] = n.z;
}
template <typename T>
__device__ __forceinline__ void load_witness(const IO<T>& io, uint32_t pair, V3<T>& p1, V3<T>& p2, V3<T>& n) {
  const auto* r = &io.out[pair];
  n = mk<T>(T(r->normal[0]), T(r->normal[1]), T(r->normal[2]));
  p1 = mk<T>(T(r->p1[0]), T(r->p1[1]), T(r->p1[2]));
  p2 = mk<T>(T(r->p2[0]), T(r->p2[1]), T(r->p2[2]));
}
__device__ __forceinline__ void store_bvh_record_head(const IO<double>& io, uint32_t pair, double distance, uint32_t nc, int b1, int b2, bool overflow) {
  hfcl_result* r = &io.out[pair];
  r->distance = distance;
  r->b1 = b1;
  r->b2 = b2;
  r->status = (nc ? 128u : 0u) | (overflow ? 0xC0000000u : 0u);
  r->num_contacts = int(nc & 0x7FFFFFFFu);
}
__device__ __forceinline__ void store_bvh_record_head(const IO<float>& io, uint32_t pair, float distance, uint32_t nc, int, int, bool overflow) {
  hfcl_result_f32* r = &io.out[pair];
  r->distance = distance;
  r->status = (nc ? 128u : 0u) | (overflow ? 0xC0000000u : 0u);
}

template <typename T>
__device__ __forceinline__ void write_guess(const IO<T>&, uint32_t, const V3<T>&, int, int) {}
template <>
__device__ __forceinline__ void write_guess<double>(const IO<double>& io, uint32_t pair, const V3<double>& g, int h0, int h1) {
  if (io.gout) {
    hfcl_guess r;
    r.gjk_guess[0] = g.x; r.gjk_guess[1] = g.y; r.gjk_guess[2] = g.z;
    r.support_guess[0] = h0;
    r.support_guess[1] = h1;
    io.gout[pair] = r;
  }
}

template <typename T>
__device__ __forceinline__ V3<T> initial_guess(const IO<T>& io, const QParams<T>& q, uint32_t pair) {
  if (q.guess_mode == HFCL_GUESS_CACHED || q.guess_mode == HFCL_GUESS_BOUNDING_VOLUME)
    return mk<T>(q.guess[0], q.guess[1], q.guess[2]);  // BoundingVolumeGuess: the solver's cached guess = the request's
  return mk<T>(T(1), T(0), T(0));
}
template <>
__device__ __forceinline__ V3<double> initial_guess<double>(const IO<double>& io, const QParams<double>& q, uint32_t pair) {
  if (q.guess_mode == HFCL_GUESS_CACHED) {
    if (io.gin) return mk<double>(io.gin[pair].gjk_guess[0], io.gin[pair].gjk_guess[1], io.gin[pair].gjk_guess[2]);
    return mk<double>(q.guess[0], q.guess[1], q.guess[2]);
  }
  if (q.guess_mode == HFCL_GUESS_BOUNDING_VOLUME) return mk<double>(q.guess[0], q.guess[1], q.guess[2]);
  return mk<double>(1.0, 0.0, 0.0);
}
// ---------------------------------------------------------------------------------------
// Convex hull held by a W-lane group: lane l owns vertices [l*VPL, (l+1)*VPL).
// getShapeSupportLinear (support_functions.cpp:400-421): first index of the maximum dot.
// ---------------------------------------------------------------------------------------
// Partner value for stage M of an all-reduce over an aligned W-lane group (see butterfly_stages).  Stages
// within a row of 16 lanes are DPP moves (VALU rate, no LDS-pipe round trip as ds_bpermute has):
// quad_perm [1,0,3,2] / [2,3,0,1] for M = 1 / 2, row_half_mirror (lane ^ 7) for M = 4, row_mirror
// (lane ^ 15) for M = 8; wider stages go through __shfl_xor.
template <int CTRL, class X>
__device__ __forceinline__ X dpp_move(X v) {
  static_assert(sizeof(X) % 4 == 0, "32-bit words");
  int w[sizeof(X) / 4];
  __builtin_memcpy(w, &v, sizeof(X));
#pragma unroll
  for (int i = 0; i < int(sizeof(X) / 4); ++i) w[i] = __builtin_amdgcn_update_dpp(w[i], w[i], CTRL, 0xF, 0xF, false);
  X r;
  __builtin_memcpy(&r, w, sizeof(X));
  return r;
}
template <int W, int M, class X>
__device__ __forceinline__ X group_exchange(X v) {
  static_assert(M >= 1 && M < W, "stage of a W-lane butterfly");
  if constexpr (M == 1) return dpp_move<0xB1>(v);
  else if constexpr (M == 2) return dpp_move<0x4E>(v);
  else if constexpr (M == 4) return dpp_move<0x141>(v);
  else if constexpr (M == 8) return dpp_move<0x140>(v);
  else {
    int w[sizeof(X) / 4];
    __builtin_memcpy(w, &v, sizeof(X));
#pragma unroll
    for (int i = 0; i < int(sizeof(X) / 4); ++i) w[i] = __shfl_xor(w[i], M, W);
    X r;
    __builtin_memcpy(&r, w, sizeof(X));
    return r;
  }
}

constexpr int HULL_MAX = 32;  // ConvexBase::num_vertices_large_convex_threshold (geometric_shapes.h:709)
constexpr int HULL_LARGE_MAX = 1 << 16;  // hulls above HULL_MAX are scanned from memory (k_gjk_large)

template <typename T, int W>
struct HullRegs {
  static constexpr int VPL = (HULL_MAX + W - 1) / W;
  V3<T> v[VPL];

  __device__ __forceinline__ void load(const T* verts, uint32_t n, int lig) {
#pragma unroll
    for (int k = 0; k < VPL; ++k) {
      uint32_t idx = uint32_t(lig * VPL + k);
      idx = idx < n ? idx : 0u;  // padding duplicates vertex 0 (never wins the first-index tie-break)
      const T* p = verts + 3 * size_t(idx);
      v[k] = mk<T>(p[0], p[1], p[2]);
    }
  }
  // idx_out: the index of the returned vertex in the hull (first index of the maximum)
  __device__ __forceinline__ V3<T> support(const V3<T>& dir, int lig, int* idx_out = nullptr) const {
    T best = dot(v[0], dir);
    int bi = lig * VPL;
#pragma unroll
    for (int k = 1; k < VPL; ++k) {
      const T d = dot(v[k], dir);
      if (d > best) {
        best = d;
        bi = lig * VPL + k;
      }
    }
    butterfly_stages<W>([&](auto stage) {
      constexpr int M = decltype(stage)::value;
      const T od = group_exchange<W, M>(best);
      const int oi = group_exchange<W, M>(bi);
      // (selects, not a short-circuit: as `if (a || (b && c))` every stage of every support was three exec-masked branches)
      const bool take = (od > best) | ((od == best) & (oi < bi));
      best = take ? od : best;
      bi = take ? oi : bi;
    });
    // the winner's coordinates come from a run-time lane (ds_bpermute): carrying them through the stages,
    // or OR-reducing the owner's bits, costs the GJK kernels registers they do not have (spills; measured)
    const int owner = bi / VPL, slot = bi % VPL;
    V3<T> c = v[0];
#pragma unroll
    for (int k = 1; k < VPL; ++k)
      if (slot == k) c = v[k];
    if (idx_out) *idx_out = bi;
    return mk<T>(__shfl(c.x, owner, W), __shfl(c.y, owner, W), __shfl(c.z, owner, W));
  }
};
// ---------------------------------------------------------------------------------------
// k_gjk_large: GJK for pairs with a hull of more than 32 vertices (either side; the other side may be
// any convex kind).  One pair per LW-lane group, vertices streamed from memory (L2-resident).
// ---------------------------------------------------------------------------------------
// Linear-scan support of a hull too large for registers: lane l of the W-lane group looks at vertices
// l, l+W, ... (coalesced), the group reduces to the first index of the maximum (the tie rule of
// getShapeSupportLinear; the reference's neighbour hill-climbing, support_functions.cpp:323-397, reaches
// a vertex of the same support value, possibly another one on a plateau -- see DESIGN.md).
template <typename T, int W>
__device__ __forceinline__ V3<T> scan_support(const T* v, uint32_t n, const V3<T>& dir, int lig, uint32_t* idx_out = nullptr) {
  T best = -Lim<T>::max();
  uint32_t bi = 0xFFFFFFFFu;
  // The scan is a chain of loads unless several vertices are in flight at once: UNR vertices per lane are fetched
  // before any of them is compared (profiles/r02_o: 16 384-vertex hulls 1.46 ms -> see there per scan); the comparison
  // order is the index order, so the first index of the maximum still wins.
  constexpr int UNR = 8;
  uint32_t i = uint32_t(lig);
  for (; i + uint32_t((UNR - 1) * W) < n; i += uint32_t(UNR * W)) {
    T d[UNR];
#pragma unroll
    for (int u = 0; u < UNR; ++u) {
      const T* p = v + 3 * size_t(i + uint32_t(u * W));
      d[u] = p[0] * dir.x + p[1] * dir.y + p[2] * dir.z;
    }
#pragma unroll
    for (int u = 0; u < UNR; ++u)
      if (d[u] > best) {
        best = d[u];
        bi = i + uint32_t(u * W);
      }
  }
  for (; i < n; i += W) {
    const T d = v[3 * i] * dir.x + v[3 * i + 1] * dir.y + v[3 * i + 2] * dir.z;
    if (d > best) {
      best = d;
      bi = i;
    }
  }
  butterfly_stages<W>([&](auto stage) {
    constexpr int M = decltype(stage)::value;
    const T od = group_exchange<W, M>(best);
    const uint32_t oi = group_exchange<W, M>(bi);
    const bool take = (od > best) | ((od == best) & (oi < bi));
    best = take ? od : best;
    bi = take ? oi : bi;
  });
  if (idx_out) *idx_out = bi;
  return mk<T>(v[3 * bi], v[3 * bi + 1], v[3 * bi + 2]);
}
// ---------------------------------------------------------------------------------------
// Hill-climbing support over the vertex adjacency (getShapeSupportLog, support_functions.cpp:323-397): start at `hint`
// (the vertex the previous call of this query returned; < 0: the best of the hull's 14 warm-start vertices,
// ConvexBase::support_warm_starts, geometric_shapes.cpp buildSupportWarmStart), look at all neighbours of the current
// vertex at once -- one per lane, coordinates inline in the adjacency entry so a hop is two dependent fetches (offsets,
// entries) -- and move to the best one while it is strictly better.  On a convex polytope whose listed points are all
// extreme that ends at a vertex of maximal support (no improving edge = optimal).  A point in the interior of a flat,
// triangulated facet (a subdivided box, the cap centre of a cylinder mesh) has all its neighbours in the facet's plane:
// along the facet's INWARD normal every neighbour ties and none improves although the facet is the hull's minimum.
// The reference walks on over equal neighbours until its first strict improvement (loose_check, with a visited set to
// stay finite, :368-382); here a climb that ends where it started, without any strict improvement and with a neighbour
// that ties, is that plateau case and is answered by the scan (the only place a tie can hide a better vertex: after a
// strict improvement a plateau reached is a maximum).  Where several vertices tie at the maximum the result may be
// another of them than the scan's / the reference's.
// A hop costs O(degree) instead of O(num_points): measured crossover against scan_support in profiles/r02_p.
template <typename T, int W>
__device__ __forceinline__ V3<T> climb_support(const T* v, uint32_t n, const HullGraph<T>& g, const V3<T>& dir, int lig, int& hint) {
  uint32_t cur;
  T best;
  if (hint < 0) {
    best = -Lim<T>::max();
    cur = 0;
    uint32_t bk = 0xFFFFFFFFu;
    for (int k = lig; k < HFCL_WARM_STARTS; k += W) {
      const uint32_t id = g.off[k - HFCL_WARM_STARTS];
      const T d = v[3 * size_t(id)] * dir.x + v[3 * size_t(id) + 1] * dir.y + v[3 * size_t(id) + 2] * dir.z;
      if (d > best) {
        best = d;
        cur = id;
        bk = uint32_t(k);
      }
    }
    butterfly_stages<W>([&](auto stage) {
      constexpr int M = decltype(stage)::value;
      const T od = group_exchange<W, M>(best);
      const uint32_t oc = group_exchange<W, M>(cur);
      const uint32_t ok = group_exchange<W, M>(bk);
      const bool take = (od > best) | ((od == best) & (ok < bk));
      best = take ? od : best;
      cur = take ? oc : cur;
      bk = take ? ok : bk;
    });
  } else {
    cur = uint32_t(hint);
    best = v[3 * size_t(cur)] * dir.x + v[3 * size_t(cur) + 1] * dir.y + v[3 * size_t(cur) + 2] * dir.z;
  }
  bool improved = false;
  for (;;) {
    const uint32_t b = g.off[cur], e = g.off[cur + 1];
    T mb = best;
    uint32_t mi = 0xFFFFFFFFu, mpos = 0xFFFFFFFFu;
    bool tie = false;
    for (uint32_t k = b + uint32_t(lig); k < e; k += W) {
      const NbrEntry<T> nb = g.ent[k];
      const T d = nb.x * dir.x + nb.y * dir.y + nb.z * dir.z;
      tie = tie || d == best;
      if (d > mb) {
        mb = d;
        mi = nb.id;
        mpos = k;
      }
    }
    butterfly_stages<W>([&](auto stage) {
      constexpr int M = decltype(stage)::value;
      const T od = group_exchange<W, M>(mb);
      const uint32_t oi = group_exchange<W, M>(mi);
      const uint32_t ok = group_exchange<W, M>(mpos);
      const bool take = (od > mb) | ((od == mb) & (ok < mpos));
      mb = take ? od : mb;
      mi = take ? oi : mi;
      mpos = take ? ok : mpos;
    });
    if (mi == 0xFFFFFFFFu) {  // no neighbour is strictly better (uniform over the group)
      if (!improved) {        // ... and none ever was: a neighbour that ties may hide a better vertex behind the plateau
        uint32_t any = tie ? 1u : 0u;
        butterfly_stages<W>([&](auto stage) {
          constexpr int M = decltype(stage)::value;
          any |= group_exchange<W, M>(any);
        });
        if (any) {
          uint32_t si = 0;
          const V3<T> r = scan_support<T, W>(v, n, dir, lig, &si);
          hint = int(si);
          return r;
        }
      }
      break;
    }
    improved = true;
    cur = mi;
    best = mb;
  }
  hint = int(cur);
  return mk<T>(v[3 * size_t(cur)], v[3 * size_t(cur) + 1], v[3 * size_t(cur) + 2]);
}
// scan or climb: what a hull too large for registers answers a support query with
template <typename T, int W>
__device__ __forceinline__ V3<T> large_hull_support(const T* v, uint32_t n, const HullGraph<T>& g, const V3<T>& dir, int lig, int& hint) {
  if (g.off != nullptr) return climb_support<T, W>(v, n, g, dir, lig, hint);
  return scan_support<T, W>(v, n, dir, lig);
}
// ---------------------------------------------------------------------------------------
// k_epa: EPA on the pairs GJK left in `Collision`.  One polytope per WE-lane group, 64/WE polytopes
// per wavefront, scratch blocks in LDS.  Two tiers:
//   tier 1  fp32: WE = 8, CAP = 17, 8 polytopes per wave share one instruction stream (streaming form, k_epa_stream);
//           fp64: one kernel per class of pairs -- polytope pairs WE = 8, CAP = 13; pairs with a curved shape WE = 16,
//           CAP = 29 -- each 20 KB of LDS per wave (two waves per SIMD).  A polytope that outgrows the block is saved at
//           the start of that iteration and queued
//   tier 2  CAP = 64 (the reference capacity), WE = 16 (fp32) / 32 (fp64): continues the saved polytopes
// ---------------------------------------------------------------------------------------
template <int W_>
struct LaneGroup {
  static constexpr int W = W_;
  static __device__ __forceinline__ int lane() { return threadIdx.x & (W_ - 1); }
  template <int M, class X> static __device__ __forceinline__ X exchange(X v) { return group_exchange<W_, M>(v); }
  // Lanes of a group exchange data through LDS: the wavefront-scope fence keeps the compiler from moving or
  // reusing LDS accesses across the exchange point (the barrier alone only pins instruction scheduling).
  static __device__ __forceinline__ void sync() {
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    __builtin_amdgcn_wave_barrier();
  }
  static __device__ __forceinline__ uint32_t atomic_inc(uint32_t* p) { return atomicAdd(p, 1u); }  // LDS (ds_add_rtn)
  // the lanes of this group for which `pred` holds, as bits of the wave's ballot ...
  static __device__ __forceinline__ uint64_t group_bits(bool pred) {
    const uint64_t mine = (W_ == 64 ? ~uint64_t(0) : ((uint64_t(1) << (W_ & 63)) - 1u)) << ((threadIdx.x & 63) & ~(W_ - 1));
    return __ballot(pred) & mine;
  }
  // ... as a mask of the group (bit = lane in group) ...
  static __device__ __forceinline__ uint64_t ballot(bool pred) { return group_bits(pred) >> ((threadIdx.x & 63) & ~(W_ - 1)); }
  // ... and counted: below the calling lane, in the whole group
  static __device__ __forceinline__ void count(bool pred, int& below, int& total) {
    const uint64_t b = group_bits(pred);
    below = int(__builtin_amdgcn_mbcnt_hi(uint32_t(b >> 32), __builtin_amdgcn_mbcnt_lo(uint32_t(b), 0u)));
    total = __popcll(b);
  }
};

// Capacity (iterations) of the fast tier's block.  fp32: 17 -- the convex x convex form (k_epa_stream<.., CC>) then has
// a 1568-byte block per polytope (no shape-0 support points: tags; 12-byte face topology; one-byte horizon entries),
// 12544 bytes per wave of 8 = ten of the 1280-byte units the hardware hands LDS out in (tools/occupancy_probe.hip), so
// 12 waves fit a CU (3 per SIMD).  18 iterations are 13184 bytes = 11 units = 11 waves.
#ifndef HFCL_EPA_FAST_CAP
#define HFCL_EPA_FAST_CAP 17
#endif
constexpr int EPA_FAST_CAP = HFCL_EPA_FAST_CAP;
#ifndef HFCL_EPA_FAST_CAP64
#define HFCL_EPA_FAST_CAP64 29  // the largest whose block (40 832 B) still puts four waves on a CU (32 of the 128 LDS units each); cfg5 fast + full ms: 24: 1.20+0.89, 26: 1.34+0.64, 28: 1.45+0.51, 29: 1.45+0.46
#endif
// fp64: the pairs without a curved shape (their own queue, finish_gjk) end after 2-8 iterations on average; a block for 13
// is 20 208 B per wave of 8 = 16 LDS units, which puts two waves on a SIMD where the curved class's block allows one
#ifndef HFCL_EPA_SMALL_CAP64
#define HFCL_EPA_SMALL_CAP64 13
#endif
// capacity of the fast tier's block per precision (fp64: of the curved class), and of the fp64 polytope class
template <typename T> constexpr int epa_fast_cap = sizeof(T) == 4 ? EPA_FAST_CAP : HFCL_EPA_FAST_CAP64;
template <typename T> constexpr int epa_small_cap = sizeof(T) == 4 ? EPA_FAST_CAP : HFCL_EPA_SMALL_CAP64;
// one slot of the hand-over area
template <typename T> constexpr size_t epa_resume_stride = sizeof(EpaSaved<T, epa_fast_cap<T>>);
#ifndef HFCL_EPA_WE
#define HFCL_EPA_WE 8
#endif
constexpr int EPA_WE = HFCL_EPA_WE;
#ifndef HFCL_EPA_WE2
#define HFCL_EPA_WE2 16
#endif
constexpr int EPA_WE2 = HFCL_EPA_WE2;  // lanes per polytope in the full-capacity tier (fp32)
#ifndef HFCL_EPA_WE2_64
#define HFCL_EPA_WE2_64 32  // fp64: 2 polytopes x 8.6 KB per wave = two waves per SIMD (16 lanes: 4 polytopes, 34 KB, one wave); cfg5 k_epa<full> 0.445 -> 0.35 ms
#endif
template <typename T> constexpr int epa_we2 = sizeof(T) == 4 ? HFCL_EPA_WE2 : HFCL_EPA_WE2_64;
#ifndef HFCL_EPA_CC_RESUME_WE
#define HFCL_EPA_CC_RESUME_WE 32  // lanes per polytope of k_epa_resume_cc (hfcl_k_epa.hip)
#endif

// ---------------------------------------------------------------------------------------
// k_bvh_collide: BVHModel<OBBRSS> x BVHModel<OBBRSS> collide().
// Traversal = collisionRecurse (src/traversal/traversal_recurse.cpp:44-85) with the recursion
// flattened into a per-lane LDS stack; children are pushed right-then-left so they pop in the
// reference's order, and the walk ends as soon as num_max_contacts contacts exist (canStop()).
// ---------------------------------------------------------------------------------------
struct DMesh {
  uint32_t node_off, vert_off, tri_off, n_nodes;
};
template <typename T>
struct BvhView {
  const DNode<T>* nodes;
  const DNodeF* fnodes;  // filter records of the same nodes (hfcl_bvh.hpp: obb_filter); nullptr: plain fp64 tests
  const DRss<T>* rss;
  const DNodeD<T>* dnodes;  // the distance() walk's packed node records (hfcl_bvh.hpp)
  const T* verts;        // xyz
  const uint32_t* tris;  // 3 local vertex ids per triangle
  const DMesh* meshes;
  uint32_t n_meshes;
};
struct BvhParams {
  uint32_t num_max_contacts;
  hfcl_contact* contacts;   // optional device contact list
  uint32_t contacts_cap;
  uint32_t* contacts_count;
};

// ---- long mesh x mesh traversals cut into tasks (k_bvh_collide / k_bvh_combine, hfcl_k_bvh.hip) ----
struct BvhTask {
  uint32_t pair;    // the query
  uint32_t parent;  // summary slot of the unit that made the task
  uint32_t entry;   // the subtree pair to walk (b1 | b2 << 16); 0xFFFFFFFF: no-op (the task table was full)
  uint32_t order;   // position among the parent's children (0 = visited first)
};
enum { BVH_SUM_SUSPENDED = 1u, BVH_SUM_OVERFLOW = 2u };
template <typename T>
struct BvhSum {  // what a unit (query or task) knows when it ends or suspends
  T dlb, rec_dist;   // lower bound over its events, and the record distance that goes with it
  T cand_val;        // value of its last leaf that lowered the bound (max(): none) ...
  V3<T> np1, np2, nn;  // ... and that leaf's witness points / normal
  int32_t fb1, fb2;  // first contact
  uint32_t ncontacts, first_child, n_child, flags;
  // cancellation of speculative work: the smallest `order` among this unit's children that found a contact (the walk
  // ends there: children after it, and everything below them, are moot), and where the unit itself hangs
  uint32_t contact_order, parent, order, pad_;
};
// Forms that lost their A/B and stay only as identity references for the tests -- the fp32 filter in front of the fp64 box test
// (k_bvh_collide<.., FILT>, profiles/r03_b), the general EPA queues in three stages (k_epa_*_general, profiles/r05_e) -- are compiled
// only with -DHFCL_KEEP_AB_FORMS=1 (tools/build_variant.sh ab host,k_bvh,k_epa -DHFCL_KEEP_AB_FORMS=1 -> build/ab/lib_ab.so, selected with
// HFCL_LIB_PATH): the product library does not carry them, their options are refused there (hfcl_has_ab_forms() says which build this is).
#ifndef HFCL_KEEP_AB_FORMS
#define HFCL_KEEP_AB_FORMS 0
#endif
// counters of a split traversal (device words)
enum { BVH_CTR_TASKS = 0, BVH_CTR_SUSPENDED = 1, BVH_CTR_LEVEL0 = 2 /* [2 + k] = number of tasks made before level k ended */, BVH_CTR_CUT = 15 /* words of BvhSplit::cut_words in use */, BVH_CTR_WORDS = 16 };
// ---------------------------------------------------------------------------------------
// Mesh x mesh collide() with the walk and the leaves in kernels of their own (round 6; hfcl_k_bvh.hip: k_bvh_walk, k_tri_leaves,
// k_bvh_resolve).  collisionRecurse's box tests do not depend on anything a triangle pair reports -- only WHERE the walk ends does (the
// first contact) -- so a lane can walk ahead of its triangle pairs: it lists the next `k` leaves it meets (with the smallest box bound
// seen in front of each: the events between two leaves only ever form a minimum), a dense kernel evaluates every listed pair of the
// batch, one per lane, and a third replays each query's events in the reference's order: bound, witness, first contact.  A query that
// is not over after a round -- its `k` leaves listed without a contact -- walks on in the next; what is left after the last round, and
// every walk that used up its step budget, goes to k_bvh_coop with its stack as before.
// What it buys: the walk without the leaf's GJK fits 3-4 waves per SIMD instead of 2 and no lane waits for another's leaf (the probe
// of round 3: 2.4x the steps per second); the leaves run 64 to a wave instead of the 5-10 a parked wave held.  What it costs: the
// steps and leaves a walk takes beyond its first contact (at most k - 1 leaves of one round).
// ---------------------------------------------------------------------------------------
constexpr int BVH_STACK_WALK = 48;  // k_bvh_walk's LDS stack (entries per lane) and its image in WalkRec
constexpr int WALK_K = 16;       // most leaves a walk lists per round
constexpr int WALK_ROUNDS = 4;   // most rounds
enum { WALK_OVER = 1u, WALK_BUDGET = 2u, WALK_LOST = 4u /* its items did not fit the list */, WALK_VOID = 8u /* mesh x solid: flagged unsupported by the walk, no record to write */ };
template <typename T>
struct WalkRec {  // one per query of the batch
  uint32_t pair, sp, n_leaf, flags;  // flags: the stack ran empty (WALK_OVER) / the step budget or the stack's capacity ended the round (WALK_BUDGET)
  uint32_t first_item, pad_[3];      // its leaves of this round are items first_item ... first_item + n_leaf - 1
  T dlb, rec_dist, cand_val;         // the query's state after the rounds resolved so far (the witness lives in its record)
  T pre[WALK_K + 1];                 // smallest bound of the disjoint boxes in front of leaf i (max(): none); [n_leaf]: behind the last leaf
  uint32_t leaf1[WALK_K], leaf2[WALK_K];  // triangle ids
  uint32_t stack[BVH_STACK_WALK];    // bottom first
};
struct WalkArgs {
  void* recs;         // WalkRec<T>[n]
  uint32_t* items;    // rec index | slot << 28, WALK_K per query at most
  void* res;          // TriLeafOut<T> per item
  uint32_t* ctr;      // per round r at ctr + 8 r: [0] ticket of the walk, [1] items listed, [2] queries that walk on in round r + 1,
                      // [3] the suspended queries as round r left them (k_walk_snap), [4] the ticket of the launch that continues those round r added
  uint32_t* list_in;  // rec indices of this round's queries (round 0: the B_BVH bucket, rec index = position in the bucket's list)
  uint32_t* list_out;
  uint32_t* redo;     // mesh x solid: the items (as in `items`) whose leaf ended its walk needing EPA; their number at ctr[8 r + 5], list_stride entries
  uint32_t* perm;     // mesh x solid: the items ordered by the kind of their solid (item_cap entries, then list_stride for the redo list); nullptr: as listed
  uint32_t* hist;     // ... [0..15] items per kind, [16..31] the scatter's cursors; [32..63] the same for the redo list (zeroed by the host)
  uint32_t round, k, budget, last;
  uint32_t item_cap, list_stride;  // entries of items / res; of one of the two lists list_in / list_out alternate between
};
struct BvhSplit {
  BvhTask* tasks;       // task table (cap entries)
  void* sums;           // BvhSum<T>[n_queries + cap]: suspended queries first, then one per task
  uint32_t* suspended;  // pairs of the suspended queries (slot i <-> sums[i])
  uint32_t* ctr;        // BVH_CTR_*
  uint32_t cap, n_queries;
  uint32_t budget;      // BV-test steps before a unit suspends (0: never)
  uint32_t budget0;     // ... of the queries themselves (level 0); `budget` is that of the tasks
  uint32_t level, n_levels;
  uint32_t can_suspend;
  uint32_t leaf_cost;   // SOLID form: steps a GJK leaf counts for (a closed-form leaf: an eighth of it)
  uint32_t coop;        // suspended queries are continued by k_bvh_shape_coop / k_bvh_coop (a lane group per query) instead of task levels
  uint32_t coop_grid;   // ... blocks of that kernel (0: the launcher's choice)
  // A walk one of those kernels has worked on for `cut_ticks` clock ticks is CUT: its stack, in chunks of COOP_CHUNK entries, becomes
  // tasks for the next launch of the same kernel (a chunk task's `entry` is the index of its first word in cut_words, its `order`
  // carries the number of its entries in the high half), its state a summary, and k_bvh_combine folds the chunks' summaries back in
  // DFS order as it does for k_bvh_collide's task levels.  0: never (the walk stays with its wave to the end).
  uint32_t cut_ticks;
  uint32_t cut_cap;     // words of cut_words (and values of cut_vals)
  uint32_t cut_task_cap;  // chunk tasks the cuts of a batch may make in all (mesh x solid: every chunk is a unit that can queue one EPA item, and
                          // the host sized that queue for n queries + this many chunks; a walk that would exceed it is not cut)
  uint32_t* cut_words;  // stack entries of the cut walks
  void* cut_vals;       // T[cut_cap]: what is known about them (k_bvh_shape_coop's tagged entries: a box's bound, a triangle's distance)
  // mesh x mesh: the queries' own phase as walk / leaves / resolve rounds (walk.recs != nullptr) instead of k_bvh_collide with its budget
  WalkArgs walk;
  uint32_t walk_rounds, walk_k[WALK_ROUNDS], walk_budget[WALK_ROUNDS];
  // which of the suspended queries a launch of k_bvh_coop continues: all (0); those round r of the walk handed over (1 + r: beside the
  // later rounds, on a stream of its own, with walk.ctr[8 r + 4] as its ticket); those added since the snapshot of round r (0x100 + r:
  // what the last round added, on the caller's stream)
  uint32_t coop_range;
  // ... and in which order its waves draw them: order[t] = the suspended slot of ticket t (k_walk_order: the queries with the most
  // stack entries left first -- a wave draws its next walk when the last is over, so a long walk drawn last is the launch's tail);
  // nullptr: in the order they were suspended
  uint32_t* order;
};
enum { WALK_CTR_SNAP = 3, WALK_CTR_TICKET_EARLY = 4 };
constexpr int COOP_CHUNK = 16;


// Step budget per unit (the compile-time default; without HFCL_BVH_* in the environment the host chooses per batch, see
// hfcl_lib::bvh_auto).  0: units only suspend when their LDS stack is full -- the task mechanism is then the overflow path
// of deep traversals and costs nothing otherwise.  One budget for all levels does not pay (profiles/r02_k: cfg4 7.3 ms
// unsplit, 7.9 .. 30 ms split: a stack is one heavy subtree next to many small ones, every level halves the longest chain
// at best, small budgets drown in tasks).  What pays for batches that do not fill the chip's lanes is a generous budget for
// the queries themselves (level 0) and a small one for the tasks of their remainders (profiles/r02_w).
#ifndef HFCL_BVH_BUDGET
#define HFCL_BVH_BUDGET 0
#endif
#ifndef HFCL_BVH_BUDGET0
#define HFCL_BVH_BUDGET0 HFCL_BVH_BUDGET
#endif
// Global-memory continuation of the per-lane traversal stacks (models with more than 65535 BV nodes): `cap` entries per
// lane of the launch, sized by the host from the depths of the registered models; the grid is limited to `max_blocks` so
// that the slabs of all lanes fit the allocation.
struct BvhSpill {
  void* slab;
  uint32_t cap;         // entries per lane
  uint32_t max_blocks;  // blocks the slab allocation covers
  uint32_t wide;        // 32-bit node ids in the stack entries
  // distance(), narrow form: a query that has taken `budget` steps leaves its state and its stack in a DistSusp record
  // (susp[slot], slot from *susp_count) and is continued by k_bvh_distance_coop; budget 0: walks stay with their lanes
  void* susp;
  uint32_t* susp_count;
  uint32_t budget;
  // ... or, pool != 0, by k_bvh_distance_pool (hfcl_k_bvhd.hip): several walks per wave, their box and triangle tests pooled
  uint32_t pool;
  uint32_t* pool_ticket;
  uint32_t pool_leaf_min, pool_starve, pool_part_min;  // its scheduling knobs
  // The pooled continuations evaluate a walk's entries out of the reference's order.  Their result is the sequential walk's unless a
  // pair that set the minimum stands under a bound that exceeds its own distance (the rounding of two different computations: an ulp of
  // the scene's size).  Such a walk is not written: its slot takes the suspended record again and walks it in the reference's order
  // (the "ordered" mode of the pool kernels), bounded by what the pooled pass found.  Non-null = enabled; counts those walks.
  uint32_t* rerun_count;
  uint32_t rerun_all;  // (test knob) every pooled walk is walked again in order
};
constexpr int BVH_MAX_LEVELS = 12;  // most task levels a batch can be given (HFCL_BVH_LEVELS, the automatic choice)
#ifndef HFCL_BVH_LEVELS
#define HFCL_BVH_LEVELS 6
#endif

// LDS stack entries per lane of k_bvh_collide.  48 (it was 96): with the witness slab of the leaf GJK the block is 36 KB =
// 29 allocation units, so that the four blocks of two waves per SIMD fit a CU; a 5000-triangle model needs ~30
// (depth1 + depth2 + 2), deeper traversals suspend into tasks (HFCL_BVH_LEVELS levels) or take the wide form.
#ifndef HFCL_BVH_STACK
#define HFCL_BVH_STACK 48
#endif
constexpr int BVH_STACK = HFCL_BVH_STACK;

// WIDE: models of more than 65535 BV nodes.  A stack entry is then two 32-bit node ids (the LDS stack holds half as
// many), and a lane whose LDS stack is full moves its lower half to a slab of its own in global memory (BvhSpill) and
// takes it back when the LDS part runs empty: the reference's stack is a growable std::vector
// (traversal_recurse.cpp:95), a traversal here is bounded by the slab the host sized from the depths of the models.
template <bool WIDE> struct BvhEntry;
template <> struct BvhEntry<false> {
  typedef uint32_t E;
  static constexpr int STACK = BVH_STACK;
  static __device__ __forceinline__ E pack(uint32_t b1, uint32_t b2) { return b1 | (b2 << 16); }
  static __device__ __forceinline__ uint32_t first(E e) { return e & 0xFFFFu; }
  static __device__ __forceinline__ uint32_t second(E e) { return e >> 16; }
};
template <> struct BvhEntry<true> {
  typedef uint64_t E;
  static constexpr int STACK = BVH_STACK / 2;
  static __device__ __forceinline__ E pack(uint32_t b1, uint32_t b2) { return uint64_t(b1) | (uint64_t(b2) << 32); }
  static __device__ __forceinline__ uint32_t first(E e) { return uint32_t(e); }
  static __device__ __forceinline__ uint32_t second(E e) { return uint32_t(e >> 32); }
};

#ifndef HFCL_BVH_STACK_FILT
#define HFCL_BVH_STACK_FILT 26  // 26 x 4 B x 128 lanes + the 12 KB witness slab = 25 600 B = 20 LDS units: six blocks per CU
#endif
constexpr int BVH_STACK_FILT = HFCL_BVH_STACK_FILT;
constexpr int BVH_BLOCK = 128;
#ifndef HFCL_BVH_REFILL_MIN
#define HFCL_BVH_REFILL_MIN 8
#endif
constexpr int BVH_REFILL_MIN = HFCL_BVH_REFILL_MIN;  // idle lanes of a wave that trigger a refill (4 / 8 / 16 / 24: 17.5 / 17.1 / 18.7 / 20.3 ms per 1M cfg4 queries)

// block shapes the host needs to size the grids
constexpr int CLS_BLOCK = 1024;  // k_classify
constexpr int LARGE_W = 16;      // lanes per pair in k_gjk_large
constexpr int BS_W = 16;         // lanes per query in k_bvh_shape / k_bvh_shape_distance / k_triangle
constexpr int BS_STACK = 128;
// LDS stack of k_bvh_distance: (entry, lower bound in 4 bytes) per slot; 40 slots x 8 B x 64 lanes = 20 480 B = 16 LDS units per wave:
// eight waves per CU (it was 64 slots x 12 B = 48 KB: three waves per CU).  A traversal holds at most depth1 + depth2 + 2
// entries: models up to 19 levels deep (5 000 triangles: 16); deeper ones take the wide form with its global slabs.
constexpr int BVHD_STACK = 40;
constexpr int BVHD_BLOCK = 64;   // k_bvh_distance
// The pooled distance() continuations keep, per stack entry, the largest bound among the entry's ancestors where it exceeds the entry's own
// (0: none does) as a float rounded UP: an upper estimate, exact in the case that matters (an entry whose own bound is the largest of its
// chain).  (In 16 bits -- 0.8 % above -- half of cfg4d's walks are flagged, profiles/r06_a.)
__device__ __forceinline__ float chain_up(double d) { return __double2float_ru(d); }
__device__ __forceinline__ float chain_up(float d) { return d; }
template <typename T>
struct ShapeDistSusp {  // a suspended mesh x solid distance() walk (the witness of the minimum is in the query's record)
  uint32_t pair, sp;
  int32_t fb1, pad_;
  T mind;
  uint32_t entry[BVHD_STACK];  // mesh nodes, bottom first
  T bound[BVHD_STACK];
};
template <typename T>
struct DistSusp {  // a suspended mesh x mesh distance() walk
  uint32_t pair, sp;
  int32_t fb1, fb2;
  T mind;
  V3<T> np1, np2;                // witness of the minimum, model-1 frame
  uint32_t entry[BVHD_STACK];    // stack, bottom first
  T bound[BVHD_STACK];
};

// ---------------------------------------------------------------------------------------
// Witness payload of the GJK simplex parked in LDS (policy of gjk_run / gjk_finish, hfcl_pair.hpp).  The support
// point on shape 0 of each simplex vertex is only read after the loop, but as a register payload it is 12 VGPRs
// (24 in fp64) that every simplex shift / Voronoi-region select drags along -- state the register allocator ends up
// spilling to scratch memory, i.e. through L2 to HBM.  Here a vertex carries a 2-bit slot number instead and the
// point sits in a per-lane LDS slab (slot-major, lane-minor: conflict-free); a new vertex takes the slot no live
// vertex uses (at most 3 are live when one is appended).
// ---------------------------------------------------------------------------------------
struct PSlot {
  uint32_t s;
};
__device__ __forceinline__ PSlot psel(bool c, const PSlot& a, const PSlot& b) { return PSlot{c ? a.s : b.s}; }
template <typename T, int NT>
struct W0Lds {
  typedef PSlot P;
  static constexpr int WORDS = 4 * 3 * NT;  // slab size in T
  T* lane;                                   // &slab[threadIdx.x]
  template <class G>
  __device__ __forceinline__ P put(const G& g, const V3<T>& w0) const {
    uint32_t used = g.rank > 0 ? 1u << (g.s0.p.s & 3u) : 0u;
    used |= g.rank > 1 ? 1u << (g.s1.p.s & 3u) : 0u;
    used |= g.rank > 2 ? 1u << (g.s2.p.s & 3u) : 0u;
    const uint32_t s = uint32_t(__builtin_ctz(~used)) & 3u;
    lane[(3 * s + 0) * NT] = w0.x;
    lane[(3 * s + 1) * NT] = w0.y;
    lane[(3 * s + 2) * NT] = w0.z;
    return P{s};
  }
  __device__ __forceinline__ V3<T> get(const P& p) const {
    const uint32_t s = p.s & 3u;
    return mk<T>(lane[(3 * s + 0) * NT], lane[(3 * s + 1) * NT], lane[(3 * s + 2) * NT]);
  }
};
#ifndef HFCL_GJK_W0_LDS
#define HFCL_GJK_W0_LDS 1
#endif
