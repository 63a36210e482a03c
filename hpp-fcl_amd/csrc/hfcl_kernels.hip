// hfcl_kernels.hip -- HIP kernels (gfx950 / CDNA4, wave64) and the C-ABI implementation of
// include/hppfcl_amd.h.  No CPU fallback anywhere in this file: every compute entry point
// needs a HIP device and fails loudly without one.
//
// Kernel map (pair buckets follow the reference's dispatch table,
// include/hpp/fcl/internal/shape_shape_func.h:185-211 and src/collision_func_matrix.cpp:279-733):
//   k_classify       pair -> bucket lists (block-aggregated atomics), one pass over the shape ids
//   k_closed<T>      closed forms (sphere / capsule / cylinder / box-sphere pairs, every Plane / Halfspace
//                    row), one pair per lane
//   k_gjk_prim<T>    GJK for Box/Capsule/Cone/Cylinder/Ellipsoid/Sphere pairs, one pair per lane
//   k_gjk_cvx<W,M>   GJK with hulls of <= 32 vertices: one pair per W-lane group, hull vertices in the
//                    group's registers, support = per-lane dots + DPP-butterfly arg-max (fp32 / fp64
//                    entry points with their own register budgets)
//   k_gjk_large<T>   GJK when a hull has more than 32 vertices: 16-lane groups scan the vertices from memory
//   k_epa<T,WE,CAP,TIER>  EPA on the pairs GJK left in `Collision`: one polytope per WE-lane group in LDS;
//                    tier 1 = 8 polytopes per wave in small blocks, tier 2 = full capacity (continues the
//                    polytopes tier 1 saved when they outgrew their block; every pair with a large hull)
//   k_epa_stream<T,WE,CAP>  tier 1 for fp32: same blocks, but a lane group whose polytope is done starts the
//                    wave's next item instead of waiting for the slowest of the 8
//   k_bvh_collide<T> / k_bvh_distance<T>   BVHModel<OBBRSS> x BVHModel<OBBRSS>: one mesh pair per lane,
//                    explicit DFS stack in LDS (reference order), OBB SAT / RSS bounds, triangle-triangle leaves
//   k_bvh_shape<T> / k_bvh_shape_distance<T>   BVHModel<OBBRSS> x convex solid or Plane/Halfspace: one query
//                    per 16-lane group, sequential traversal, leaves = TriangleP-vs-solid GJK + EPA in LDS
//   k_unsupported<T> flags the pairs of a bucket the engine cannot evaluate (never computed elsewhere)
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <mutex>
#include <string>
#include <vector>

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
// k_epa carries no occupancy attribute on purpose: forcing the fp64 instantiation to 2 waves/SIMD
// (488 B/lane of scratch) produced wrong EPA results on gfx950 (profiles/r01_c_waves_per_eu_ab.txt);
// the compiler's own choice (fp32: 2 waves, fp64: 1 wave + AGPRs) is what the parity tests cover.
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
#define HFCL_WPE_EPA64 1
#endif
#ifndef HFCL_WPE_BVH
#define HFCL_WPE_BVH 1
#endif

// ---------------------------------------------------------------------------------------
// bucket ids (finer than hfcl_shapes.hpp's pair_class: the convex bucket is split by which
// side carries vertices so the kernel is specialised at compile time)
// ---------------------------------------------------------------------------------------
enum { B_CLOSED = 0, B_PRIM = 1, B_CC = 2, B_PC = 3, B_CP = 4, B_BVH = 5, B_UNSUPPORTED = 6, B_LARGE = 7, B_BVHSHAPE = 8, B_TRI = 9, B_COUNT = 10 };

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
struct LibView {
  const DShape<T>* shapes;
  const T* verts;
  const uint8_t* kinds;
  uint32_t n_shapes;
};

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
                     // + [B_COUNT+2] = ticket counter of the streaming BVH kernel
  void* epa_queue;
  void* epa_queue2;  // polytopes that outgrew the fast EPA kernel's scratch block
  void* epa_v0;      // shape-0 support points of the polytopes in flight in the full-capacity EPA kernel
  void* epa_resume;  // saved polytopes (EpaScratch<T, EPA_FAST_CAP>) of the first `resume_cap` slots of epa_queue2
  uint32_t resume_cap;
};
constexpr int32_t EPA_RESUME_FLAG = 0x100;  // EpaSeed::rank bit: "continue the saved polytope of this slot"

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
// k_classify: bucket every pair by (kind1, kind2).  Wave-aggregated list append.
// ---------------------------------------------------------------------------------------
// One global atomic per (block trip, bucket) reserves the block's range in the bucket list: these same-address
// atomics serialise (~20 ns each), so the trips are made large -- 1024 threads x 8 pairs (4M pairs: 39 us at
// 2048 pairs per trip).
constexpr int CLS_BLOCK = 1024;
__global__ void __launch_bounds__(CLS_BLOCK) k_classify(Work wk, const uint8_t* kinds, uint32_t n_shapes, bool distance_mode) {
  // Each block handles CHUNK consecutive pairs per trip: per-bucket counts are built in LDS, one
  // global atomic per (block, bucket) reserves a range, then every lane writes its pair index.
  constexpr int PER_THREAD = 8;
  constexpr uint32_t CHUNK = CLS_BLOCK * PER_THREAD;
  __shared__ uint32_t s_count[B_COUNT];
  __shared__ uint32_t s_base[B_COUNT];
  for (uint32_t start = blockIdx.x * CHUNK; start < wk.n; start += gridDim.x * CHUNK) {
    if (threadIdx.x < B_COUNT) s_count[threadIdx.x] = 0;
    __syncthreads();
    int bk[PER_THREAD];
    uint32_t rk[PER_THREAD];
#pragma unroll
    for (int k = 0; k < PER_THREAD; ++k) {
      const uint32_t i = start + k * CLS_BLOCK + threadIdx.x;
      bk[k] = -1;
      rk[k] = 0;
      if (i < wk.n) {
        const uint32_t s1 = wk.shape1[i], s2 = wk.shape2[i];
        bk[k] = (s1 < n_shapes && s2 < n_shapes) ? bucket_of(kinds[s1], kinds[s2], distance_mode) : B_UNSUPPORTED;
      }
      // wave-aggregated LDS counter update: one trip per bucket present in the wave (one for a homogeneous batch)
      unsigned long long todo = __ballot(bk[k] >= 0);
      const int lane = threadIdx.x & 63;
      while (todo) {
        const int leader = __ffsll((long long)todo) - 1;
        const int c = __shfl(bk[k], leader, 64);
        const unsigned long long m = __ballot(bk[k] == c);
        uint32_t base = 0;
        if (lane == leader) base = atomicAdd(&s_count[c], (uint32_t)__popcll(m));
        base = __shfl(base, leader, 64);
        if (bk[k] == c) rk[k] = base + (uint32_t)__popcll(m & ((1ull << lane) - 1ull));
        todo &= ~m;
      }
    }
    __syncthreads();
    if (threadIdx.x < B_COUNT) {
      const uint32_t c = s_count[threadIdx.x];
      s_base[threadIdx.x] = c ? atomicAdd(&wk.counts[threadIdx.x], c) : 0u;
    }
    __syncthreads();
#pragma unroll
    for (int k = 0; k < PER_THREAD; ++k) {
      if (bk[k] >= 0) wk.lists[size_t(bk[k]) * wk.n + s_base[bk[k]] + rk[k]] = start + k * CLS_BLOCK + threadIdx.x;
    }
    __syncthreads();
  }
}

// pairs the engine cannot evaluate: flagged, never silently computed elsewhere
template <typename T>
__global__ void __launch_bounds__(256) k_unsupported(Work wk, IO<T> io, int bucket) {
  const uint32_t cnt = wk.counts[bucket];
  for (uint32_t it = blockIdx.x * blockDim.x + threadIdx.x; it < cnt; it += gridDim.x * blockDim.x) {
    const uint32_t pair = wk.lists[size_t(bucket) * wk.n + it];
    auto r = io.out[pair];
    memset(&r, 0, sizeof(r));
    r.status = 0x80000000u;
    io.out[pair] = r;
  }
}

// ---------------------------------------------------------------------------------------
// k_closed: closed-form pairs, one pair per lane.
// ---------------------------------------------------------------------------------------
template <typename T>
__global__ void __launch_bounds__(256) k_closed(Work wk, LibView<T> lib, IO<T> io, QParams<T> q) {
  const uint32_t cnt = wk.counts[B_CLOSED];
  for (uint32_t it = blockIdx.x * blockDim.x + threadIdx.x; it < cnt; it += gridDim.x * blockDim.x) {
    const uint32_t pair = wk.lists[size_t(B_CLOSED) * wk.n + it];
    const DShape<T> a = lib.shapes[wk.shape1[pair]], b = lib.shapes[wk.shape2[pair]];
    const Pose<T> tf1 = load_pose(io.tf1, pair), tf2 = load_pose(io.tf2, pair);
    PairOut<T> o;
    o.distance = closed_form_distance(a, tf1, b, tf2, lib.verts, o.p1, o.p2, o.normal);
    o.gjk_status = GJK_DID_NOT_RUN;
    o.epa_status = EPA_DID_NOT_RUN;
    o.gjk_iters = o.epa_iters = 0;
    write_out<T>(io, q, pair, o);
    // the closed forms never touch the solver's cached guess: it stays at its initial value
    write_guess<T>(io, pair, initial_guess<T>(io, q, pair), 0, 0);
  }
}

// fp64 form with the poses and records staged through LDS: a lane's own 96-byte pose / record is six 16-byte
// pieces 96 bytes apart from its neighbour's, which the memory system only turns into full-line traffic through
// cache merging (non-temporal accesses: 2.5x slower, profiles/); here the block's 256 poses are fetched as
// 1536 consecutive 16-byte pieces (lane-contiguous when the bucket list is in input order, as it is up to the
// interleaving of blocks in k_classify), handed over in LDS, and the records leave the same way.
typedef double hfcl_d2 __attribute__((ext_vector_type(2)));
#ifndef HFCL_WPE_CLOSED
#define HFCL_WPE_CLOSED 2
#endif
__global__ void __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(HFCL_WPE_CLOSED, 8))) k_closed_staged(Work wk, LibView<double> lib, IO<double> io, QParams<double> q) {
  constexpr int NB = 256, PIECES = 6;  // 96 B = 6 x 16 B
  __shared__ uint32_t s_pair[NB];
  __shared__ hfcl_d2 s_a[NB * PIECES];  // poses of shape 1, later the records
  __shared__ hfcl_d2 s_b[NB * PIECES];  // poses of shape 2
  static_assert(sizeof(hfcl_result) == 96, "record = 6 pieces");
  const uint32_t cnt = wk.counts[B_CLOSED];
  const uint32_t t = threadIdx.x;
  for (uint32_t base = blockIdx.x * NB; base < cnt; base += gridDim.x * NB) {
    const uint32_t nvalid = min(uint32_t(NB), cnt - base);
    s_pair[t] = wk.lists[size_t(B_CLOSED) * wk.n + base + (t < nvalid ? t : 0u)];
    __syncthreads();
#pragma unroll
    for (int j = 0; j < PIECES; ++j) {
      const uint32_t c = t + NB * j, p = c / PIECES, part = c % PIECES;
      const size_t pr = s_pair[p];
      s_a[c] = reinterpret_cast<const hfcl_d2*>(io.tf1 + 12 * pr)[part];
      s_b[c] = reinterpret_cast<const hfcl_d2*>(io.tf2 + 12 * pr)[part];
    }
    const uint32_t pair = s_pair[t];
    const DShape<double> a = lib.shapes[wk.shape1[pair]], b = lib.shapes[wk.shape2[pair]];
    __syncthreads();
    const Pose<double> tf1 = pose_from_abi<double>(reinterpret_cast<const double*>(s_a + PIECES * t));
    const Pose<double> tf2 = pose_from_abi<double>(reinterpret_cast<const double*>(s_b + PIECES * t));
    PairOut<double> o;
    o.distance = closed_form_distance(a, tf1, b, tf2, lib.verts, o.p1, o.p2, o.normal);
    o.gjk_status = GJK_DID_NOT_RUN;
    o.epa_status = EPA_DID_NOT_RUN;
    o.gjk_iters = o.epa_iters = 0;
    __syncthreads();  // every pose has been read: s_a becomes the record buffer
    IO<double> lio = io;
    lio.out = reinterpret_cast<hfcl_result*>(s_a);
    write_out<double>(lio, q, t, o);
    if (t < nvalid) write_guess<double>(io, pair, initial_guess<double>(io, q, pair), 0, 0);
    __syncthreads();
#pragma unroll
    for (int j = 0; j < PIECES; ++j) {
      const uint32_t c = t + NB * j, p = c / PIECES, part = c % PIECES;
      if (p < nvalid) reinterpret_cast<hfcl_d2*>(io.out + s_pair[p])[part] = s_a[c];
    }
    __syncthreads();
  }
}

// ---------------------------------------------------------------------------------------
// Shared GJK epilogue: final record, or hand-off to k_epa through the device queue.
// ---------------------------------------------------------------------------------------
template <typename T>
__device__ __forceinline__ void finish_gjk(const Gjk<T, PW0<T>>& g, const Work& wk, const IO<T>& io, const QParams<T>& q,
                                           uint32_t pair, const Pose<T>& tf1, T r0, T r1, const V3<T>& guess0,
                                           bool writer, bool full_tier = false) {
  PairOut<T> o;
  EpaSeed<T> seed;
  const bool to_epa = gjk_finish(g, q, tf1, r0, r1, guess0, o, seed);
  if (!writer) return;
  if (to_epa) {
    // full_tier: straight to the full-capacity EPA queue (pairs with a large hull: only that tier
    // can scan vertices from memory)
    const uint32_t slot = atomicAdd(&wk.counts[full_tier ? B_COUNT + 1 : B_COUNT], 1u);
    seed.pair = pair;
    reinterpret_cast<EpaSeed<T>*>(full_tier ? wk.epa_queue2 : wk.epa_queue)[slot] = seed;
  } else {
    write_out<T>(io, q, pair, o);
    write_guess<T>(io, pair, o.cached_guess, 0, 0);
  }
}

// ---------------------------------------------------------------------------------------
// k_gjk_prim: primitive x primitive GJK, one pair per lane.
// ---------------------------------------------------------------------------------------
// BVG: GJKInitialGuess::BoundingVolumeGuess.  A separate instantiation on purpose: the register allocation of the GJK
// kernels is sensitive to anything live across their loop (the run-time form of this one select cost k_gjk_cvx<2,0>
// 0.96 -> 1.51 ms), so the default-guess kernels are compiled without it.
template <typename T, bool BVG>
__global__ void __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(HFCL_WPE_PRIM, 8))) k_gjk_prim(Work wk, LibView<T> lib, IO<T> io, QParams<T> q) {
  const uint32_t cnt = wk.counts[B_PRIM];
  for (uint32_t it = blockIdx.x * blockDim.x + threadIdx.x; it < cnt; it += gridDim.x * blockDim.x) {
    const uint32_t pair = wk.lists[size_t(B_PRIM) * wk.n + it];
    const DShape<T> a = lib.shapes[wk.shape1[pair]], b = lib.shapes[wk.shape2[pair]];
    const Pose<T> tf1 = load_pose(io.tf1, pair), tf2 = load_pose(io.tf2, pair);
    SerialSupport<T> sup;
    sup.a = a;
    sup.b = b;
    sup.va = sup.vb = nullptr;
    sup.md = make_mdiff(tf1, tf2);
    const T r0 = swept_radius(a), r1 = swept_radius(b);
    const V3<T> guess0 = initial_guess<T>(io, q, pair);
    Gjk<T, PW0<T>> g;
    if constexpr (BVG)
      gjk_run(g, q.gjk, start_guess(q, a, b, sup.md, guess0), r0 + r1, false, sup);
    else
      gjk_run(g, q.gjk, guess0, r0 + r1, false, sup);
    finish_gjk<T>(g, wk, io, q, pair, tf1, r0, r1, guess0, true);
  }
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
  __device__ __forceinline__ V3<T> support(const V3<T>& dir, int lig) const {
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
      if (od > best || (od == best && oi < bi)) {
        best = od;
        bi = oi;
      }
    });
    // the winner's coordinates come from a run-time lane (ds_bpermute): carrying them through the stages,
    // or OR-reducing the owner's bits, costs the GJK kernels registers they do not have (spills; measured)
    const int owner = bi / VPL, slot = bi % VPL;
    V3<T> c = v[0];
#pragma unroll
    for (int k = 1; k < VPL; ++k)
      if (slot == k) c = v[k];
    return mk<T>(__shfl(c.x, owner, W), __shfl(c.y, owner, W), __shfl(c.z, owner, W));
  }
};

// M: 0 = convex-convex, 1 = prim-convex, 2 = convex-prim
template <typename T, int W, int M>
struct CvxSupport {
  DShape<T> a, b;
  HullRegs<T, W> h0, h1;
  MDiff<T> md;
  int lig;
  __device__ __forceinline__ void eval(const V3<T>& dir, V3<T>& w, V3<T>& w0) const {
    if (M == 1)
      w0 = prim_support(a, dir);
    else
      w0 = h0.support(dir, lig);
    const V3<T> d1 = md.identity ? -dir : -tmul(md.oR1, dir);
    V3<T> s1;
    if (M == 2)
      s1 = prim_support(b, d1);
    else
      s1 = h1.support(d1, lig);
    s1 = md.identity ? s1 : (mul(md.oR1, s1) + md.ot1);
    w = w0 - s1;
  }
  __device__ __forceinline__ void operator()(const V3<T>& dir, V3<T>& w, V3<T>& w0) const { eval(dir, w, w0); }
};

template <typename T, int W, int M, bool BVG>
__device__ __forceinline__ void gjk_cvx_body(const Work& wk, const LibView<T>& lib, const IO<T>& io, const QParams<T>& q) {
  constexpr int BUCKET = (M == 0) ? B_CC : (M == 1 ? B_PC : B_CP);
  const uint32_t cnt = wk.counts[BUCKET];
  const int lig = threadIdx.x & (W - 1);
  const uint32_t groups = (gridDim.x * blockDim.x) / W;
  for (uint32_t it = (blockIdx.x * blockDim.x + threadIdx.x) / W; it < cnt; it += groups) {
    const uint32_t pair = wk.lists[size_t(BUCKET) * wk.n + it];
    CvxSupport<T, W, M> sup;
    sup.a = lib.shapes[wk.shape1[pair]];
    sup.b = lib.shapes[wk.shape2[pair]];
    sup.lig = lig;
    if (M != 1) sup.h0.load(lib.verts + 3 * size_t(sup.a.vertex_offset), sup.a.num_points, lig);
    if (M != 2) sup.h1.load(lib.verts + 3 * size_t(sup.b.vertex_offset), sup.b.num_points, lig);
    const Pose<T> tf1 = load_pose(io.tf1, pair), tf2 = load_pose(io.tf2, pair);
    sup.md = make_mdiff(tf1, tf2);
    const T r0 = swept_radius(sup.a), r1 = swept_radius(sup.b);
    const V3<T> guess0 = initial_guess<T>(io, q, pair);
    Gjk<T, PW0<T>> g;
    // normalize_support_direction only when both are ConvexBase (minkowski_difference.cpp:261-266)
    if constexpr (BVG)
      gjk_run(g, q.gjk, start_guess(q, sup.a, sup.b, sup.md, guess0), r0 + r1, M == 0, sup);
    else
      gjk_run(g, q.gjk, guess0, r0 + r1, M == 0, sup);
    finish_gjk<T>(g, wk, io, q, pair, tf1, r0, r1, guess0, lig == 0);
  }
}

// Two entry points so that each precision gets its own register budget (waves per SIMD): the fp64
// instantiation spills heavily at the fp32 setting (A/B in profiles/).
template <int W, int M, bool BVG>
__global__ void __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(W == 2 ? HFCL_WPE_GJK_W2 : HFCL_WPE_GJK, 8)))
k_gjk_cvx(Work wk, LibView<float> lib, IO<float> io, QParams<float> q) {
  gjk_cvx_body<float, W, M, BVG>(wk, lib, io, q);
}
template <int W, int M, bool BVG>
__global__ void __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(HFCL_WPE_GJK64, 8)))
k_gjk_cvx64(Work wk, LibView<double> lib, IO<double> io, QParams<double> q) {
  gjk_cvx_body<double, W, M, BVG>(wk, lib, io, q);
}
template <int W, int M, bool BVG = false>
static void launch_gjk_cvx(int grid, hipStream_t st, const Work& wk, const LibView<float>& lv, const IO<float>& io, const QParams<float>& q) {
  hipLaunchKernelGGL((k_gjk_cvx<W, M, BVG>), dim3(grid), dim3(256), 0, st, wk, lv, io, q);
}
template <int W, int M, bool BVG = false>
static void launch_gjk_cvx(int grid, hipStream_t st, const Work& wk, const LibView<double>& lv, const IO<double>& io, const QParams<double>& q) {
  hipLaunchKernelGGL((k_gjk_cvx64<W, M, BVG>), dim3(grid), dim3(256), 0, st, wk, lv, io, q);
}

// ---------------------------------------------------------------------------------------
// k_gjk_large: GJK for pairs with a hull of more than 32 vertices (either side; the other side may be
// any convex kind).  One pair per LW-lane group, vertices streamed from memory (L2-resident).
// ---------------------------------------------------------------------------------------
// Linear-scan support of a hull too large for registers: lane l of the W-lane group looks at vertices
// l, l+W, ... (coalesced), the group reduces to the first index of the maximum (the tie rule of
// getShapeSupportLinear; the reference's neighbour hill-climbing, support_functions.cpp:323-397, reaches
// a vertex of the same support value, possibly another one on a plateau -- see DESIGN.md).
template <typename T, int W>
__device__ __forceinline__ V3<T> scan_support(const T* v, uint32_t n, const V3<T>& dir, int lig) {
  T best = -Lim<T>::max();
  uint32_t bi = 0xFFFFFFFFu;
  for (uint32_t i = uint32_t(lig); i < n; i += W) {
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
    if (od > best || (od == best && oi < bi)) {
      best = od;
      bi = oi;
    }
  });
  return mk<T>(v[3 * bi], v[3 * bi + 1], v[3 * bi + 2]);
}

constexpr int LARGE_W = 16;
template <typename T>
struct LargeSupport {
  DShape<T> a, b;
  const T* va;
  const T* vb;
  MDiff<T> md;
  int lig;
  __device__ __forceinline__ V3<T> one(const DShape<T>& s, const T* v, const V3<T>& d) const {
    return s.kind == K_CONVEX ? scan_support<T, LARGE_W>(v, s.num_points, d, lig) : prim_support(s, d);
  }
  __device__ __forceinline__ void operator()(const V3<T>& dir, V3<T>& w, V3<T>& w0) const {
    w0 = one(a, va, dir);
    const V3<T> d1 = md.identity ? -dir : -tmul(md.oR1, dir);
    V3<T> s1 = one(b, vb, d1);
    s1 = md.identity ? s1 : (mul(md.oR1, s1) + md.ot1);
    w = w0 - s1;
  }
};

template <typename T, bool BVG>
__global__ void __launch_bounds__(256) k_gjk_large(Work wk, LibView<T> lib, IO<T> io, QParams<T> q) {
  const uint32_t cnt = wk.counts[B_LARGE];
  const int lig = threadIdx.x & (LARGE_W - 1);
  const uint32_t groups = (gridDim.x * blockDim.x) / LARGE_W;
  for (uint32_t it = (blockIdx.x * blockDim.x + threadIdx.x) / LARGE_W; it < cnt; it += groups) {
    const uint32_t pair = wk.lists[size_t(B_LARGE) * wk.n + it];
    LargeSupport<T> sup;
    sup.a = lib.shapes[wk.shape1[pair]];
    sup.b = lib.shapes[wk.shape2[pair]];
    sup.va = lib.verts + 3 * size_t(sup.a.vertex_offset);
    sup.vb = lib.verts + 3 * size_t(sup.b.vertex_offset);
    sup.lig = lig;
    const Pose<T> tf1 = load_pose(io.tf1, pair), tf2 = load_pose(io.tf2, pair);
    sup.md = make_mdiff(tf1, tf2);
    const T r0 = swept_radius(sup.a), r1 = swept_radius(sup.b);
    const V3<T> guess0 = initial_guess<T>(io, q, pair);
    Gjk<T, PW0<T>> g;
    if constexpr (BVG)
      gjk_run(g, q.gjk, start_guess(q, sup.a, sup.b, sup.md, guess0), r0 + r1, sup.a.kind == K_CONVEX && sup.b.kind == K_CONVEX, sup);
    else
      gjk_run(g, q.gjk, guess0, r0 + r1, sup.a.kind == K_CONVEX && sup.b.kind == K_CONVEX, sup);
    finish_gjk<T>(g, wk, io, q, pair, tf1, r0, r1, guess0, lig == 0, true);
  }
}

// ---------------------------------------------------------------------------------------
// k_epa: EPA on the pairs GJK left in `Collision`.  One polytope per WE-lane group, 64/WE polytopes
// per wavefront, scratch blocks in LDS.  Two tiers:
//   tier 1  WE = 8, CAP = 20 (fp32) / 24 (fp64) iterations: 8 polytopes per wave share one instruction
//           stream; a polytope that outgrows the small block is saved at the start of that iteration and queued
//   tier 2  WE = 16, CAP = 64 (the reference capacity): 4 polytopes per wave, continues the saved polytopes
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
};

#ifndef HFCL_EPA_FAST_CAP
#define HFCL_EPA_FAST_CAP 20
#endif
constexpr int EPA_FAST_CAP = HFCL_EPA_FAST_CAP;
#ifndef HFCL_EPA_FAST_CAP64
#define HFCL_EPA_FAST_CAP64 24  // cfg5 (fast + full ms): 12: 0.84+2.16, 16: 1.06+1.74, 20: 1.28+1.19, 24: 1.49+0.89; 28 would cost a wave per CU
#endif
// capacity of the fast tier's block per precision (fp64 blocks are twice the size; the LDS holds 4 waves x 8 either way)
template <typename T> constexpr int epa_fast_cap = sizeof(T) == 4 ? EPA_FAST_CAP : HFCL_EPA_FAST_CAP64;
#ifndef HFCL_EPA_WE
#define HFCL_EPA_WE 8
#endif
constexpr int EPA_WE = HFCL_EPA_WE;
#ifndef HFCL_EPA_WE2
#define HFCL_EPA_WE2 16
#endif
constexpr int EPA_WE2 = HFCL_EPA_WE2;  // lanes per polytope in the full-capacity tier

// LARGE: hulls of more than HULL_MAX vertices may occur (scanned from memory); only the
// full-capacity tier is built that way, so the fast tier keeps its register budget.
template <typename T, int WE, bool LARGE>
struct EpaSupport {  // any pair kind, evaluated by one lane group
  DShape<T> a, b;
  HullRegs<T, WE> h0, h1;
  const T* va;
  const T* vb;
  MDiff<T> md;
  int lig;
  __device__ __forceinline__ V3<T> hull(const DShape<T>& s, const HullRegs<T, WE>& h, const T* v, const V3<T>& d) const {
    if (LARGE && s.num_points > uint32_t(HULL_MAX)) return scan_support<T, WE>(v, s.num_points, d, lig);
    return h.support(d, lig);
  }
  __device__ __forceinline__ void operator()(const V3<T>& dir, V3<T>& w, V3<T>& w0) const {
    if (a.kind == K_CONVEX)
      w0 = hull(a, h0, va, dir);
    else
      w0 = prim_support(a, dir);
    const V3<T> d1 = md.identity ? -dir : -tmul(md.oR1, dir);
    V3<T> s1;
    if (b.kind == K_CONVEX)
      s1 = hull(b, h1, vb, d1);
    else
      s1 = prim_support(b, d1);
    s1 = md.identity ? s1 : (mul(md.oR1, s1) + md.ot1);
    w = w0 - s1;
  }
};

// TIER: 1 reads queue 1 and may push to queue 2; 2 reads queue 2 (never overflows: CAP = 64)
template <typename T, int WE, int CAP, int TIER>
__global__ void __launch_bounds__(64) __attribute__((amdgpu_waves_per_eu(sizeof(T) == 4 ? (TIER == 1 ? HFCL_WPE_EPA32 : 2) : HFCL_WPE_EPA64, 8)))
k_epa(Work wk, LibView<T> lib, IO<T> io, QParams<T> q) {
  constexpr int G = 64 / WE;
  // the full-capacity tier keeps the shape-0 support points in global memory: its LDS block bounds the
  // occupancy (fp64: 45.5 KB -> 3 waves per CU with them, 36.5 KB -> 4 without)
  constexpr bool V0IN = TIER == 1;
  __shared__ EpaScratch<T, CAP, V0IN> scratch[G];
  Quad<T>* const v0_ext = V0IN ? nullptr : reinterpret_cast<Quad<T>*>(wk.epa_v0) + size_t(blockIdx.x * G + threadIdx.x / WE) * (CAP + 4);
  const uint32_t cnt = wk.counts[TIER == 1 ? B_COUNT : B_COUNT + 1];
  const int lane = threadIdx.x & 63, grp = lane / WE, lig = lane & (WE - 1);
  const uint32_t groups = gridDim.x * G;
  const EpaItem<T>* queue = reinterpret_cast<const EpaItem<T>*>(TIER == 1 ? wk.epa_queue : wk.epa_queue2);
  for (uint32_t it = blockIdx.x * G + grp; it < cnt; it += groups) {
    // the seed stays in memory and is read where it is used (as a local copy it is spilled across the hull loads)
    const EpaItem<T>& item = queue[it];
    const uint32_t pair = item.pair;
    EpaSupport<T, WE, TIER == 2> sup;
    sup.a = lib.shapes[wk.shape1[pair]];
    sup.b = lib.shapes[wk.shape2[pair]];
    sup.lig = lig;
    const T* va = lib.verts + 3 * size_t(sup.a.vertex_offset);
    const T* vb = lib.verts + 3 * size_t(sup.b.vertex_offset);
    if (TIER == 2) {
      sup.va = va;
      sup.vb = vb;
    }
    if (sup.a.kind == K_CONVEX && (TIER != 2 || sup.a.num_points <= uint32_t(HULL_MAX))) sup.h0.load(va, sup.a.num_points, lig);
    if (sup.b.kind == K_CONVEX && (TIER != 2 || sup.b.num_points <= uint32_t(HULL_MAX))) sup.h1.load(vb, sup.b.num_points, lig);
    const Pose<T> tf1 = load_pose(io.tf1, pair), tf2 = load_pose(io.tf2, pair);
    sup.md = make_mdiff(tf1, tf2);
    const T r0 = swept_radius(sup.a), r1 = swept_radius(sup.b);
    PairOut<T> o;
    int rc = 1;
    if constexpr (TIER == 2) {
      if (item.rank & EPA_RESUME_FLAG) {  // continue what the fast tier saved for this slot (the seed's rank is not used)
        epa_resume<T, LaneGroup<WE>, epa_fast_cap<T>, CAP>(&scratch[grp], reinterpret_cast<const EpaScratch<T, epa_fast_cap<T>>*>(wk.epa_resume) + it,
                                                         item, q, tf1, r0, r1, sup, o, v0_ext);
      } else {
        rc = epa_run<T, LaneGroup<WE>, CAP>(&scratch[grp], item, q, tf1, r0, r1, sup, o, v0_ext);
      }
      if (lig == 0) {
        write_out<T>(io, q, pair, o);
        write_guess<T>(io, pair, o.cached_guess, 0, 0);
      }
    } else {
      rc = epa_run<T, LaneGroup<WE>, CAP>(&scratch[grp], item, q, tf1, r0, r1, sup, o);
      if (rc == 1) {
        if (lig == 0) {
          write_out<T>(io, q, pair, o);
          write_guess<T>(io, pair, o.cached_guess, 0, 0);
        }
      } else {  // hand over to the full-capacity tier: the seed, and the polytope itself when it can be continued
        uint32_t slot = 0;
        if (lig == 0) slot = atomicAdd(&wk.counts[B_COUNT + 1], 1u);
        slot = __shfl(slot, 0, WE);
        const bool save = rc == 2 && slot < wk.resume_cap;
        if (save) epa_save_block<T, LaneGroup<WE>, CAP>(&scratch[grp], reinterpret_cast<EpaScratch<T, CAP>*>(wk.epa_resume) + slot);
        if (lig == 0) {  // queue to queue, no local copy (a local EpaItem lives in scratch memory)
          EpaItem<T>* dst = reinterpret_cast<EpaItem<T>*>(wk.epa_queue2) + slot;
          *dst = item;
          if (save) dst->rank = item.rank | EPA_RESUME_FLAG;
        }
      }
    }
    LaneGroup<WE>::sync();
  }
}

// Fast tier as a stream: the 64/WE lane groups of a wave walk through the wave's share of queue 1 and a
// group that is done does not wait for the slowest polytope of the wave (EPA runs 1 .. CAP trips per
// polytope, mean ~7 on convex pairs: in lockstep batches of 8 only ~56 % of the trips are useful).
// The wave alternates between two uniform phases:
//   trip   : every group with a live polytope does one expansion step (Epa::step);
//   refill : once at least EPA_REFILL_MIN groups are without one (or none is live), those groups write
//            the record of the polytope they finished (or hand it over to the full-capacity tier) and
//            start the next item of the wave: seed, hulls, encloseOrigin, first tetrahedron (Epa::begin).
// Batching the refills matters: a refill costs about 1.5 trips of the whole wave whoever takes part.
#ifndef HFCL_EPA_REFILL_MIN
#define HFCL_EPA_REFILL_MIN 3
#endif
template <typename T, int WE, int CAP>
__global__ void __launch_bounds__(64) __attribute__((amdgpu_waves_per_eu(HFCL_WPE_EPA32, 8)))
k_epa_stream(Work wk, LibView<T> lib, IO<T> io, QParams<T> q) {
  constexpr int G = 64 / WE;
  typedef LaneGroup<WE> Grp;
  __shared__ EpaScratch<T, CAP> scratch[G];
  const uint32_t cnt = wk.counts[B_COUNT];
  const int lane = threadIdx.x & 63, grp = lane / WE, lig = lane & (WE - 1);
  const EpaItem<T>* queue = reinterpret_cast<const EpaItem<T>*>(wk.epa_queue);
  enum { IDLE = 0, LIVE = 1, DONE = 2, HANDOVER = 3 };
  int state = IDLE;
  uint32_t it = 0;              // queue slot of this group's polytope
  uint32_t next = blockIdx.x;   // wave-uniform: the wave's items are next, next + gridDim.x, ...
  EpaSupport<T, WE, false> sup;
  sup.lig = lig;
  Epa<T, Grp, CAP> epa;
  EpaLoop<T> L;
  Pose<T> tf1;
  T r0 = T(0), r1 = T(0);
  while (true) {
    const uint64_t live = __ballot(state == LIVE);
    const int n_live = __popcll(live) / WE;
    const bool more = next < cnt;
    if (n_live == 0 || (more && G - n_live >= HFCL_EPA_REFILL_MIN)) {
      // ---- refill phase (uniform decision; groups with a live polytope sit it out) ----
      if (state != LIVE) {
        if (state != IDLE) {
          if (state == DONE) {
            EpaResult<T> res;
            epa.loop_result(L, r0 + r1, res);
            PairOut<T> o;
            epa_finish(res, queue[it].gjk_iters, tf1, r0, r1, o);
            if (lig == 0) {
              const uint32_t pair = queue[it].pair;
              write_out<T>(io, q, pair, o);
              write_guess<T>(io, pair, o.cached_guess, 0, 0);
            }
          } else {  // hand over to the full-capacity tier: the seed and, room permitting, the polytope itself
            uint32_t slot = 0;
            if (lig == 0) slot = atomicAdd(&wk.counts[B_COUNT + 1], 1u);
            slot = __shfl(slot, 0, WE);
            const bool save = epa.resumable && slot < wk.resume_cap;
            if (save) epa_save_block<T, Grp, CAP>(&scratch[grp], reinterpret_cast<EpaScratch<T, CAP>*>(wk.epa_resume) + slot);
            if (lig == 0) {  // queue to queue, no local copy (a local EpaItem lives in scratch memory)
              EpaItem<T>* dst = reinterpret_cast<EpaItem<T>*>(wk.epa_queue2) + slot;
              *dst = queue[it];
              if (save) dst->rank = queue[it].rank | EPA_RESUME_FLAG;
            }
          }
          Grp::sync();
          state = IDLE;
        }
        // rank of this group among the groups taking part, in lane order
        const uint64_t lower = live | ~((uint64_t(1) << (grp * WE)) - 1);  // live lanes and lanes >= mine do not count
        const uint32_t rank = uint32_t(__popcll(~lower)) / WE;
        it = next + rank * gridDim.x;
        if (it < cnt) {
          // the seed is read field by field where it is used: as one struct it would sit in registers
          // across the hull loads and get spilled (1.6 KB of scratch traffic per polytope, measured)
          const EpaItem<T>* ip = queue + it;
          const uint32_t pair = ip->pair;
          sup.a = lib.shapes[wk.shape1[pair]];
          sup.b = lib.shapes[wk.shape2[pair]];
          if (sup.a.kind == K_CONVEX) sup.h0.load(lib.verts + 3 * size_t(sup.a.vertex_offset), sup.a.num_points, lig);
          if (sup.b.kind == K_CONVEX) sup.h1.load(lib.verts + 3 * size_t(sup.b.vertex_offset), sup.b.num_points, lig);
          tf1 = load_pose(io.tf1, pair);
          const Pose<T> tf2 = load_pose(io.tf2, pair);
          sup.md = make_mdiff(tf1, tf2);
          r0 = swept_radius(sup.a);
          r1 = swept_radius(sup.b);
          epa.reset(&scratch[grp], q.epa_max_iterations, q.epa_tolerance);
          epa.set_vert(0, ip->w[0], ip->w0[0]);
          epa.set_vert(1, ip->w[1], ip->w0[1]);
          epa.set_vert(2, ip->w[2], ip->w0[2]);
          epa.set_vert(3, ip->w[3], ip->w0[3]);
          Grp::sync();
          EpaResult<T> res;
          const int closest0 = epa.begin(ip->rank, -ip->guess, sup, res);
          if (closest0 != EPA_NULL) {
            epa.loop_enter(L, closest0, 0, 0);
            state = LIVE;
          } else if (epa.overflow) {
            state = HANDOVER;  // (a block too small for the first tetrahedron: not with CAP >= 1)
          } else {  // FallBack: final without a loop
            PairOut<T> o;
            epa_finish(res, ip->gjk_iters, tf1, r0, r1, o);
            if (lig == 0) {
              write_out<T>(io, q, pair, o);
              write_guess<T>(io, pair, o.cached_guess, 0, 0);
            }
          }
        }
      }
      next += uint32_t(G - n_live) * gridDim.x;
      if (n_live == 0 && !more) {
        // nothing was live and nothing was left to start: only a FallBack/empty refill can have happened
        if (__ballot(state == LIVE) == 0) break;
      }
      continue;
    }
    // ---- trip ----
    if (state == LIVE) {
      const int r = epa.step(L, sup);
      if (r != 0) state = r == 1 ? DONE : HANDOVER;
    }
  }
}

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
  const DRss<T>* rss;
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

constexpr int BVH_STACK = 96;
constexpr int BVH_BLOCK = 128;
#ifndef HFCL_BVH_REFILL_MIN
#define HFCL_BVH_REFILL_MIN 8
#endif
constexpr int BVH_REFILL_MIN = HFCL_BVH_REFILL_MIN;  // idle lanes of a wave that trigger a refill (4 / 8 / 16 / 24: 17.5 / 17.1 / 18.7 / 20.3 ms per 1M cfg4 queries)

template <typename T>
__global__ void __launch_bounds__(BVH_BLOCK) __attribute__((amdgpu_waves_per_eu(HFCL_WPE_BVH, 8))) k_bvh_collide(Work wk, LibView<T> lib, BvhView<T> bv, IO<T> io, QParams<T> q,
                                                          BvhParams bp, T break_distance2) {
  __shared__ uint32_t stack[BVH_STACK][BVH_BLOCK];
  const uint32_t cnt = wk.counts[B_BVH];
  uint32_t* const ticket = &wk.counts[B_COUNT + 2];
  const int tid = threadIdx.x, lane = tid & 63;
  const T nanv = Lim<T>::nan();
  // Per-lane query state.  Queries differ by an order of magnitude in length (cfg4: 135 BV tests on
  // average, 663 for the longest of 64), so the lanes of a wave do not advance through the batch in
  // lockstep: a lane whose traversal is over parks its result (`pending`) and, as soon as
  // BVH_REFILL_MIN lanes of the wave are idle, all of them write their records and take the next
  // queries from a global ticket counter.
  constexpr int refill_min = BVH_REFILL_MIN;
  bool live = false, pending = false, exhausted = false;  // exhausted is wave-uniform
  uint32_t pair = 0;
  DMesh m1 = {0, 0, 0, 0}, m2 = {0, 0, 0, 0};
  Pose<T> tf1, tf2;
  M3<T> RT_R;
  V3<T> RT_T;
  int sp = 0;
  bool overflow = false;
  uint32_t ncontacts = 0;
  T dlb = Lim<T>::max(), rec_dist = Lim<T>::max();
  V3<T> np1 = mk<T>(nanv, nanv, nanv), np2 = np1, nn = np1;
  int fb1 = -1, fb2 = -1;
  bool have_leaf = false;
  uint32_t lb1 = 0, lb2 = 0;
  auto flush = [&]() {  // record of the query this lane finished
    PairOut<T> o;
    o.distance = rec_dist;
    o.normal = nn;
    o.p1 = np1;
    o.p2 = np2;
    o.gjk_status = GJK_DID_NOT_RUN;
    o.epa_status = EPA_DID_NOT_RUN;
    o.gjk_iters = o.epa_iters = 0;
    store_bvh_record(io, pair, o, ncontacts, fb1, fb2, overflow);
  };
  for (;;) {
    if (live && !have_leaf && sp == 0) {  // traversal over
      live = false;
      pending = true;
    }
    const uint64_t live_mask = __ballot(live);
    const int n_live = __popcll(live_mask);
    if (exhausted ? n_live == 0 : 64 - n_live >= refill_min) {
      // ---- refill (wave-uniform decision; live lanes sit it out)
      if (pending) {
        flush();
        pending = false;
      }
      if (exhausted) break;
      const int n_need = 64 - n_live;
      uint32_t base = 0;
      if (lane == 0) base = atomicAdd(ticket, uint32_t(n_need));
      base = __builtin_amdgcn_readfirstlane(base);
      if (!live) {
        const uint32_t rank = uint32_t(__popcll(~live_mask & ((uint64_t(1) << lane) - 1)));
        const uint32_t it = base + rank;
        if (it < cnt) {
          pair = wk.lists[size_t(B_BVH) * wk.n + it];
          const DShape<T> a = lib.shapes[wk.shape1[pair]], b = lib.shapes[wk.shape2[pair]];
          m1 = bv.meshes[a.bvh_index];
          m2 = bv.meshes[b.bvh_index];
          tf1 = load_pose(io.tf1, pair);
          tf2 = load_pose(io.tf2, pair);
          RT_R = tmul(tf1.R, tf2.R);  // traversal_node_setup.h:560-563
          RT_T = tmul(tf1.R, tf2.t - tf1.t);
          stack[0][tid] = 0u;  // (b1 = 0, b2 = 0)
          sp = 1;
          overflow = false;
          ncontacts = 0;
          dlb = rec_dist = Lim<T>::max();
          np1 = np2 = nn = mk<T>(nanv, nanv, nanv);
          fb1 = fb2 = -1;
          have_leaf = false;
          live = true;
        }
      }
      if (base + uint32_t(n_need) >= cnt) exhausted = true;
      continue;
    }
    // ---- BV phase: advance every lane that has no leaf test pending, until half the wave waits for a
    // leaf test, nobody can advance, or enough lanes ran out of work to make a refill due
    for (;;) {
      const bool can_bv = live && !have_leaf && sp > 0;
      if (!__any(can_bv)) break;
      if (__popcll(__ballot(have_leaf)) >= 32) break;
      if (!exhausted && 64 - __popcll(__ballot(live && (have_leaf || sp > 0))) >= refill_min) break;
      if (can_bv) {
        const uint32_t e = stack[--sp][tid];
        const uint32_t b1 = e & 0xFFFFu, b2 = e >> 16;
        const DNode<T> n1 = bv.nodes[m1.node_off + b1];
        const DNode<T> n2 = bv.nodes[m2.node_off + b2];
        const bool l1 = n1.first_child < 0, l2 = n2.first_child < 0;
        if (l1 && l2) {
          have_leaf = true;
          lb1 = uint32_t(-(n1.first_child + 1));
          lb2 = uint32_t(-(n2.first_child + 1));
        } else {
          T sq;
          // argument order of the reference: overlap(RT.R, RT.T, model2.bv(b2), model1.bv(b1))
          const bool disjoint = obb_disjoint(RT_R, RT_T, n2, n1, q.security_margin, break_distance2, sq);
          if (disjoint) {  // updateDistanceLowerBoundFromBV
            if (!(dlb <= T(0))) {
              const T nd = hsqrt(sq);
              if (nd < dlb) {
                dlb = nd;
                rec_dist = nd + q.security_margin;
              }
            }
          } else {
            const T sz1 = sqnorm(n1.extent), sz2 = sqnorm(n2.extent);
            const bool first = l2 || (!l1 && (sz1 > sz2));  // firstOverSecond
            uint32_t ea, eb;
            if (first) {
              const uint32_t c1 = uint32_t(n1.first_child);
              ea = c1 | (b2 << 16);
              eb = (c1 + 1) | (b2 << 16);
            } else {
              const uint32_t c1 = uint32_t(n2.first_child);
              ea = b1 | (c1 << 16);
              eb = b1 | ((c1 + 1) << 16);
            }
            if (sp + 2 > BVH_STACK) {
              overflow = true;
              sp = 0;
            } else {
              stack[sp++][tid] = eb;  // second child below
              stack[sp++][tid] = ea;  // first child on top
            }
          }
        }
      }
    }
    // ---- leaf phase (leafCollides, traversal_node_bvhs.h:184-233)
    if (have_leaf) {
      have_leaf = false;
      const uint32_t* t1 = bv.tris + 3 * size_t(m1.tri_off + lb1);
      const uint32_t* t2 = bv.tris + 3 * size_t(m2.tri_off + lb2);
      const T* v1 = bv.verts + 3 * size_t(m1.vert_off);
      const T* v2 = bv.verts + 3 * size_t(m2.vert_off);
      auto vtx = [](const T* v, uint32_t i) { return mk<T>(v[3 * size_t(i)], v[3 * size_t(i) + 1], v[3 * size_t(i) + 2]); };
      TriSupport<T> tri;
      tri.p1 = xform(tf1, vtx(v1, t1[0]));
      tri.p2 = xform(tf1, vtx(v1, t1[1]));
      tri.p3 = xform(tf1, vtx(v1, t1[2]));
      tri.q1 = xform(tf2, vtx(v2, t2[0]));
      tri.q2 = xform(tf2, vtx(v2, t2[1]));
      tri.q3 = xform(tf2, vtx(v2, t2[2]));
      V3<T> p1, p2, n;
      int gst, git;
      const T distance = tri_tri_distance(tri, q.gjk, q.guess_mode == HFCL_GUESS_CACHED,
                                          mk<T>(q.guess[0], q.guess[1], q.guess[2]), p1, p2, n, gst, git);
      const T dtc = distance - q.security_margin;
      if (dtc < dlb) {  // updateDistanceLowerBoundFromLeaf
        dlb = dtc;
        rec_dist = distance;
        np1 = p1;
        np2 = p2;
        nn = n;
      }
      if (dtc <= q.collision_distance_threshold) {
        if (ncontacts < bp.num_max_contacts) {
          if (ncontacts == 0) {
            fb1 = int(lb1);
            fb2 = int(lb2);
          }
          ++ncontacts;
          if (bp.contacts) {
            const uint32_t slot = atomicAdd(bp.contacts_count, 1u);
            if (slot < bp.contacts_cap) {
              hfcl_contact c;
              c.pair = pair;
              c.b1 = int(lb1);
              c.b2 = int(lb2);
              c._pad = 0;
              c.penetration_depth = double(distance);
              c.normal[0] = n.x; c.normal[1] = n.y; c.normal[2] = n.z;
              c.p1[0] = p1.x; c.p1[1] = p1.y; c.p1[2] = p1.z;
              c.p2[0] = p2.x; c.p2[1] = p2.y; c.p2[2] = p2.z;
              bp.contacts[slot] = c;
            }
          }
        }
        if (ncontacts >= bp.num_max_contacts) sp = 0;  // canStop(): nothing else is visited
      }
    }
  }
}


// ---------------------------------------------------------------------------------------
// k_bvh_shape: BVHModel<OBBRSS> x convex solid collide(), either operand order.  One query per BS_W-lane
// group (hfcl_bvh_shape.hpp: sequential traversal, the group's lanes share support scans and EPA face
// work); DFS stack and the full-capacity polytope of each group in LDS.
// ---------------------------------------------------------------------------------------
constexpr int BS_W = 16;
constexpr int BS_STACK = 128;

template <typename T>
struct GroupSolid {  // support of the solid in its own frame, evaluated by the lane group
  DShape<T> s;
  HullRegs<T, BS_W> h;
  const T* v;
  int lig;
  __device__ __forceinline__ V3<T> operator()(const V3<T>& d) const {
    if (s.kind != K_CONVEX) return prim_support(s, d);
    if (s.num_points > uint32_t(HULL_MAX)) return scan_support<T, BS_W>(v, s.num_points, d, lig);
    return h.support(d, lig);
  }
};

template <typename T>
__global__ void __launch_bounds__(64) k_bvh_shape(Work wk, LibView<T> lib, BvhView<T> bv, IO<T> io, QParams<T> q, BvhParams bp,
                                                  T break_distance2) {
  constexpr int G = 64 / BS_W;
  __shared__ EpaScratch<T, EPA_MAX_ITER> scratch[G];
  __shared__ uint16_t stacks[G][BS_STACK];
  const uint32_t cnt = wk.counts[B_BVHSHAPE];
  const int lane = threadIdx.x & 63, grp = lane / BS_W, lig = lane & (BS_W - 1);
  for (uint32_t it = blockIdx.x * G + grp; it < cnt; it += gridDim.x * G) {
    const uint32_t pair = wk.lists[size_t(B_BVHSHAPE) * wk.n + it];
    const DShape<T> a = lib.shapes[wk.shape1[pair]], b = lib.shapes[wk.shape2[pair]];
    const bool swapped = a.kind != K_BVH;  // (shape, BVH): collide(o2, o1) then swapObjects (collision.cpp:93-108)
    const DShape<T> ms = swapped ? b : a;
    GroupSolid<T> solid;
    solid.s = swapped ? a : b;
    solid.v = lib.verts + 3 * size_t(solid.s.vertex_offset);
    solid.lig = lig;
    if (solid.s.kind == K_CONVEX && solid.s.num_points <= uint32_t(HULL_MAX)) solid.h.load(solid.v, solid.s.num_points, lig);
    const Pose<T> tf1 = load_pose(io.tf1, pair), tf2 = load_pose(io.tf2, pair);
    const Pose<T> tfm = swapped ? tf2 : tf1, tfs = swapped ? tf1 : tf2;
    const DMesh m = bv.meshes[ms.bvh_index];
    MeshShapeState<T> st;
    auto on_contact = [&](int prim, T distance, const V3<T>& p1, const V3<T>& p2, const V3<T>& nn) {
      if (lig != 0 || !bp.contacts) return;
      const uint32_t slot = atomicAdd(bp.contacts_count, 1u);
      if (slot >= bp.contacts_cap) return;
      hfcl_contact c;
      c.pair = pair;
      c.b1 = swapped ? -1 : prim;
      c.b2 = swapped ? prim : -1;
      c._pad = 0;
      c.penetration_depth = double(distance);
      const V3<T> a1 = swapped ? p2 : p1, a2 = swapped ? p1 : p2, an = swapped ? -nn : nn;
      c.normal[0] = an.x; c.normal[1] = an.y; c.normal[2] = an.z;
      c.p1[0] = a1.x; c.p1[1] = a1.y; c.p1[2] = a1.z;
      c.p2[0] = a2.x; c.p2[1] = a2.y; c.p2[2] = a2.z;
      bp.contacts[slot] = c;
    };
    mesh_shape_collide<T, LaneGroup<BS_W>>(bv.nodes + m.node_off, bv.verts + 3 * size_t(m.vert_off), bv.tris + 3 * size_t(m.tri_off),
                                           tfm, solid.s, lib.verts, tfs, solid, q, bp.num_max_contacts, break_distance2,
                                           stacks[grp], BS_STACK, &scratch[grp], initial_guess<T>(io, q, pair), on_contact, st);
    if (lig == 0) {
      if (st.unsupported) {
        auto r = io.out[pair];
        memset(&r, 0, sizeof(r));
        r.status = 0x80000000u;
        io.out[pair] = r;
      } else {
        PairOut<T> o;
        o.distance = st.rec_dist;
        o.normal = swapped ? -st.nn : st.nn;
        o.p1 = swapped ? st.np2 : st.np1;
        o.p2 = swapped ? st.np1 : st.np2;
        o.gjk_status = GJK_DID_NOT_RUN;
        o.epa_status = EPA_DID_NOT_RUN;
        o.gjk_iters = o.epa_iters = 0;
        store_bvh_record(io, pair, o, st.ncontacts, swapped ? -1 : st.first_prim, swapped ? st.first_prim : -1, st.overflow);
        write_guess<T>(io, pair, st.guess, 0, 0);
      }
    }
    LaneGroup<BS_W>::sync();
  }
}

// ---------------------------------------------------------------------------------------
// k_triangle: top-level TriangleP pairs (other than against Plane / Halfspace, which are closed forms):
// TriangleP x TriangleP (triangle_triangle.cpp:46-105), TriangleP x Sphere (triangle_sphere.cpp:45-68) and
// TriangleP x {Box, Capsule, Cone, Cylinder, Ellipsoid, ConvexBase} through GJKSolver::shapeDistance's
// TriangleP overloads (narrowphase.h:320-348).  One pair per BS_W-lane group, as the mesh x solid leaves.
// ---------------------------------------------------------------------------------------
template <typename T>
__global__ void __launch_bounds__(64) k_triangle(Work wk, LibView<T> lib, IO<T> io, QParams<T> q) {
  constexpr int G = 64 / BS_W;
  __shared__ EpaScratch<T, EPA_MAX_ITER> scratch[G];
  const uint32_t cnt = wk.counts[B_TRI];
  const int lane = threadIdx.x & 63, grp = lane / BS_W, lig = lane & (BS_W - 1);
  for (uint32_t it = blockIdx.x * G + grp; it < cnt; it += gridDim.x * G) {
    const uint32_t pair = wk.lists[size_t(B_TRI) * wk.n + it];
    const DShape<T> a = lib.shapes[wk.shape1[pair]], b = lib.shapes[wk.shape2[pair]];
    const Pose<T> tf1 = load_pose(io.tf1, pair), tf2 = load_pose(io.tf2, pair);
    const bool t1 = a.kind == K_TRIANGLE;
    GroupSolid<T> solid;  // the non-triangle shape (unused for TriangleP x TriangleP)
    solid.s = t1 ? b : a;
    solid.v = lib.verts + 3 * size_t(solid.s.vertex_offset);
    solid.lig = lig;
    if (solid.s.kind == K_CONVEX && solid.s.num_points <= uint32_t(HULL_MAX)) solid.h.load(solid.v, solid.s.num_points, lig);
    PairOut<T> o;
    triangle_pair<T, LaneGroup<BS_W>>(a, b, lib.verts, tf1, tf2, solid, q, initial_guess<T>(io, q, pair), &scratch[grp], o);
    if (lig == 0) {
      write_out<T>(io, q, pair, o);
      write_guess<T>(io, pair, o.cached_guess, 0, 0);
    }
    LaneGroup<BS_W>::sync();
  }
}

// distance() counterpart: same lane-group layout, RSS lower bounds instead of OBB overlap tests.
template <typename T>
__global__ void __launch_bounds__(64) k_bvh_shape_distance(Work wk, LibView<T> lib, BvhView<T> bv, IO<T> io, QParams<T> q) {
  constexpr int G = 64 / BS_W;
  __shared__ EpaScratch<T, EPA_MAX_ITER> scratch[G];
  __shared__ uint16_t stack_n[G][BS_STACK];
  __shared__ T stack_d[G][BS_STACK];
  const uint32_t cnt = wk.counts[B_BVHSHAPE];
  const int lane = threadIdx.x & 63, grp = lane / BS_W, lig = lane & (BS_W - 1);
  for (uint32_t it = blockIdx.x * G + grp; it < cnt; it += gridDim.x * G) {
    const uint32_t pair = wk.lists[size_t(B_BVHSHAPE) * wk.n + it];
    const DShape<T> a = lib.shapes[wk.shape1[pair]], b = lib.shapes[wk.shape2[pair]];
    const bool swapped = a.kind != K_BVH;  // distance.cpp:74-88
    const DShape<T> ms = swapped ? b : a;
    GroupSolid<T> solid;
    solid.s = swapped ? a : b;
    solid.v = lib.verts + 3 * size_t(solid.s.vertex_offset);
    solid.lig = lig;
    if (solid.s.kind == K_CONVEX && solid.s.num_points <= uint32_t(HULL_MAX)) solid.h.load(solid.v, solid.s.num_points, lig);
    const Pose<T> tf1 = load_pose(io.tf1, pair), tf2 = load_pose(io.tf2, pair);
    const Pose<T> tfm = swapped ? tf2 : tf1, tfs = swapped ? tf1 : tf2;
    const DMesh m = bv.meshes[ms.bvh_index];
    MeshShapeDist<T> st;
    mesh_shape_distance<T, LaneGroup<BS_W>>(bv.nodes + m.node_off, bv.rss + m.node_off, bv.verts + 3 * size_t(m.vert_off),
                                            bv.tris + 3 * size_t(m.tri_off), tfm, solid.s, lib.verts, tfs, solid, q, stack_n[grp],
                                            stack_d[grp], BS_STACK, &scratch[grp], initial_guess<T>(io, q, pair), st);
    if (lig == 0) {
      if (st.unsupported) {
        auto r = io.out[pair];
        memset(&r, 0, sizeof(r));
        r.status = 0x80000000u;
        io.out[pair] = r;
      } else {
        PairOut<T> o;
        o.distance = st.min_distance;
        o.normal = swapped ? -st.nn : st.nn;
        o.p1 = swapped ? st.np2 : st.np1;
        o.p2 = swapped ? st.np1 : st.np2;
        o.gjk_status = GJK_DID_NOT_RUN;
        o.epa_status = EPA_DID_NOT_RUN;
        o.gjk_iters = o.epa_iters = 0;
        // b1 = the triangle, b2 = NONE whatever the operand order (distance.cpp:84-88 swaps o1/o2 only)
        store_bvh_record(io, pair, o, st.min_distance <= T(0) ? 0x80000000u : 0u, st.prim, -1, st.overflow);
        write_guess<T>(io, pair, st.guess, 0, 0);
      }
    }
    LaneGroup<BS_W>::sync();
  }
}

// ---------------------------------------------------------------------------------------
// k_bvh_distance: BVHModel<OBBRSS> x BVHModel<OBBRSS> distance().  distanceRecurse
// (src/traversal/traversal_recurse.cpp:153-203) flattened: both child pairs get their RSS lower
// bound, the farther one is pushed first (with its bound), the nearer one on top; a popped entry is
// skipped when its bound can no longer beat the current minimum (canStop, rel_err = abs_err = 0 as
// latched by the reference's traversal node, traversal_node_bvhs.h:409-410).  Leaves =
// sqrTriDistance in model 1's frame; the result is seeded with triangle 0 x triangle 0 (preprocess).
// ---------------------------------------------------------------------------------------
constexpr int BVHD_STACK = 64;
constexpr int BVHD_BLOCK = 64;

template <typename T>
__global__ void __launch_bounds__(BVHD_BLOCK) __attribute__((amdgpu_waves_per_eu(HFCL_WPE_BVH, 8))) k_bvh_distance(Work wk, LibView<T> lib, BvhView<T> bv, IO<T> io, QParams<T> q) {
  __shared__ uint32_t stack_e[BVHD_STACK][BVHD_BLOCK];
  __shared__ T stack_d[BVHD_STACK][BVHD_BLOCK];
  const uint32_t cnt = wk.counts[B_BVH];
  uint32_t* const ticket = &wk.counts[B_COUNT + 2];
  const int tid = threadIdx.x, lane = tid & 63;
  const T nanv = Lim<T>::nan();
  // streaming as in k_bvh_collide: per-lane query state, refill once BVH_REFILL_MIN lanes are idle
  bool live = false, pending = false, exhausted = false;  // exhausted is wave-uniform
  uint32_t pair = 0;
  DMesh m1 = {0, 0, 0, 0}, m2 = {0, 0, 0, 0};
  Pose<T> tf1;
  M3<T> RT_R;
  V3<T> RT_T;
  T mind = Lim<T>::max();
  int fb1 = -1, fb2 = -1;
  V3<T> np1 = mk<T>(nanv, nanv, nanv), np2 = np1;
  bool overflow = false;
  int sp = 0;
  auto vtx = [](const T* v, uint32_t i) { return mk<T>(v[3 * size_t(i)], v[3 * size_t(i) + 1], v[3 * size_t(i) + 2]); };
  auto leaf = [&](uint32_t p1i, uint32_t p2i) {
    const T* v1 = bv.verts + 3 * size_t(m1.vert_off);
    const T* v2 = bv.verts + 3 * size_t(m2.vert_off);
    const uint32_t* t1 = bv.tris + 3 * size_t(m1.tri_off + p1i);
    const uint32_t* t2 = bv.tris + 3 * size_t(m2.tri_off + p2i);
    V3<T> P, Q;
    const T d2 = sqr_tri_distance(vtx(v1, t1[0]), vtx(v1, t1[1]), vtx(v1, t1[2]), mul(RT_R, vtx(v2, t2[0])) + RT_T,
                                  mul(RT_R, vtx(v2, t2[1])) + RT_T, mul(RT_R, vtx(v2, t2[2])) + RT_T, P, Q);
    const T d = hsqrt(d2);
    if (mind > d) {  // DistanceResult::update
      mind = d;
      fb1 = int(p1i);
      fb2 = int(p2i);
      np1 = P;
      np2 = Q;
    }
  };
  for (;;) {
    if (live && sp == 0) {
      live = false;
      pending = true;
    }
    const uint64_t live_mask = __ballot(live);
    const int n_live = __popcll(live_mask);
    if (exhausted ? n_live == 0 : 64 - n_live >= BVH_REFILL_MIN) {
      if (pending) {
        PairOut<T> o;
        o.distance = mind;
        o.normal = mk<T>(nanv, nanv, nanv);  // not set by the reference on this path (traversal_node_bvhs.h:454,465)
        o.p1 = xform(tf1, np1);              // postprocess(): model-1 frame -> world
        o.p2 = xform(tf1, np2);
        o.gjk_status = GJK_DID_NOT_RUN;
        o.epa_status = EPA_DID_NOT_RUN;
        o.gjk_iters = o.epa_iters = 0;
        store_bvh_record(io, pair, o, mind <= T(0) ? 0x80000000u : 0u, fb1, fb2, overflow);
        pending = false;
      }
      if (exhausted) break;
      const int n_need = 64 - n_live;
      uint32_t base = 0;
      if (lane == 0) base = atomicAdd(ticket, uint32_t(n_need));
      base = __builtin_amdgcn_readfirstlane(base);
      if (!live) {
        const uint32_t it = base + uint32_t(__popcll(~live_mask & ((uint64_t(1) << lane) - 1)));
        if (it < cnt) {
          pair = wk.lists[size_t(B_BVH) * wk.n + it];
          const DShape<T> a = lib.shapes[wk.shape1[pair]], b = lib.shapes[wk.shape2[pair]];
          m1 = bv.meshes[a.bvh_index];
          m2 = bv.meshes[b.bvh_index];
          tf1 = load_pose(io.tf1, pair);
          const Pose<T> tf2 = load_pose(io.tf2, pair);
          RT_R = tmul(tf1.R, tf2.R);
          RT_T = tmul(tf1.R, tf2.t - tf1.t);
          mind = Lim<T>::max();
          fb1 = fb2 = -1;
          np1 = np2 = mk<T>(nanv, nanv, nanv);
          overflow = false;
          leaf(0u, 0u);  // preprocess()
          sp = 1;
          stack_e[0][tid] = 0u;
          stack_d[0][tid] = T(-1);
          live = true;
        }
      }
      if (base + uint32_t(n_need) >= cnt) exhausted = true;
      continue;
    }
    for (;;) {
      const bool run = live && sp > 0;
      const int n_run = __popcll(__ballot(run));
      if (n_run == 0 || (!exhausted && 64 - n_run >= BVH_REFILL_MIN)) break;
      if (!run) continue;
      --sp;
      const uint32_t e = stack_e[sp][tid];
      const T de = stack_d[sp][tid];
      if (de >= T(0) && de >= mind) continue;  // canStop(d)
      const uint32_t b1 = e & 0xFFFFu, b2 = e >> 16;
      const DNode<T> n1 = bv.nodes[m1.node_off + b1];
      const DNode<T> n2 = bv.nodes[m2.node_off + b2];
      const bool l1 = n1.first_child < 0, l2 = n2.first_child < 0;
      if (l1 && l2) {
        leaf(uint32_t(-(n1.first_child + 1)), uint32_t(-(n2.first_child + 1)));
        continue;
      }
      uint32_t a1, a2, c1, c2;
      if (l2 || (!l1 && (sqnorm(n1.extent) > sqnorm(n2.extent)))) {
        a1 = uint32_t(n1.first_child);
        a2 = b2;
        c1 = a1 + 1;
        c2 = b2;
      } else {
        a1 = b1;
        a2 = uint32_t(n2.first_child);
        c1 = b1;
        c2 = a2 + 1;
      }
      const T d1 = rss_lower_bound(RT_R, RT_T, bv.nodes[m1.node_off + a1], bv.rss[m1.node_off + a1],
                                   bv.nodes[m2.node_off + a2], bv.rss[m2.node_off + a2]);
      const T d2 = rss_lower_bound(RT_R, RT_T, bv.nodes[m1.node_off + c1], bv.rss[m1.node_off + c1],
                                   bv.nodes[m2.node_off + c2], bv.rss[m2.node_off + c2]);
      if (sp + 2 > BVHD_STACK) {
        overflow = true;
        sp = 0;
        continue;
      }
      const uint32_t ea = a1 | (a2 << 16), ec = c1 | (c2 << 16);
      const bool c_first = d2 < d1;  // visit (c1,c2) first when it is strictly nearer
      stack_e[sp][tid] = c_first ? ea : ec;
      stack_d[sp][tid] = c_first ? d1 : d2;
      ++sp;
      stack_e[sp][tid] = c_first ? ec : ea;
      stack_d[sp][tid] = c_first ? d2 : d1;
      ++sp;
    }
  }
}

// =======================================================================================
// Host side: library object + C ABI
// =======================================================================================
static thread_local std::string g_last_error;
static void set_error(const std::string& s) { g_last_error = s; }

#define HIP_TRY(expr)                                                                         \
  do {                                                                                        \
    hipError_t _e = (expr);                                                                   \
    if (_e != hipSuccess) {                                                                   \
      set_error(std::string(#expr) + ": " + hipGetErrorString(_e));                           \
      return HFCL_ERR_HIP;                                                                    \
    }                                                                                         \
  } while (0)

struct KernelTime {
  const char* name;
  hipEvent_t e0, e1;
  bool used;
};

struct hfcl_lib {
  int device = 0;
  size_t n_shapes = 0;
  std::vector<hfcl_shape> h_shapes;
  DShape<double>* d_shapes64 = nullptr;
  DShape<float>* d_shapes32 = nullptr;
  double* d_verts64 = nullptr;
  float* d_verts32 = nullptr;
  uint8_t* d_kinds = nullptr;
  // workspace (grown on demand)
  size_t ws_capacity = 0;  // pairs
  uint32_t* d_lists = nullptr;
  uint32_t* d_counts = nullptr;
  void* d_epa_queue = nullptr;
  void* d_epa_queue2 = nullptr;
  void* d_epa_resume = nullptr;
  void* d_epa_v0 = nullptr;
  size_t resume_cap = 0;
  // host-call staging buffers
  size_t st_capacity = 0;
  uint32_t *d_s1 = nullptr, *d_s2 = nullptr;
  double *d_tf1 = nullptr, *d_tf2 = nullptr;
  hfcl_result* d_out = nullptr;
  hfcl_guess *d_gin = nullptr, *d_gout = nullptr;
  // instrumentation
  std::vector<KernelTime> timers;
  // A batch can run as two halves on two streams (hfcl_lib_set_split): the second half goes to `helper`, a shallow
  // clone (same device shape tables, own workspace / counters / timers) on the internal stream `side`, whose kernels
  // fill the drain phases of the first half's GJK / EPA launches (profiles/r01_k_two_stream_overlap.txt).
  int split = 0;  // 0 = automatic (auto_split), 1 = never, 2 = always (large batches without meshes)
  hfcl_lib* helper = nullptr;
  bool is_helper = false;   // does not own the shape tables
  bool last_split = false;  // the last batch ran split: counters / timers of the helper belong to it
  hipStream_t side = nullptr;
  hipEvent_t ev_fork = nullptr, ev_join = nullptr;
  bool kernel_timing = true;         // HIP events around every kernel (hfcl_lib_set_kernel_timing)
  uint32_t possible_buckets = ~0u;   // bit b: some pair of this library's shape kinds classifies into bucket b
  int cvx_w = 0;  // 0 = per kernel (auto_cvx_w); HFCL_CVX_W forces one width for all
  bool closed_staged = true;  // HFCL_CLOSED_STAGED=0: A/B switch back to the direct-access k_closed<double>
  int n_cus = 256;
  std::string dominant;
  // bucket populations of the last call; PINNED host memory so that the device-to-host copy at the end of a batch is
  // asynchronous (a pageable destination makes hipMemcpyAsync block the host until the whole batch has run)
  uint32_t* h_counts = nullptr;
  // BVH models (host staging + device images in both precisions; uploaded lazily)
  std::vector<hfcl_bvh_node> h_bvh_nodes;
  std::vector<double> h_bvh_verts;
  std::vector<uint32_t> h_bvh_tris;
  std::vector<DMesh> h_meshes;
  bool bvh_dirty = false;
  DNode<double>* d_nodes64 = nullptr;
  DNode<float>* d_nodes32 = nullptr;
  DRss<double>* d_rss64 = nullptr;
  DRss<float>* d_rss32 = nullptr;
  double* d_bverts64 = nullptr;
  float* d_bverts32 = nullptr;
  uint32_t* d_btris = nullptr;
  DMesh* d_meshes = nullptr;
  // contact list of the last hfcl_collide_batch_contacts call
  hfcl_contact* d_contacts = nullptr;
  size_t contacts_cap = 0;
  uint32_t* d_contacts_count = nullptr;
  BvhParams bvh_params = {1u, nullptr, 0u, nullptr};
  double break_distance = 1e-3;
};

// bucket population i of the last batch (both halves of a split batch)
static uint32_t total_count(const hfcl_lib* lib, int i) {
  uint32_t c = lib->h_counts ? lib->h_counts[i] : 0u;
  if (lib->last_split && lib->helper && lib->helper->h_counts) c += lib->helper->h_counts[i];
  return c;
}

static int ensure_device(int device) {
  int n = 0;
  hipError_t e = hipGetDeviceCount(&n);
  if (e != hipSuccess || n == 0) {
    set_error("no HIP device available (hipGetDeviceCount): the engine has no CPU fallback");
    return HFCL_ERR_NO_DEVICE;
  }
  if (device < 0 || device >= n) {
    set_error("device index out of range");
    return HFCL_ERR_INVALID_ARGUMENT;
  }
  HIP_TRY(hipSetDevice(device));
  return HFCL_OK;
}

extern "C" {

int hfcl_abi_version(void) { return HFCL_ABI_VERSION; }
int hfcl_pair_supported(int32_t t1, int32_t t2, int for_distance) {
  if (t1 < 0 || t1 > 255 || t2 < 0 || t2 > 255) return 0;
  return bucket_of(t1, t2, for_distance != 0) != B_UNSUPPORTED;
}
int hfcl_device_count(void) {
  int n = 0;
  if (hipGetDeviceCount(&n) != hipSuccess) return 0;
  return n;
}
const char* hfcl_last_error(void) { return g_last_error.c_str(); }

static void query_defaults(hfcl_query_request* q) {
  q->gjk_initial_guess = HFCL_GUESS_DEFAULT;
  q->gjk_variant = HFCL_GJK_DEFAULT;
  q->gjk_convergence_criterion = HFCL_CRIT_DEFAULT;
  q->gjk_convergence_criterion_type = HFCL_CRIT_RELATIVE;
  q->gjk_max_iterations = 128;
  q->epa_max_iterations = 64;
  q->gjk_tolerance = 1e-6;
  q->epa_tolerance = 1e-6;
  q->collision_distance_threshold = 1e-12;
  q->cached_gjk_guess[0] = 1.0;
  q->cached_gjk_guess[1] = 0.0;
  q->cached_gjk_guess[2] = 0.0;
  q->cached_support_func_guess[0] = 0;
  q->cached_support_func_guess[1] = 0;
}
void hfcl_collision_request_init(hfcl_collision_request* r) {
  memset(r, 0, sizeof(*r));
  query_defaults(&r->q);
  r->num_max_contacts = 1;
  r->enable_contact = 1;
  r->security_margin = 0.0;
  r->break_distance = 1e-3;
  r->distance_upper_bound = 1.7976931348623157e+308;
}
void hfcl_distance_request_init(hfcl_distance_request* r) {
  memset(r, 0, sizeof(*r));
  query_defaults(&r->q);
  r->enable_nearest_points = 1;
  r->enable_signed_distance = 1;
  r->rel_err = 0.0;
  r->abs_err = 0.0;
}

hfcl_lib* hfcl_lib_create(const hfcl_shape* shapes, size_t n_shapes, const double* vertices, size_t n_vertices,
                          int device) {
  if (ensure_device(device) != HFCL_OK) return nullptr;
  if (!shapes || n_shapes == 0) {
    set_error("hfcl_lib_create: empty shape table");
    return nullptr;
  }
  for (size_t i = 0; i < n_shapes; ++i) {
    const hfcl_shape& s = shapes[i];
    const bool ok_kind = s.type == HFCL_GEOM_BOX || s.type == HFCL_GEOM_SPHERE || s.type == HFCL_GEOM_CAPSULE ||
                         s.type == HFCL_GEOM_ELLIPSOID || s.type == HFCL_GEOM_CONVEX || s.type == HFCL_BV_OBBRSS ||
                         s.type == HFCL_GEOM_TRIANGLE || s.type == HFCL_GEOM_CONE || s.type == HFCL_GEOM_CYLINDER ||
                         s.type == HFCL_GEOM_PLANE || s.type == HFCL_GEOM_HALFSPACE;
    if (!ok_kind) {
      set_error("hfcl_lib_create: unsupported shape type " + std::to_string(s.type));
      return nullptr;
    }
    if (s.type == HFCL_GEOM_CONVEX) {
      if (s.num_points == 0 || s.num_points > (uint32_t)HULL_LARGE_MAX) {
        set_error("hfcl_lib_create: convex shapes must have 1.." + std::to_string(HULL_LARGE_MAX) + " vertices; got " +
                  std::to_string(s.num_points));
        return nullptr;
      }
      if (size_t(s.vertex_offset) + s.num_points > n_vertices) {
        set_error("hfcl_lib_create: convex vertex range out of bounds");
        return nullptr;
      }
    }
  }
  hfcl_lib* lib = new hfcl_lib();
  lib->device = device;
  lib->n_shapes = n_shapes;
  lib->h_shapes.assign(shapes, shapes + n_shapes);
  std::vector<DShape<double>> s64(n_shapes);
  std::vector<DShape<float>> s32(n_shapes);
  std::vector<uint8_t> kinds(n_shapes);
  for (size_t i = 0; i < n_shapes; ++i) {
    const hfcl_shape& s = shapes[i];
    s64[i].kind = s.type;
    s64[i].num_points = s.num_points;
    s64[i].vertex_offset = s.vertex_offset;
    s64[i].bvh_index = s.bvh_index;
    s64[i].p0 = s.params[0];
    s64[i].p1 = s.params[1];
    s64[i].p2 = s.params[2];
    s64[i].p3 = s.params[3];
    s64[i].ssr = s.swept_sphere_radius;
    s32[i].kind = s.type;
    s32[i].num_points = s.num_points;
    s32[i].vertex_offset = s.vertex_offset;
    s32[i].bvh_index = s.bvh_index;
    s32[i].p0 = float(s.params[0]);
    s32[i].p1 = float(s.params[1]);
    s32[i].p2 = float(s.params[2]);
    s32[i].p3 = float(s.params[3]);
    s32[i].ssr = float(s.swept_sphere_radius);
    if (s.type == HFCL_GEOM_CONVEX && s.num_points > 0 && vertices &&
        size_t(s.vertex_offset) + s.num_points <= n_vertices) {  // centre of aabb_local, for BoundingVolumeGuess
      double mn[3], mx[3];
      const double* v = vertices + 3 * size_t(s.vertex_offset);
      for (int k = 0; k < 3; ++k) mn[k] = mx[k] = v[k];
      for (uint32_t j = 1; j < s.num_points; ++j)
        for (int k = 0; k < 3; ++k) {
          mn[k] = std::min(mn[k], v[3 * size_t(j) + k]);
          mx[k] = std::max(mx[k], v[3 * size_t(j) + k]);
        }
      s64[i].p0 = (mn[0] + mx[0]) * 0.5; s64[i].p1 = (mn[1] + mx[1]) * 0.5; s64[i].p2 = (mn[2] + mx[2]) * 0.5;
      s32[i].p0 = float(s64[i].p0); s32[i].p1 = float(s64[i].p1); s32[i].p2 = float(s64[i].p2);
    }
    kinds[i] = uint8_t(s.type == HFCL_GEOM_CONVEX && s.num_points > (uint32_t)HULL_MAX ? K_CONVEX_LARGE : s.type);
  }
  {
    bool present[256] = {false};
    for (size_t i = 0; i < n_shapes; ++i) present[kinds[i]] = true;
    uint32_t mask = 1u << B_UNSUPPORTED;  // shape ids out of range can always occur
    for (int a = 0; a < 256; ++a)
      if (present[a])
        for (int b = 0; b < 256; ++b)
          if (present[b]) mask |= 1u << bucket_of(a, b);
    lib->possible_buckets = mask;
  }
  std::vector<float> v32(3 * n_vertices + 3);
  for (size_t i = 0; i < 3 * n_vertices; ++i) v32[i] = float(vertices[i]);
  bool ok = true;
  ok = ok && hipMalloc(&lib->d_shapes64, n_shapes * sizeof(DShape<double>)) == hipSuccess;
  ok = ok && hipMalloc(&lib->d_shapes32, n_shapes * sizeof(DShape<float>)) == hipSuccess;
  ok = ok && hipMalloc(&lib->d_kinds, n_shapes) == hipSuccess;
  ok = ok && hipMalloc(&lib->d_verts64, (3 * n_vertices + 3) * sizeof(double)) == hipSuccess;
  ok = ok && hipMalloc(&lib->d_verts32, (3 * n_vertices + 3) * sizeof(float)) == hipSuccess;
  ok = ok && hipMalloc(&lib->d_counts, (B_COUNT + 3) * sizeof(uint32_t)) == hipSuccess;
  ok = ok && hipHostMalloc((void**)&lib->h_counts, (B_COUNT + 2) * sizeof(uint32_t), hipHostMallocDefault) == hipSuccess;
  if (ok) memset(lib->h_counts, 0, (B_COUNT + 2) * sizeof(uint32_t));
  if (ok) {
    ok = ok && hipMemcpy(lib->d_shapes64, s64.data(), n_shapes * sizeof(DShape<double>), hipMemcpyHostToDevice) == hipSuccess;
    ok = ok && hipMemcpy(lib->d_shapes32, s32.data(), n_shapes * sizeof(DShape<float>), hipMemcpyHostToDevice) == hipSuccess;
    ok = ok && hipMemcpy(lib->d_kinds, kinds.data(), n_shapes, hipMemcpyHostToDevice) == hipSuccess;
    if (n_vertices) {
      ok = ok && hipMemcpy(lib->d_verts64, vertices, 3 * n_vertices * sizeof(double), hipMemcpyHostToDevice) == hipSuccess;
      ok = ok && hipMemcpy(lib->d_verts32, v32.data(), 3 * n_vertices * sizeof(float), hipMemcpyHostToDevice) == hipSuccess;
    }
  }
  if (!ok) {
    set_error("hfcl_lib_create: HIP allocation/copy failed");
    hfcl_lib_destroy(lib);
    return nullptr;
  }
  hipDeviceProp_t prop;
  if (hipGetDeviceProperties(&prop, device) == hipSuccess) lib->n_cus = prop.multiProcessorCount;
  // one slot per lane group of the largest full-capacity EPA grid (run_batch caps grids at n_cus * 16 blocks)
  if (hipMalloc(&lib->d_epa_v0, size_t(lib->n_cus) * 16 * (64 / EPA_WE2) * EPA_MAX_VERTS * sizeof(Quad<double>)) != hipSuccess) {
    set_error("hfcl_lib_create: HIP allocation failed");
    hfcl_lib_destroy(lib);
    return nullptr;
  }
  if (const char* v = getenv("HFCL_CLOSED_STAGED")) lib->closed_staged = atoi(v) != 0;
  if (const char* v = getenv("HFCL_SPLIT")) lib->split = atoi(v) >= 2 ? 2 : (atoi(v) == 1 ? 1 : 0);
  if (const char* w = getenv("HFCL_CVX_W")) {
    int v = atoi(w);
    if (v == 2 || v == 4 || v == 8 || v == 16 || v == 32 || v == 64) lib->cvx_w = v;
  }
  return lib;
}

void hfcl_lib_destroy(hfcl_lib* lib) {
  if (!lib) return;
  hipSetDevice(lib->device);
  hipDeviceSynchronize();
  if (lib->helper) hfcl_lib_destroy(lib->helper);
  if (lib->side) hipStreamDestroy(lib->side);
  if (lib->ev_fork) hipEventDestroy(lib->ev_fork);
  if (lib->ev_join) hipEventDestroy(lib->ev_join);
  if (!lib->is_helper) {
    hipFree(lib->d_shapes64);
    hipFree(lib->d_shapes32);
    hipFree(lib->d_kinds);
    hipFree(lib->d_verts64);
    hipFree(lib->d_verts32);
  }
  hipFree(lib->d_counts);
  if (lib->h_counts) hipHostFree(lib->h_counts);
  hipFree(lib->d_lists);
  hipFree(lib->d_epa_queue);
  hipFree(lib->d_epa_queue2);
  hipFree(lib->d_epa_resume);
  hipFree(lib->d_epa_v0);
  hipFree(lib->d_s1);
  hipFree(lib->d_s2);
  hipFree(lib->d_tf1);
  hipFree(lib->d_tf2);
  hipFree(lib->d_out);
  hipFree(lib->d_gin);
  hipFree(lib->d_gout);
  hipFree(lib->d_nodes64);
  hipFree(lib->d_nodes32);
  hipFree(lib->d_rss64);
  hipFree(lib->d_rss32);
  hipFree(lib->d_bverts64);
  hipFree(lib->d_bverts32);
  hipFree(lib->d_btris);
  hipFree(lib->d_meshes);
  hipFree(lib->d_contacts);
  hipFree(lib->d_contacts_count);
  for (auto& t : lib->timers) {
    hipEventDestroy(t.e0);
    hipEventDestroy(t.e1);
  }
  delete lib;
}
size_t hfcl_lib_num_shapes(const hfcl_lib* lib) { return lib ? lib->n_shapes : 0; }
int hfcl_lib_device(const hfcl_lib* lib) { return lib ? lib->device : -1; }

int hfcl_lib_add_bvh(hfcl_lib* lib, const hfcl_bvh_node* nodes, size_t n_nodes, const double* vertices,
                     size_t n_vertices, const uint32_t* triangles, size_t n_tris) {
  if (!lib || !nodes || !vertices || !triangles || n_tris == 0) {
    set_error("hfcl_lib_add_bvh: null/empty input");
    return -1;
  }
  if (n_nodes != 2 * n_tris - 1) {  // BVH_model.cpp:821-825
    set_error("hfcl_lib_add_bvh: a BVHModel with T triangles has exactly 2T-1 nodes");
    return -1;
  }
  if (n_nodes > 65535) {
    set_error("hfcl_lib_add_bvh: more than 65535 BV nodes per model (16-bit node ids on the device stack)");
    return -1;
  }
  for (size_t i = 0; i < n_nodes; ++i) {
    const int fc = nodes[i].first_child;
    if (fc == 0 || (fc > 0 && size_t(fc) + 1 > n_nodes - 1) || (fc < 0 && size_t(-(fc + 1)) >= n_tris)) {
      set_error("hfcl_lib_add_bvh: malformed node array (first_child out of range)");
      return -1;
    }
  }
  for (size_t i = 0; i < 3 * n_tris; ++i)
    if (triangles[i] >= n_vertices) {
      set_error("hfcl_lib_add_bvh: triangle vertex index out of range");
      return -1;
    }
  DMesh m;
  m.node_off = uint32_t(lib->h_bvh_nodes.size());
  m.vert_off = uint32_t(lib->h_bvh_verts.size() / 3);
  m.tri_off = uint32_t(lib->h_bvh_tris.size() / 3);
  m.n_nodes = uint32_t(n_nodes);
  lib->h_bvh_nodes.insert(lib->h_bvh_nodes.end(), nodes, nodes + n_nodes);
  lib->h_bvh_verts.insert(lib->h_bvh_verts.end(), vertices, vertices + 3 * n_vertices);
  lib->h_bvh_tris.insert(lib->h_bvh_tris.end(), triangles, triangles + 3 * n_tris);
  lib->h_meshes.push_back(m);
  lib->bvh_dirty = true;
  return int(lib->h_meshes.size() - 1);
}

}  // extern "C"

static int ensure_workspace(hfcl_lib* lib, size_t n) {
  if (n <= lib->ws_capacity) return HFCL_OK;
  size_t cap = n + n / 8 + 1024;
  hipFree(lib->d_lists);
  hipFree(lib->d_epa_queue);
  hipFree(lib->d_epa_queue2);
  hipFree(lib->d_epa_resume);
  lib->d_lists = nullptr;
  lib->d_epa_queue = nullptr;
  lib->d_epa_queue2 = nullptr;
  lib->d_epa_resume = nullptr;
  lib->resume_cap = 0;
  lib->ws_capacity = 0;
  HIP_TRY(hipMalloc(&lib->d_lists, size_t(B_COUNT) * cap * sizeof(uint32_t)));
  HIP_TRY(hipMalloc(&lib->d_epa_queue, cap * sizeof(EpaItem<double>)));
  HIP_TRY(hipMalloc(&lib->d_epa_queue2, cap * sizeof(EpaItem<double>)));
  // saved polytopes for the tier hand-over: room for a third of the batch (beyond that the full tier
  // simply redoes the pair from its seed); 4 KB per slot in fp64
  size_t rcap = std::min(cap, std::max<size_t>(65536, cap / 3));
  if (const char* e = getenv("HFCL_EPA_RESUME_SLOTS")) rcap = std::max<size_t>(1, std::min<size_t>(cap, strtoull(e, nullptr, 10)));  // test knob
  HIP_TRY(hipMalloc(&lib->d_epa_resume, rcap * sizeof(EpaScratch<double, epa_fast_cap<double>>)));
  lib->resume_cap = rcap;
  lib->ws_capacity = cap;
  return HFCL_OK;
}

template <typename T>
static DNode<T> pack_node(const hfcl_bvh_node& n) {
  DNode<T> d;
  d.first_child = n.first_child;
  d.pad_ = 0;
  const double* a = n.obb_axes;  // column-major
  d.axes.r0 = mk<T>(T(a[0]), T(a[3]), T(a[6]));
  d.axes.r1 = mk<T>(T(a[1]), T(a[4]), T(a[7]));
  d.axes.r2 = mk<T>(T(a[2]), T(a[5]), T(a[8]));
  d.To = mk<T>(T(n.obb_To[0]), T(n.obb_To[1]), T(n.obb_To[2]));
  d.extent = mk<T>(T(n.obb_extent[0]), T(n.obb_extent[1]), T(n.obb_extent[2]));
  return d;
}

static int upload_bvh(hfcl_lib* lib) {
  if (!lib->bvh_dirty) return HFCL_OK;
  hipFree(lib->d_nodes64); hipFree(lib->d_nodes32); hipFree(lib->d_bverts64); hipFree(lib->d_bverts32);
  hipFree(lib->d_btris); hipFree(lib->d_meshes); hipFree(lib->d_rss64); hipFree(lib->d_rss32);
  lib->d_rss64 = nullptr; lib->d_rss32 = nullptr;
  lib->d_nodes64 = nullptr; lib->d_nodes32 = nullptr; lib->d_bverts64 = nullptr; lib->d_bverts32 = nullptr;
  lib->d_btris = nullptr; lib->d_meshes = nullptr;
  const size_t nn = lib->h_bvh_nodes.size(), nv = lib->h_bvh_verts.size(), nt = lib->h_bvh_tris.size();
  std::vector<DNode<double>> n64(nn);
  std::vector<DNode<float>> n32(nn);
  std::vector<DRss<double>> r64(nn);
  std::vector<DRss<float>> r32(nn);
  for (size_t i = 0; i < nn; ++i) {
    const hfcl_bvh_node& hn = lib->h_bvh_nodes[i];
    n64[i] = pack_node<double>(hn);
    n32[i] = pack_node<float>(hn);
    r64[i].Tr = mk<double>(hn.rss_Tr[0], hn.rss_Tr[1], hn.rss_Tr[2]);
    r64[i].l0 = hn.rss_length[0];
    r64[i].l1 = hn.rss_length[1];
    r64[i].r = hn.rss_radius;
    r32[i].Tr = mk<float>(float(hn.rss_Tr[0]), float(hn.rss_Tr[1]), float(hn.rss_Tr[2]));
    // fp32 image of the RSS must still contain the fp64 one: round the radius up a little
    r32[i].l0 = float(hn.rss_length[0]);
    r32[i].l1 = float(hn.rss_length[1]);
    r32[i].r = float(hn.rss_radius) * (1.0f + 4e-7f) + 1e-7f;
  }
  std::vector<float> v32(nv);
  for (size_t i = 0; i < nv; ++i) v32[i] = float(lib->h_bvh_verts[i]);
  HIP_TRY(hipMalloc(&lib->d_nodes64, nn * sizeof(DNode<double>)));
  HIP_TRY(hipMalloc(&lib->d_nodes32, nn * sizeof(DNode<float>)));
  HIP_TRY(hipMalloc(&lib->d_bverts64, nv * sizeof(double)));
  HIP_TRY(hipMalloc(&lib->d_bverts32, nv * sizeof(float)));
  HIP_TRY(hipMalloc(&lib->d_btris, nt * sizeof(uint32_t)));
  HIP_TRY(hipMalloc(&lib->d_meshes, lib->h_meshes.size() * sizeof(DMesh)));
  HIP_TRY(hipMemcpy(lib->d_nodes64, n64.data(), nn * sizeof(DNode<double>), hipMemcpyHostToDevice));
  HIP_TRY(hipMemcpy(lib->d_nodes32, n32.data(), nn * sizeof(DNode<float>), hipMemcpyHostToDevice));
  HIP_TRY(hipMalloc(&lib->d_rss64, nn * sizeof(DRss<double>)));
  HIP_TRY(hipMalloc(&lib->d_rss32, nn * sizeof(DRss<float>)));
  HIP_TRY(hipMemcpy(lib->d_rss64, r64.data(), nn * sizeof(DRss<double>), hipMemcpyHostToDevice));
  HIP_TRY(hipMemcpy(lib->d_rss32, r32.data(), nn * sizeof(DRss<float>), hipMemcpyHostToDevice));
  HIP_TRY(hipMemcpy(lib->d_bverts64, lib->h_bvh_verts.data(), nv * sizeof(double), hipMemcpyHostToDevice));
  HIP_TRY(hipMemcpy(lib->d_bverts32, v32.data(), nv * sizeof(float), hipMemcpyHostToDevice));
  HIP_TRY(hipMemcpy(lib->d_btris, lib->h_bvh_tris.data(), nt * sizeof(uint32_t), hipMemcpyHostToDevice));
  HIP_TRY(hipMemcpy(lib->d_meshes, lib->h_meshes.data(), lib->h_meshes.size() * sizeof(DMesh), hipMemcpyHostToDevice));
  lib->bvh_dirty = false;
  return HFCL_OK;
}

static KernelTime* timer_slot(hfcl_lib* lib, size_t i, const char* name) {
  while (lib->timers.size() <= i) {
    KernelTime t;
    t.name = "";
    t.used = false;
    hipEventCreate(&t.e0);
    hipEventCreate(&t.e1);
    lib->timers.push_back(t);
  }
  lib->timers[i].name = name;
  lib->timers[i].used = true;
  return &lib->timers[i];
}

template <typename T>
static void fill_qparams(QParams<T>& q, const hfcl_query_request& r) {
  q.gjk.tolerance = T(r.gjk_tolerance);
  q.gjk.max_iterations = r.gjk_max_iterations;
  q.gjk.variant = r.gjk_variant;
  q.gjk.crit = r.gjk_convergence_criterion;
  q.gjk.crit_type = r.gjk_convergence_criterion_type;
  q.epa_tolerance = T(r.epa_tolerance);
  q.epa_max_iterations = int(r.epa_max_iterations);
  q.collision_distance_threshold = T(r.collision_distance_threshold);
  q.guess_mode = r.gjk_initial_guess;
  q.guess[0] = T(r.cached_gjk_guess[0]);
  q.guess[1] = T(r.cached_gjk_guess[1]);
  q.guess[2] = T(r.cached_gjk_guess[2]);
}

static int validate_query(const hfcl_query_request& q) {
  if (!(q.gjk_tolerance > 0) || !(q.epa_tolerance > 0)) {
    set_error("tolerance must be positive (gjk.cpp:62)");
    return HFCL_ERR_INVALID_ARGUMENT;
  }
  if (q.epa_max_iterations > (uint32_t)EPA_MAX_ITER) {
    set_error("epa_max_iterations > 64 exceeds the device polytope capacity");
    return HFCL_ERR_LIMIT;
  }
  if (q.gjk_variant < 0 || q.gjk_variant > 2 || q.gjk_convergence_criterion < 0 || q.gjk_convergence_criterion > 2 ||
      q.gjk_convergence_criterion_type < 0 || q.gjk_convergence_criterion_type > 1) {
    set_error("invalid GJK variant / convergence criterion");
    return HFCL_ERR_INVALID_ARGUMENT;
  }
  if (q.gjk_initial_guess < 0 || q.gjk_initial_guess > HFCL_GUESS_BOUNDING_VOLUME) {
    set_error("Wrong initial guess for GJK.");  // narrowphase.h:379-380
    return HFCL_ERR_INVALID_ARGUMENT;
  }
  return HFCL_OK;
}

// Lane-group width of the convex GJK kernels.  A/B on cfg3 / cfg5 (profiles/r01_k_gjk_lane_group_w2.txt): 2-lane
// groups (16 vertices of each hull per lane, 32 pairs per wave: half the redundancy of the serial simplex code)
// beat 4-lane groups wherever their 96 / 192 vertex registers fit -- everywhere but fp64 convex x convex.
template <typename T, int M>
static int auto_cvx_w() {
  return (sizeof(T) == 8 && M == 0) ? 4 : 2;
}
template <typename T, int M>
static void launch_cvx_m(hfcl_lib* lib, const Work& wk, const LibView<T>& lv, const IO<T>& io, const QParams<T>& q,
                         hipStream_t st, size_t n) {
  const int w = lib->cvx_w ? lib->cvx_w : auto_cvx_w<T, M>();
  size_t b = (n + size_t(256 / w) - 1) / size_t(256 / w);
  if (b < 1) b = 1;
  if (b > size_t(lib->n_cus) * 16) b = size_t(lib->n_cus) * 16;
  const int grid = int(b);
  if (q.guess_mode == HFCL_GUESS_BOUNDING_VOLUME) {  // instantiated for the widths in use only
    if (w == 2) launch_gjk_cvx<2, M, true>(grid, st, wk, lv, io, q);
    else launch_gjk_cvx<4, M, true>(grid, st, wk, lv, io, q);
    return;
  }
  switch (w) {
    case 2: launch_gjk_cvx<2, M>(grid, st, wk, lv, io, q); break;
    case 8: launch_gjk_cvx<8, M>(grid, st, wk, lv, io, q); break;
    case 16: launch_gjk_cvx<16, M>(grid, st, wk, lv, io, q); break;
    case 32: launch_gjk_cvx<32, M>(grid, st, wk, lv, io, q); break;
    case 64: launch_gjk_cvx<64, M>(grid, st, wk, lv, io, q); break;
    default: launch_gjk_cvx<4, M>(grid, st, wk, lv, io, q); break;
  }
}
template <typename T>
static void launch_cvx(hfcl_lib* lib, const Work& wk, const LibView<T>& lv, const IO<T>& io, const QParams<T>& q,
                       hipStream_t st, size_t& ti, size_t n) {
  KernelTime* t = nullptr;
  auto tbeg = [&](const char* name) {
    if (!lib->kernel_timing) return;
    t = timer_slot(lib, ti++, name);
    hipEventRecord(t->e0, st);
  };
  auto tend = [&]() {
    if (lib->kernel_timing) hipEventRecord(t->e1, st);
  };
  if ((lib->possible_buckets >> B_CC) & 1u) {
    tbeg("k_gjk_cvx<cc>");
    launch_cvx_m<T, 0>(lib, wk, lv, io, q, st, n);
    tend();
  }
  if ((lib->possible_buckets >> B_PC) & 1u) {
    tbeg("k_gjk_cvx<pc>");
    launch_cvx_m<T, 1>(lib, wk, lv, io, q, st, n);
    tend();
  }
  if ((lib->possible_buckets >> B_CP) & 1u) {
    tbeg("k_gjk_cvx<cp>");
    launch_cvx_m<T, 2>(lib, wk, lv, io, q, st, n);
    tend();
  }
}

// The whole pipeline for one batch, asynchronous on `st`.
template <typename T>
static int run_batch_one(hfcl_lib* lib, const uint32_t* d_s1, const uint32_t* d_s2, IO<T> io, size_t n, QParams<T> q,
                         hipStream_t st) {
  if (n == 0) return HFCL_OK;
  if (n > 0xFFFFFFF0ull) {
    set_error("batch too large (max 2^32-16 pairs per call)");
    return HFCL_ERR_LIMIT;
  }
  HIP_TRY(hipSetDevice(lib->device));
  int rc = ensure_workspace(lib, n);
  if (rc) return rc;
  Work wk;
  wk.shape1 = d_s1;
  wk.shape2 = d_s2;
  wk.n = uint32_t(n);
  wk.lists = lib->d_lists;
  wk.counts = lib->d_counts;
  wk.epa_queue = lib->d_epa_queue;
  wk.epa_queue2 = lib->d_epa_queue2;
  wk.epa_resume = lib->d_epa_resume;
  wk.epa_v0 = lib->d_epa_v0;
  wk.resume_cap = uint32_t(std::min<size_t>(lib->resume_cap, 0xFFFFFFFFu));
  LibView<T> lv;
  lv.shapes = std::is_same<T, double>::value ? (const DShape<T>*)lib->d_shapes64 : (const DShape<T>*)lib->d_shapes32;
  lv.verts = std::is_same<T, double>::value ? (const T*)lib->d_verts64 : (const T*)lib->d_verts32;
  lv.kinds = lib->d_kinds;
  lv.n_shapes = uint32_t(lib->n_shapes);

  for (auto& t : lib->timers) t.used = false;
  size_t ti = 0;
  const int max_blocks = lib->n_cus * 16;
  auto blocks_for = [&](size_t items, size_t per_block) {
    size_t b = (items + per_block - 1) / per_block;
    if (b < 1) b = 1;
    if (b > (size_t)max_blocks) b = max_blocks;
    return int(b);
  };
  KernelTime* t = nullptr;
  auto tbeg = [&](const char* name) {
    if (!lib->kernel_timing) return;
    t = timer_slot(lib, ti++, name);
    hipEventRecord(t->e0, st);
  };
  auto tend = [&]() {
    if (lib->kernel_timing) hipEventRecord(t->e1, st);
  };
  // buckets no pair of this library's shape kinds can fall into are not launched at all
  auto may = [&](int b) { return (lib->possible_buckets >> b) & 1u; };
  const bool any_gjk = may(B_PRIM) || may(B_CC) || may(B_PC) || may(B_CP) || may(B_LARGE);
  const bool bvg = q.guess_mode == HFCL_GUESS_BOUNDING_VOLUME;
  HIP_TRY(hipMemsetAsync(lib->d_counts, 0, (B_COUNT + 3) * sizeof(uint32_t), st));
  tbeg("k_classify");
  hipLaunchKernelGGL(k_classify, dim3(blocks_for(n, CLS_BLOCK * 8)), dim3(CLS_BLOCK), 0, st, wk, lib->d_kinds, uint32_t(lib->n_shapes), q.mode != 1);
  tend();

  if (may(B_CLOSED)) {
    tbeg("k_closed");
    if constexpr (sizeof(T) == 8) {
      if (lib->closed_staged)
        hipLaunchKernelGGL(k_closed_staged, dim3(blocks_for(n, 256)), dim3(256), 0, st, wk, lv, io, q);
      else
        hipLaunchKernelGGL((k_closed<T>), dim3(blocks_for(n, 256)), dim3(256), 0, st, wk, lv, io, q);
    } else {
      hipLaunchKernelGGL((k_closed<T>), dim3(blocks_for(n, 256)), dim3(256), 0, st, wk, lv, io, q);
    }
    tend();
  }
  if (may(B_PRIM)) {
    tbeg("k_gjk_prim");
    if (bvg)
      hipLaunchKernelGGL((k_gjk_prim<T, true>), dim3(blocks_for(n, 256)), dim3(256), 0, st, wk, lv, io, q);
    else
      hipLaunchKernelGGL((k_gjk_prim<T, false>), dim3(blocks_for(n, 256)), dim3(256), 0, st, wk, lv, io, q);
    tend();
  }

  launch_cvx<T>(lib, wk, lv, io, q, st, ti, n);

  if (may(B_LARGE)) {
    tbeg("k_gjk_large");
    if (bvg)
      hipLaunchKernelGGL((k_gjk_large<T, true>), dim3(blocks_for(n, 256 / LARGE_W)), dim3(256), 0, st, wk, lv, io, q);
    else
      hipLaunchKernelGGL((k_gjk_large<T, false>), dim3(blocks_for(n, 256 / LARGE_W)), dim3(256), 0, st, wk, lv, io, q);
    tend();
  }

  if (may(B_TRI)) {
    tbeg("k_triangle");
    hipLaunchKernelGGL((k_triangle<T>), dim3(blocks_for(n / 8 + 1, 64 / BS_W)), dim3(64), 0, st, wk, lv, io, q);
    tend();
  }

  if (!lib->h_meshes.empty() && (may(B_BVH) || may(B_BVHSHAPE))) {
    rc = upload_bvh(lib);
    if (rc) return rc;
    BvhView<T> bv;
    bv.nodes = std::is_same<T, double>::value ? (const DNode<T>*)lib->d_nodes64 : (const DNode<T>*)lib->d_nodes32;
    bv.rss = std::is_same<T, double>::value ? (const DRss<T>*)lib->d_rss64 : (const DRss<T>*)lib->d_rss32;
    bv.verts = std::is_same<T, double>::value ? (const T*)lib->d_bverts64 : (const T*)lib->d_bverts32;
    bv.tris = lib->d_btris;
    bv.meshes = lib->d_meshes;
    bv.n_meshes = uint32_t(lib->h_meshes.size());
    if (q.mode == 1) {
      tbeg("k_bvh_shape");
      hipLaunchKernelGGL((k_bvh_shape<T>), dim3(blocks_for(n / 8 + 1, 64 / BS_W)), dim3(64), 0, st, wk, lv, bv, io, q, lib->bvh_params,
                         T(lib->break_distance * lib->break_distance));
      tend();
      tbeg("k_bvh_collide");
      hipLaunchKernelGGL((k_bvh_collide<T>), dim3(blocks_for(n, BVH_BLOCK)), dim3(BVH_BLOCK), 0, st, wk, lv, bv, io, q,
                         lib->bvh_params, T(lib->break_distance * lib->break_distance));
      tend();
    } else {
      tbeg("k_bvh_shape_distance");
      hipLaunchKernelGGL((k_bvh_shape_distance<T>), dim3(blocks_for(n / 8 + 1, 64 / BS_W)), dim3(64), 0, st, wk, lv, bv, io, q);
      tend();
      tbeg("k_bvh_distance");
      hipLaunchKernelGGL((k_bvh_distance<T>), dim3(blocks_for(n, BVHD_BLOCK)), dim3(BVHD_BLOCK), 0, st, wk, lv, bv, io, q);
      tend();
    }
  }

  tbeg("k_unsupported");
  hipLaunchKernelGGL((k_unsupported<T>), dim3(blocks_for(n, 256 * 64)), dim3(256), 0, st, wk, io, int(B_UNSUPPORTED));
  if (lib->h_meshes.empty() && may(B_BVHSHAPE))  // BVHModel x shape pairs without any registered mesh
    hipLaunchKernelGGL((k_unsupported<T>), dim3(blocks_for(n, 256 * 64)), dim3(256), 0, st, wk, io, int(B_BVHSHAPE));
  tend();

  if (q.compute_penetration && any_gjk) {
    tbeg("k_epa<fast>");
    // fp32 streams (two waves per SIMD hide the refill's global loads: k_epa<fast> 1.87 -> 1.76 ms on cfg3);
    // fp64 runs one wave per SIMD, where the more frequent refills cost more than the idle groups (cfg5
    // 1.27 -> 1.55 ms), and stays with the batch form
    if constexpr (sizeof(T) == 4)
      hipLaunchKernelGGL((k_epa_stream<T, EPA_WE, EPA_FAST_CAP>), dim3(blocks_for(n, 64 / EPA_WE)), dim3(64), 0, st, wk, lv, io, q);
    else
      hipLaunchKernelGGL((k_epa<T, EPA_WE, epa_fast_cap<T>, 1>), dim3(blocks_for(n, 64 / EPA_WE)), dim3(64), 0, st, wk, lv, io, q);
    tend();
    tbeg("k_epa<full>");
    hipLaunchKernelGGL((k_epa<T, EPA_WE2, EPA_MAX_ITER, 2>), dim3(blocks_for(n / 16 + 1, 64 / EPA_WE2)), dim3(64), 0, st, wk, lv, io, q);
    tend();
  }
  HIP_TRY(hipMemcpyAsync(lib->h_counts, lib->d_counts, (B_COUNT + 2) * sizeof(uint32_t), hipMemcpyDeviceToHost, st));
  HIP_TRY(hipGetLastError());
  return HFCL_OK;
}

// shallow clone for the second half of a split batch: shares the device shape tables, owns everything else
static hfcl_lib* make_helper(hfcl_lib* lib) {
  hfcl_lib* h = new hfcl_lib;
  h->is_helper = true;
  h->device = lib->device;
  h->n_shapes = lib->n_shapes;
  h->d_shapes64 = lib->d_shapes64;
  h->d_shapes32 = lib->d_shapes32;
  h->d_verts64 = lib->d_verts64;
  h->d_verts32 = lib->d_verts32;
  h->d_kinds = lib->d_kinds;
  h->possible_buckets = lib->possible_buckets;
  h->cvx_w = lib->cvx_w;
  h->closed_staged = lib->closed_staged;
  h->n_cus = lib->n_cus;
  bool ok = hipMalloc(&h->d_counts, (B_COUNT + 3) * sizeof(uint32_t)) == hipSuccess;
  ok = ok && hipHostMalloc((void**)&h->h_counts, (B_COUNT + 2) * sizeof(uint32_t), hipHostMallocDefault) == hipSuccess;
  ok = ok && hipMalloc(&h->d_epa_v0, size_t(h->n_cus) * 16 * (64 / EPA_WE2) * EPA_MAX_VERTS * sizeof(Quad<double>)) == hipSuccess;
  if (!ok) {
    hfcl_lib_destroy(h);
    return nullptr;
  }
  memset(h->h_counts, 0, (B_COUNT + 2) * sizeof(uint32_t));
  return h;
}

template <typename T> static IO<T> io_at(const IO<T>& io, size_t lo);
template <> IO<double> io_at(const IO<double>& io, size_t lo) {
  return IO<double>{io.tf1 + 12 * lo, io.tf2 + 12 * lo, io.out + lo, io.gin ? io.gin + lo : nullptr, io.gout ? io.gout + lo : nullptr};
}
template <> IO<float> io_at(const IO<float>& io, size_t lo) {
  return IO<float>{io.tf1 + 7 * lo, io.tf2 + 7 * lo, io.out + lo, nullptr, nullptr};
}

template <typename T>
static int run_batch(hfcl_lib* lib, const uint32_t* d_s1, const uint32_t* d_s2, IO<T> io, size_t n, QParams<T> q,
                     hipStream_t st) {
  constexpr size_t MIN_SPLIT = 1u << 17;
  lib->last_split = false;
  // Automatic choice: a library whose pairs spread over three or more of the iterative buckets (mixed scenes: cfg5
  // 4.05 -> 3.80 ms) -- the halves then run different kernels side by side; with one or two kernels in the batch the
  // halves only share the machine phase by phase and the doubled fixed costs lose 3 % (cfg2, cfg3).  A/B in
  // profiles/r01_k_two_stream_overlap.txt.
  int parts = lib->split;
  if (parts == 0) {
    int kinds = 0;
    for (int b : {int(B_PRIM), int(B_CC), int(B_PC), int(B_CP), int(B_LARGE)}) kinds += (lib->possible_buckets >> b) & 1u;
    parts = kinds >= 3 ? 2 : 1;
  }
  // meshes keep query-wide side state (contact lists, pair ids in them): they run unsplit
  if (parts < 2 || n < MIN_SPLIT || !lib->h_meshes.empty()) return run_batch_one<T>(lib, d_s1, d_s2, io, n, q, st);
  HIP_TRY(hipSetDevice(lib->device));
  if (!lib->helper) {
    lib->helper = make_helper(lib);
    if (!lib->helper || hipStreamCreateWithFlags(&lib->side, hipStreamNonBlocking) != hipSuccess ||
        hipEventCreateWithFlags(&lib->ev_fork, hipEventDisableTiming) != hipSuccess ||
        hipEventCreateWithFlags(&lib->ev_join, hipEventDisableTiming) != hipSuccess) {
      set_error("split batches: HIP allocation failed");
      return HFCL_ERR_HIP;
    }
  }
  hfcl_lib* h2 = lib->helper;
  h2->kernel_timing = lib->kernel_timing;
  h2->break_distance = lib->break_distance;
  h2->bvh_params = lib->bvh_params;
  const size_t h = n / 2;  // unequal parts (0.35 / 0.6 / 0.7 of the batch first) measured slower on cfg3 and cfg5
  HIP_TRY(hipEventRecord(lib->ev_fork, st));  // the inputs are ready where the caller's stream stands now
  HIP_TRY(hipStreamWaitEvent(lib->side, lib->ev_fork, 0));
  int rc = run_batch_one<T>(lib, d_s1, d_s2, io, h, q, st);
  if (rc) return rc;
  rc = run_batch_one<T>(h2, d_s1 + h, d_s2 + h, io_at<T>(io, h), n - h, q, lib->side);
  if (rc) return rc;
  HIP_TRY(hipEventRecord(lib->ev_join, lib->side));
  HIP_TRY(hipStreamWaitEvent(st, lib->ev_join, 0));  // results are complete in the caller's stream order
  lib->last_split = true;
  return HFCL_OK;
}

template <typename T>
static int setup_collide(const hfcl_collision_request* req, QParams<T>& q, bool& skip_all) {
  skip_all = false;
  if (!req) {
    set_error("null request");
    return HFCL_ERR_INVALID_ARGUMENT;
  }
  if (req->num_max_contacts == 0) {  // src/collision.cpp:82-85
    set_error("Invalid number of max contacts (current value is 0).");
    return HFCL_ERR_INVALID_ARGUMENT;
  }
  int rc = validate_query(req->q);
  if (rc) return rc;
  fill_qparams(q, req->q);
  q.mode = 1;
  q.compute_penetration = (req->enable_contact || req->security_margin < 0) ? 1 : 0;  // shape_shape_func.h:141-142
  q.security_margin = T(req->security_margin);
  // narrowphase.h:228-229
  double ub = req->distance_upper_bound > req->security_margin ? req->distance_upper_bound : req->security_margin;
  if (ub < 0) ub = 0;
  q.gjk.distance_upper_bound = (ub >= double(Lim<T>::max())) ? Lim<T>::max() : T(ub);
  if (req->security_margin == -__builtin_inf()) skip_all = true;  // src/collision.cpp:73-76
  return HFCL_OK;
}
template <typename T>
static int setup_distance(const hfcl_distance_request* req, QParams<T>& q) {
  if (!req) {
    set_error("null request");
    return HFCL_ERR_INVALID_ARGUMENT;
  }
  int rc = validate_query(req->q);
  if (rc) return rc;
  fill_qparams(q, req->q);
  q.mode = 0;
  q.compute_penetration = req->enable_signed_distance ? 1 : 0;
  q.security_margin = T(0);
  q.gjk.distance_upper_bound = Lim<T>::max();  // narrowphase.h:175
  return HFCL_OK;
}

// -inf security margin: cleared results, nothing computed (src/collision.cpp:73-76)
template <typename R>
__global__ void k_fill_skipped(R* out, uint32_t n) {
  for (uint32_t i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) {
    R r;
    memset(&r, 0, sizeof(r));
    r.distance = 3.402823466e+38f;
    r.status = 0x80000000u;
    out[i] = r;
  }
}
template <>
__global__ void k_fill_skipped<hfcl_result>(hfcl_result* out, uint32_t n) {
  for (uint32_t i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) {
    hfcl_result r;
    const double x = __builtin_nan("");
    r.distance = 1.7976931348623157e+308;
    for (int k = 0; k < 3; ++k) r.normal[k] = r.p1[k] = r.p2[k] = x;
    r.b1 = r.b2 = -1;
    r.status = 0x80000000u;
    r.num_contacts = 0;
    out[i] = r;
  }
}

extern "C" {

int hfcl_collide_batch_device(hfcl_lib* lib, const uint32_t* d_shape1, const uint32_t* d_shape2, const double* d_tf1,
                              const double* d_tf2, size_t n, const hfcl_collision_request* req, hfcl_result* d_out,
                              const hfcl_guess* d_guess_in, hfcl_guess* d_guess_out, void* stream) {
  if (!lib) {
    set_error("null library");
    return HFCL_ERR_INVALID_ARGUMENT;
  }
  QParams<double> q;
  bool skip;
  int rc = setup_collide<double>(req, q, skip);
  if (rc) return rc;
  hipStream_t st = (hipStream_t)stream;
  if (skip) {
    HIP_TRY(hipSetDevice(lib->device));
    if (n) hipLaunchKernelGGL((k_fill_skipped<hfcl_result>), dim3(1024), dim3(256), 0, st, d_out, uint32_t(n));
    return HFCL_OK;
  }
  IO<double> io{d_tf1, d_tf2, d_out, d_guess_in, d_guess_out};
  lib->bvh_params.num_max_contacts = req->num_max_contacts;
  lib->break_distance = req->break_distance;
  return run_batch<double>(lib, d_shape1, d_shape2, io, n, q, st);
}

int hfcl_distance_batch_device(hfcl_lib* lib, const uint32_t* d_shape1, const uint32_t* d_shape2, const double* d_tf1,
                               const double* d_tf2, size_t n, const hfcl_distance_request* req, hfcl_result* d_out,
                               const hfcl_guess* d_guess_in, hfcl_guess* d_guess_out, void* stream) {
  if (!lib) {
    set_error("null library");
    return HFCL_ERR_INVALID_ARGUMENT;
  }
  QParams<double> q;
  int rc = setup_distance<double>(req, q);
  if (rc) return rc;
  IO<double> io{d_tf1, d_tf2, d_out, d_guess_in, d_guess_out};
  return run_batch<double>(lib, d_shape1, d_shape2, io, n, q, (hipStream_t)stream);
}

int hfcl_distance_batch_device_f32(hfcl_lib* lib, const uint32_t* d_shape1, const uint32_t* d_shape2,
                                   const float* d_pose1, const float* d_pose2, size_t n,
                                   const hfcl_distance_request* req, hfcl_result_f32* d_out, void* stream) {
  if (!lib) {
    set_error("null library");
    return HFCL_ERR_INVALID_ARGUMENT;
  }
  QParams<float> q;
  int rc = setup_distance<float>(req, q);
  if (rc) return rc;
  IO<float> io{d_pose1, d_pose2, d_out, nullptr, nullptr};
  return run_batch<float>(lib, d_shape1, d_shape2, io, n, q, (hipStream_t)stream);
}

int hfcl_collide_batch_device_f32(hfcl_lib* lib, const uint32_t* d_shape1, const uint32_t* d_shape2,
                                  const float* d_pose1, const float* d_pose2, size_t n,
                                  const hfcl_collision_request* req, hfcl_result_f32* d_out, void* stream) {
  if (!lib) {
    set_error("null library");
    return HFCL_ERR_INVALID_ARGUMENT;
  }
  QParams<float> q;
  bool skip;
  int rc = setup_collide<float>(req, q, skip);
  if (rc) return rc;
  hipStream_t st = (hipStream_t)stream;
  if (skip) {
    HIP_TRY(hipSetDevice(lib->device));
    if (n) hipLaunchKernelGGL((k_fill_skipped<hfcl_result_f32>), dim3(1024), dim3(256), 0, st, d_out, uint32_t(n));
    return HFCL_OK;
  }
  IO<float> io{d_pose1, d_pose2, d_out, nullptr, nullptr};
  lib->bvh_params.num_max_contacts = req->num_max_contacts;
  lib->break_distance = req->break_distance;
  return run_batch<float>(lib, d_shape1, d_shape2, io, n, q, st);
}

static int ensure_staging(hfcl_lib* lib, size_t n, bool gin, bool gout) {
  if (n > lib->st_capacity) {
    hipFree(lib->d_s1); hipFree(lib->d_s2); hipFree(lib->d_tf1); hipFree(lib->d_tf2); hipFree(lib->d_out);
    hipFree(lib->d_gin); hipFree(lib->d_gout);
    lib->d_s1 = lib->d_s2 = nullptr;
    lib->d_tf1 = lib->d_tf2 = nullptr;
    lib->d_out = nullptr;
    lib->d_gin = lib->d_gout = nullptr;
    lib->st_capacity = 0;
    size_t cap = n + n / 8 + 256;
    HIP_TRY(hipMalloc(&lib->d_s1, cap * sizeof(uint32_t)));
    HIP_TRY(hipMalloc(&lib->d_s2, cap * sizeof(uint32_t)));
    HIP_TRY(hipMalloc(&lib->d_tf1, cap * 12 * sizeof(double)));
    HIP_TRY(hipMalloc(&lib->d_tf2, cap * 12 * sizeof(double)));
    HIP_TRY(hipMalloc(&lib->d_out, cap * sizeof(hfcl_result)));
    lib->st_capacity = cap;
  }
  if (gin && !lib->d_gin) HIP_TRY(hipMalloc(&lib->d_gin, lib->st_capacity * sizeof(hfcl_guess)));
  if (gout && !lib->d_gout) HIP_TRY(hipMalloc(&lib->d_gout, lib->st_capacity * sizeof(hfcl_guess)));
  return HFCL_OK;
}

static int host_batch(hfcl_lib* lib, const uint32_t* s1, const uint32_t* s2, const double* tf1, const double* tf2,
                      size_t n, const hfcl_collision_request* creq, const hfcl_distance_request* dreq, hfcl_result* out,
                      const hfcl_guess* gin, hfcl_guess* gout) {
  if (!lib) {
    set_error("null library");
    return HFCL_ERR_INVALID_ARGUMENT;
  }
  if (n == 0) return HFCL_OK;
  if (!s1 || !s2 || !tf1 || !tf2 || !out) {
    set_error("null buffer");
    return HFCL_ERR_INVALID_ARGUMENT;
  }
  HIP_TRY(hipSetDevice(lib->device));
  int rc = ensure_staging(lib, n, gin != nullptr, gout != nullptr);
  if (rc) return rc;
  HIP_TRY(hipMemcpy(lib->d_s1, s1, n * sizeof(uint32_t), hipMemcpyHostToDevice));
  HIP_TRY(hipMemcpy(lib->d_s2, s2, n * sizeof(uint32_t), hipMemcpyHostToDevice));
  HIP_TRY(hipMemcpy(lib->d_tf1, tf1, n * 12 * sizeof(double), hipMemcpyHostToDevice));
  HIP_TRY(hipMemcpy(lib->d_tf2, tf2, n * 12 * sizeof(double), hipMemcpyHostToDevice));
  if (gin) HIP_TRY(hipMemcpy(lib->d_gin, gin, n * sizeof(hfcl_guess), hipMemcpyHostToDevice));
  if (creq)
    rc = hfcl_collide_batch_device(lib, lib->d_s1, lib->d_s2, lib->d_tf1, lib->d_tf2, n, creq, lib->d_out,
                                   gin ? lib->d_gin : nullptr, gout ? lib->d_gout : nullptr, nullptr);
  else
    rc = hfcl_distance_batch_device(lib, lib->d_s1, lib->d_s2, lib->d_tf1, lib->d_tf2, n, dreq, lib->d_out,
                                    gin ? lib->d_gin : nullptr, gout ? lib->d_gout : nullptr, nullptr);
  if (rc) return rc;
  HIP_TRY(hipDeviceSynchronize());
  HIP_TRY(hipMemcpy(out, lib->d_out, n * sizeof(hfcl_result), hipMemcpyDeviceToHost));
  if (gout) HIP_TRY(hipMemcpy(gout, lib->d_gout, n * sizeof(hfcl_guess), hipMemcpyDeviceToHost));
  const bool skipped = creq && creq->security_margin == -__builtin_inf();
  if (!skipped && total_count(lib, B_UNSUPPORTED) > 0) {
    set_error("Collision/distance function between some node types of the batch is not yet supported (" +
              std::to_string(total_count(lib, B_UNSUPPORTED)) + " pairs; their records carry status bit 31)");
    return HFCL_ERR_UNSUPPORTED_PAIR;
  }
  {
    // a TriangleP built inside the reference (top-level TriangleP overloads, mesh x shape leaves) never had
    // computeLocalAABB() called: BoundingVolumeGuess throws there (narrowphase.h:366-373)
    const hfcl_query_request& qq = creq ? creq->q : dreq->q;
    if (!skipped && qq.gjk_initial_guess == HFCL_GUESS_BOUNDING_VOLUME &&
        (total_count(lib, B_TRI) > 0 || total_count(lib, B_BVHSHAPE) > 0)) {
      set_error("computeLocalAABB must have been called on the shapes before using GJKInitialGuess::BoundingVolumeGuess.");
      return HFCL_ERR_INVALID_ARGUMENT;
    }
  }
  if (!skipped && creq && creq->security_margin < 0 && total_count(lib, B_BVHSHAPE) > 0) {
    set_error("Negative security margin are not handled yet for BVHModel");  // collision_func_matrix.cpp:109-112
    return HFCL_ERR_INVALID_ARGUMENT;
  }
  if (!skipped && (total_count(lib, B_BVH) > 0 || total_count(lib, B_BVHSHAPE) > 0) && lib->h_meshes.empty()) {
    set_error("BVH shapes in the batch but no BVHModel registered (hfcl_lib_add_bvh)");
    return HFCL_ERR_INVALID_ARGUMENT;
  }
  return HFCL_OK;
}

int hfcl_collide_batch(hfcl_lib* lib, const uint32_t* shape1, const uint32_t* shape2, const double* tf1,
                       const double* tf2, size_t n, const hfcl_collision_request* req, hfcl_result* out,
                       const hfcl_guess* guess_in, hfcl_guess* guess_out) {
  if (!req) {
    set_error("null request");
    return HFCL_ERR_INVALID_ARGUMENT;
  }
  return host_batch(lib, shape1, shape2, tf1, tf2, n, req, nullptr, out, guess_in, guess_out);
}
int hfcl_distance_batch(hfcl_lib* lib, const uint32_t* shape1, const uint32_t* shape2, const double* tf1,
                        const double* tf2, size_t n, const hfcl_distance_request* req, hfcl_result* out,
                        const hfcl_guess* guess_in, hfcl_guess* guess_out) {
  if (!req) {
    set_error("null request");
    return HFCL_ERR_INVALID_ARGUMENT;
  }
  return host_batch(lib, shape1, shape2, tf1, tf2, n, nullptr, req, out, guess_in, guess_out);
}

int hfcl_collide_batch_contacts(hfcl_lib* lib, const uint32_t* shape1, const uint32_t* shape2, const double* tf1,
                                const double* tf2, size_t n, const hfcl_collision_request* req, hfcl_result* out,
                                hfcl_contact* contacts, size_t max_contacts_total, size_t* n_contacts_out) {
  if (!lib || !req || !contacts || !n_contacts_out) {
    set_error("null argument");
    return HFCL_ERR_INVALID_ARGUMENT;
  }
  HIP_TRY(hipSetDevice(lib->device));
  if (max_contacts_total > lib->contacts_cap) {
    hipFree(lib->d_contacts);
    lib->d_contacts = nullptr;
    lib->contacts_cap = 0;
    HIP_TRY(hipMalloc(&lib->d_contacts, max_contacts_total * sizeof(hfcl_contact)));
    lib->contacts_cap = max_contacts_total;
  }
  if (!lib->d_contacts_count) HIP_TRY(hipMalloc(&lib->d_contacts_count, sizeof(uint32_t)));
  HIP_TRY(hipMemset(lib->d_contacts_count, 0, sizeof(uint32_t)));
  lib->bvh_params.contacts = lib->d_contacts;
  lib->bvh_params.contacts_cap = uint32_t(max_contacts_total > 0xFFFFFFFFull ? 0xFFFFFFFFull : max_contacts_total);
  lib->bvh_params.contacts_count = lib->d_contacts_count;
  int rc = host_batch(lib, shape1, shape2, tf1, tf2, n, req, nullptr, out, nullptr, nullptr);
  lib->bvh_params.contacts = nullptr;
  lib->bvh_params.contacts_cap = 0;
  lib->bvh_params.contacts_count = nullptr;
  if (rc) return rc;
  uint32_t cnt = 0;
  HIP_TRY(hipMemcpy(&cnt, lib->d_contacts_count, sizeof(uint32_t), hipMemcpyDeviceToHost));
  const size_t stored = cnt < max_contacts_total ? cnt : max_contacts_total;
  if (stored) HIP_TRY(hipMemcpy(contacts, lib->d_contacts, stored * sizeof(hfcl_contact), hipMemcpyDeviceToHost));
  *n_contacts_out = cnt;  // number produced (may exceed the capacity; the excess was dropped)
  return HFCL_OK;
}

double hfcl_last_kernel_ms(hfcl_lib* lib) {
  if (!lib) return 0.0;
  hipSetDevice(lib->device);
  double total = 0, best = -1;
  lib->dominant = "";
  for (auto& t : lib->timers) {
    if (!t.used) continue;
    if (hipEventSynchronize(t.e1) != hipSuccess) continue;
    float ms = 0;
    if (hipEventElapsedTime(&ms, t.e0, t.e1) != hipSuccess) continue;
    total += ms;
    if (ms > best) {
      best = ms;
      lib->dominant = t.name;
    }
  }
  return total;
}
const char* hfcl_last_kernel_name(hfcl_lib* lib) { return lib ? lib->dominant.c_str() : ""; }
void hfcl_lib_set_kernel_timing(hfcl_lib* lib, int on) {
  if (!lib) return;
  lib->kernel_timing = on != 0;
  if (!on)
    for (auto& t : lib->timers) t.used = false;
}

// breakdown: up to `cap` (name, ms) entries of the last call; returns the number written
int hfcl_last_kernel_breakdown(hfcl_lib* lib, const char** names, double* ms, int cap) {
  if (!lib) return 0;
  hipSetDevice(lib->device);
  int k = 0;
  for (auto& t : lib->timers) {
    if (!t.used || k >= cap) continue;
    float m = 0;
    if (hipEventSynchronize(t.e1) != hipSuccess) continue;
    if (hipEventElapsedTime(&m, t.e0, t.e1) != hipSuccess) continue;
    // split batch: the two halves ran the same launch sequence on two streams; report the mean as-run duration of a launch
    const size_t i = size_t(&t - lib->timers.data());
    if (lib->last_split && lib->helper && i < lib->helper->timers.size() && lib->helper->timers[i].used) {
      KernelTime& u = lib->helper->timers[i];
      float m2 = 0;
      if (hipEventSynchronize(u.e1) == hipSuccess && hipEventElapsedTime(&m2, u.e0, u.e1) == hipSuccess) m = 0.5f * (m + m2);
    }
    names[k] = t.name;
    ms[k] = m;
    ++k;
  }
  return k;
}

// parts = 2: batches of at least 128k pairs (libraries without meshes) run as two halves on two streams; 1: one stream
void hfcl_lib_set_split(hfcl_lib* lib, int parts) {
  if (lib) lib->split = parts >= 2 ? 2 : (parts == 1 ? 1 : 0);
}
int hfcl_lib_get_split(const hfcl_lib* lib) { return lib ? lib->split : 0; }
int hfcl_lib_last_split_parts(const hfcl_lib* lib) { return (lib && lib->last_split) ? 2 : 1; }

// bucket populations of the last call (after a stream sync): closed, prim, cc, pc, cp, bvh, unsupported,
// epa queue, epa overflow queue
void hfcl_last_bucket_counts(hfcl_lib* lib, uint32_t* out12) {  // B_COUNT buckets + the two EPA queues
  if (lib) {  // the counters travel with an asynchronous copy at the end of the batch: wait for it
    hipSetDevice(lib->device);
    hipDeviceSynchronize();
  }
  for (int i = 0; i <= B_COUNT + 1; ++i) out12[i] = lib ? total_count(lib, i) : 0;
}

}  // extern "C"
