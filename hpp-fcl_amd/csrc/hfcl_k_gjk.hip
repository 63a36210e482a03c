// hfcl_k_gjk.hip -- classification, closed forms and the GJK kernels (see hfcl_launch.hpp for the kernel map).
#include "hfcl_dev.hpp"
#include "hfcl_launch.hpp"

// This file is compiled twice (Makefile: hfcl_k_gjk32.o / hfcl_k_gjk64.o, HFCL_UNIT_PRECISION = 32 / 64): the fp64 kernels -- the
// reference's precision, everything the oracle is compared with digit for digit -- without contraction of a*b+c (-ffp-contract=off:
// iteration counts, statuses and distances of the GJK / EPA kernels then equal the oracle's in every record; profiles/r05_c),
// the fp32 kernels with it (specified by an envelope; 5 % faster that way).  The kernels without a precision (k_classify,
// k_expand_poses, ...) belong to the fp64 object.  0: one object with everything (tools).
#ifndef HFCL_UNIT_PRECISION
#define HFCL_UNIT_PRECISION 0
#endif
#define HFCL_UNIT_F32 (HFCL_UNIT_PRECISION != 64)
#define HFCL_UNIT_F64 (HFCL_UNIT_PRECISION != 32)

// ---------------------------------------------------------------------------------------
// k_classify: bucket every pair by (kind1, kind2).  Wave-aggregated list append.
// ---------------------------------------------------------------------------------------
// One global atomic per (block trip, bucket) reserves the block's range in the bucket list: these same-address
// atomics serialise (~20 ns each), so the trips are made large -- 1024 threads x 8 pairs (4M pairs: 39 us at
// 2048 pairs per trip).
#if HFCL_UNIT_F64
__global__ void __launch_bounds__(CLS_BLOCK) k_classify(Work wk, const uint8_t* kinds, uint32_t n_shapes, bool distance_mode) {
  // Each block handles CHUNK consecutive pairs per trip: per-bucket counts are built in LDS, one
  // global atomic per (block, bucket) reserves a range, then every lane writes its pair index.
  constexpr int PER_THREAD = 8;
  constexpr uint32_t CHUNK = CLS_BLOCK * PER_THREAD;
  // virtual buckets B_COUNT + b: the curved pairs of bucket b (GJK buckets only), filed from the top end of b's list
  __shared__ uint32_t s_count[2 * B_COUNT];
  __shared__ uint32_t s_base[2 * B_COUNT];
  for (uint32_t start = blockIdx.x * CHUNK; start < wk.n; start += gridDim.x * CHUNK) {
    if (threadIdx.x < 2 * B_COUNT) s_count[threadIdx.x] = 0;
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
        if (s1 < n_shapes && s2 < n_shapes) {
          const int k1 = kinds[s1], k2 = kinds[s2];
          bk[k] = bucket_of(k1, k2, distance_mode);
          if ((bk[k] == B_PRIM || bk[k] == B_PC || bk[k] == B_CP) && curved_pair(k1, k2)) bk[k] += B_COUNT;
        } else {
          bk[k] = B_UNSUPPORTED;
        }
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
    if (threadIdx.x < 2 * B_COUNT) {
      const uint32_t c = s_count[threadIdx.x];
      const int counter = threadIdx.x < B_COUNT ? int(threadIdx.x) : B_CURVED0 + int(threadIdx.x) - B_COUNT;
      s_base[threadIdx.x] = c ? atomicAdd(&wk.counts[counter], c) : 0u;
    }
    __syncthreads();
#pragma unroll
    for (int k = 0; k < PER_THREAD; ++k) {
      if (bk[k] < 0) continue;
      const uint32_t slot = s_base[bk[k]] + rk[k], id = start + k * CLS_BLOCK + threadIdx.x;
      if (bk[k] < B_COUNT)
        wk.lists[size_t(bk[k]) * wk.n + slot] = id;
      else
        wk.lists[size_t(bk[k] - B_COUNT) * wk.n + (wk.n - 1u - slot)] = id;
    }
    __syncthreads();
  }
}

#endif
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
#if HFCL_UNIT_F64
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

#endif
// ---------------------------------------------------------------------------------------
// Shared GJK epilogue: final record, or hand-off to k_epa through the device queue.
// ---------------------------------------------------------------------------------------
// EPA on a pair with a curved shape (Ellipsoid, Cone, Cylinder: supports that are not vertices) approaches the surface
// until its tolerance is met -- 20 to 30 iterations on average, against 4 to 8 where both shapes are polytopes or
// sphere / capsule cores (measured on cfg5 with the oracle).  The fp64 fast tier steps the 8 polytopes of a wave in
// lockstep, so the two classes get a queue each (the second one is the fp32 convex x convex queue, unused in fp64) and
// a wave's polytopes are of one class: the short ones are not held to the length of the long ones.
template <typename T, class P, class PS>
__device__ __forceinline__ void finish_gjk(const Gjk<T, P>& g, const Work& wk, const IO<T>& io, const QParams<T>& q,
                                           uint32_t pair, const Pose<T>& tf1, T r0, T r1, const V3<T>& guess0,
                                           bool writer, const PS& ps, bool full_tier = false, bool cc_queue = false) {
  PairOut<T> o;
  EpaSeed<T> seed;
  const bool to_epa = gjk_finish(g, q, tf1, r0, r1, guess0, o, seed, ps);
  if (!writer) return;
  if (to_epa) {
    seed.pair = pair;
    if (cc_queue) {
      // fp32 convex x convex pairs have a fast tier of their own (k_epa_stream<.., CC>): its queue is the top end of
      // epa_queue, filled downwards from slot n-1 (the two queues of a batch hold at most n items together)
#ifdef HFCL_EXPERIMENT_NO_QUEUE_ATOMIC  // timing experiment only (the EPA kernels then see an empty queue): what does the counter cost?
      const uint32_t slot = pair;
#else
      const uint32_t slot = atomicAdd(&wk.counts[B_COUNT + 3], 1u);
#endif
      reinterpret_cast<EpaSeed<T>*>(wk.epa_queue)[wk.n - 1u - slot] = seed;
    } else {
      // full_tier: straight to the full-capacity EPA queue (pairs with a large hull: only that tier
      // can scan vertices from memory)
      const uint32_t slot = atomicAdd(&wk.counts[full_tier ? B_COUNT + 1 : B_COUNT], 1u);
      reinterpret_cast<EpaSeed<T>*>(full_tier ? wk.epa_queue2 : wk.epa_queue)[slot] = seed;
    }
  } else {
    write_out<T>(io, q, pair, o);
    write_guess<T>(io, pair, o.cached_guess, 0, 0);
  }
}

// The simplex payload policy of a GJK kernel with NT threads per block (HFCL_GJK_W0_LDS=0: A/B switch back to
// register payloads).
#if HFCL_GJK_W0_LDS
template <typename T, int NT> using GjkW0 = W0Lds<T, NT>;
#define HFCL_GJK_W0_SLAB(T, NT, name)          \
  __shared__ T name##_slab[W0Lds<T, NT>::WORDS]; \
  const W0Lds<T, NT> name{name##_slab + threadIdx.x}
#else
template <typename T, int NT> using GjkW0 = W0Regs<T>;
#define HFCL_GJK_W0_SLAB(T, NT, name) const W0Regs<T> name = W0Regs<T>()
#endif

// ---------------------------------------------------------------------------------------
// k_gjk_prim: primitive x primitive GJK, one pair per lane.
// ---------------------------------------------------------------------------------------
// BVG: GJKInitialGuess::BoundingVolumeGuess.  A separate instantiation on purpose: the register allocation of the GJK
// kernels is sensitive to anything live across their loop (the run-time form of this one select cost k_gjk_cvx<2,0>
// 0.96 -> 1.51 ms), so the default-guess kernels are compiled without it.
template <typename T, bool BVG>
__global__ void __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(HFCL_WPE_PRIM, 8))) k_gjk_prim(Work wk, LibView<T> lib, IO<T> io, QParams<T> q) {
  HFCL_GJK_W0_SLAB(T, 256, ps);
  const BucketList list = bucket_list(wk, B_PRIM);
  for (uint32_t it = blockIdx.x * blockDim.x + threadIdx.x; it < list.cnt; it += gridDim.x * blockDim.x) {
    const uint32_t pair = list[it];
    const DShape<T> a = lib.shapes[wk.shape1[pair]], b = lib.shapes[wk.shape2[pair]];
    const Pose<T> tf1 = load_pose(io.tf1, pair), tf2 = load_pose(io.tf2, pair);
    SerialSupport<T> sup;
    sup.a = a;
    sup.b = b;
    sup.va = sup.vb = nullptr;
    sup.md = make_mdiff(tf1, tf2);
    const T r0 = swept_radius(a), r1 = swept_radius(b);
    const V3<T> guess0 = initial_guess<T>(io, q, pair);
    Gjk<T, typename GjkW0<T, 256>::P> g;
    if constexpr (BVG)
      gjk_run(g, q.gjk, start_guess(q, a, b, sup.md, guess0), r0 + r1, false, sup, ps);
    else
      gjk_run(g, q.gjk, guess0, r0 + r1, false, sup, ps);
    finish_gjk<T>(g, wk, io, q, pair, tf1, r0, r1, guess0, true, ps, false, sizeof(T) == 8 && curved_pair(a.kind, b.kind));
  }
}
// Hull of a W-lane group held in LDS instead of registers: element (vertex k, component c) of thread t at
// base[(3 * k + c) * NT + t] (lane-minor: conflict-free).  Two register-resident hulls of 16 vertices per lane are
// 96 VGPRs, which together with the simplex does not fit: the allocator then spills hull vertices to scratch memory
// and reloads them one by one, each behind an s_waitcnt vmcnt(0), inside the support scan of every GJK trip
// (13 serialised scratch loads per trip in k_gjk_cvx<2,0>).  With the second hull in LDS the scan issues
// its reads back to back, the winner's coordinates are one indexed LDS read instead of a select chain, and nothing
// is spilled.
#ifndef HFCL_HULL_LDS_BATCH
#define HFCL_HULL_LDS_BATCH 4
#endif
template <typename T, int W, int NT>
struct HullLds {
  static constexpr int VPL = (HULL_MAX + W - 1) / W;
  static constexpr int WORDS = VPL * 3 * NT;
  T* lane;  // &slab[threadIdx.x]
  __device__ __forceinline__ void load(const T* verts, uint32_t n, int lig) {
#pragma unroll
    for (int k = 0; k < VPL; ++k) {
      uint32_t idx = uint32_t(lig * VPL + k);
      idx = idx < n ? idx : 0u;  // padding duplicates vertex 0 (never wins the first-index tie-break)
      const T* p = verts + 3 * size_t(idx);
      lane[(3 * k + 0) * NT] = p[0];
      lane[(3 * k + 1) * NT] = p[1];
      lane[(3 * k + 2) * NT] = p[2];
    }
  }
  __device__ __forceinline__ V3<T> support(const V3<T>& dir, int lig) const {
#if HFCL_HULL_LDS_BATCH
    // the reads of HFCL_HULL_LDS_BATCH vertices in flight together, then their products: left to itself the scheduler reads two values, waits, multiplies
    // (24 exposed LDS latencies per scan at two waves per SIMD, profiles/r06_f); the products and their order are the loop's
    constexpr int B = HFCL_HULL_LDS_BATCH;
    T best = T(0);
    int bi = lig * VPL;
#pragma unroll
    for (int k0 = 0; k0 < VPL; k0 += B) {
      T vx[B], vy[B], vz[B];
#pragma unroll
      for (int j = 0; j < B; ++j) {
        vx[j] = lane[(3 * (k0 + j) + 0) * NT];
        vy[j] = lane[(3 * (k0 + j) + 1) * NT];
        vz[j] = lane[(3 * (k0 + j) + 2) * NT];
      }
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int j = 0; j < B; ++j) {
        const T d = vx[j] * dir.x + vy[j] * dir.y + vz[j] * dir.z;
        if (k0 + j == 0) {
          best = d;
        } else if (d > best) {
          best = d;
          bi = lig * VPL + k0 + j;
        }
      }
      __builtin_amdgcn_sched_barrier(0);
    }
#else
    T best = lane[0] * dir.x + lane[NT] * dir.y + lane[2 * NT] * dir.z;
    int bi = lig * VPL;
#pragma unroll
    for (int k = 1; k < VPL; ++k) {
      const T d = lane[(3 * k + 0) * NT] * dir.x + lane[(3 * k + 1) * NT] * dir.y + lane[(3 * k + 2) * NT] * dir.z;
      if (d > best) {
        best = d;
        bi = lig * VPL + k;
      }
    }
#endif
    butterfly_stages<W>([&](auto stage) {
      constexpr int M = decltype(stage)::value;
      const T od = group_exchange<W, M>(best);
      const int oi = group_exchange<W, M>(bi);
      const bool take = (od > best) | ((od == best) & (oi < bi));  // (selects: see HullRegs::support)
      best = take ? od : best;
      bi = take ? oi : bi;
    });
    const int owner = bi / VPL, slot = bi % VPL;
    const T cx = lane[(3 * slot + 0) * NT], cy = lane[(3 * slot + 1) * NT], cz = lane[(3 * slot + 2) * NT];
    return mk<T>(__shfl(cx, owner, W), __shfl(cy, owner, W), __shfl(cz, owner, W));
  }
};
#ifndef HFCL_GJK_HULLB_LDS
#define HFCL_GJK_HULLB_LDS 1
#endif
// the second hull of a convex-convex pair lives in LDS where the block's slab fits (2- and 4-lane groups)
template <typename T, int W, int M> constexpr bool hull_b_in_lds = HFCL_GJK_HULLB_LDS && M == 0 && W <= 4 && (HULL_MAX / W) * sizeof(T) <= 64;

// M: 0 = convex-convex, 1 = prim-convex, 2 = convex-prim
template <typename T, int W, int M>
struct CvxSupport {
  DShape<T> a, b;
  HullRegs<T, W> h0;
  typename std::conditional<hull_b_in_lds<T, W, M>, HullLds<T, W, 256>, HullRegs<T, W>>::type h1;
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
  HFCL_GJK_W0_SLAB(T, 256, ps);
  __shared__ T hull_slab[hull_b_in_lds<T, W, M> ? HullLds<T, W, 256>::WORDS : 1];
  const BucketList list = bucket_list(wk, BUCKET);
  const int lig = threadIdx.x & (W - 1);
  const uint32_t groups = (gridDim.x * blockDim.x) / W;
  for (uint32_t it = (blockIdx.x * blockDim.x + threadIdx.x) / W; it < list.cnt; it += groups) {
    const uint32_t pair = list[it];
    CvxSupport<T, W, M> sup;
    if constexpr (hull_b_in_lds<T, W, M>) sup.h1.lane = hull_slab + threadIdx.x;
    sup.a = lib.shapes[wk.shape1[pair]];
    sup.b = lib.shapes[wk.shape2[pair]];
    sup.lig = lig;
    if (M != 1) sup.h0.load(lib.verts + 3 * size_t(sup.a.vertex_offset), sup.a.num_points, lig);
    if (M != 2) sup.h1.load(lib.verts + 3 * size_t(sup.b.vertex_offset), sup.b.num_points, lig);
    const Pose<T> tf1 = load_pose(io.tf1, pair), tf2 = load_pose(io.tf2, pair);
    sup.md = make_mdiff(tf1, tf2);
    const T r0 = swept_radius(sup.a), r1 = swept_radius(sup.b);
    const V3<T> guess0 = initial_guess<T>(io, q, pair);
    Gjk<T, typename GjkW0<T, 256>::P> g;
    // normalize_support_direction only when both are ConvexBase (minkowski_difference.cpp:261-266)
    if constexpr (BVG)
      gjk_run(g, q.gjk, start_guess(q, sup.a, sup.b, sup.md, guess0), r0 + r1, M == 0, sup, ps);
    else
      gjk_run(g, q.gjk, guess0, r0 + r1, M == 0, sup, ps);
    // second EPA queue: fp32 -- the convex x convex pairs (their own kernel); fp64 -- the pairs with a curved shape
    finish_gjk<T>(g, wk, io, q, pair, tf1, r0, r1, guess0, lig == 0, ps, false,
                  sizeof(T) == 4 ? M == 0 : (M != 0 && curved_pair(sup.a.kind, sup.b.kind)));
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
template <typename T>
struct LargeSupport {
  DShape<T> a, b;
  const T* va;
  const T* vb;
  HullGraph<T> ga, gb;
  mutable int hint0, hint1;  // the vertex each hull's last support call returned (hill-climbing start; -1: warm start)
  MDiff<T> md;
  int lig;
  __device__ __forceinline__ V3<T> one(const DShape<T>& s, const T* v, const HullGraph<T>& g, const V3<T>& d, int& hint) const {
    return s.kind == K_CONVEX ? large_hull_support<T, LARGE_W>(v, s.num_points, g, d, lig, hint) : prim_support(s, d);
  }
  __device__ __forceinline__ void operator()(const V3<T>& dir, V3<T>& w, V3<T>& w0) const {
    w0 = one(a, va, ga, dir, hint0);
    const V3<T> d1 = md.identity ? -dir : -tmul(md.oR1, dir);
    V3<T> s1 = one(b, vb, gb, d1, hint1);
    s1 = md.identity ? s1 : (mul(md.oR1, s1) + md.ot1);
    w = w0 - s1;
  }
};

template <typename T, bool BVG>
// (two waves per SIMD: the compiler's own fp64 allocation is 260 registers -- one wave -- for 20 B of scratch less;
// k_gjk_large<double> 1.05 -> 0.60 ms per 100k 64-vertex pairs, 6.0 -> 3.4 ms at 1024 vertices)
__global__ void __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(2, 8))) k_gjk_large(Work wk, LibView<T> lib, IO<T> io, QParams<T> q) {
  HFCL_GJK_W0_SLAB(T, 256, ps);
  const uint32_t cnt = wk.counts[B_LARGE];
  const int lig = threadIdx.x & (LARGE_W - 1);
  const uint32_t groups = (gridDim.x * blockDim.x) / LARGE_W;
  for (uint32_t it = (blockIdx.x * blockDim.x + threadIdx.x) / LARGE_W; it < cnt; it += groups) {
    const uint32_t pair = wk.lists[size_t(B_LARGE) * wk.n + it];
    LargeSupport<T> sup;
    const uint32_t sid1 = wk.shape1[pair], sid2 = wk.shape2[pair];
    sup.a = lib.shapes[sid1];
    sup.b = lib.shapes[sid2];
    sup.va = lib.verts + 3 * size_t(sup.a.vertex_offset);
    sup.vb = lib.verts + 3 * size_t(sup.b.vertex_offset);
    sup.ga = sup.a.kind == K_CONVEX ? hull_graph(lib, sid1, sup.a.num_points) : HullGraph<T>{nullptr, nullptr};
    sup.gb = sup.b.kind == K_CONVEX ? hull_graph(lib, sid2, sup.b.num_points) : HullGraph<T>{nullptr, nullptr};
    sup.hint0 = sup.hint1 = -1;
    sup.lig = lig;
    const Pose<T> tf1 = load_pose(io.tf1, pair), tf2 = load_pose(io.tf2, pair);
    sup.md = make_mdiff(tf1, tf2);
    const T r0 = swept_radius(sup.a), r1 = swept_radius(sup.b);
    const V3<T> guess0 = initial_guess<T>(io, q, pair);
    Gjk<T, typename GjkW0<T, 256>::P> g;
    if constexpr (BVG)
      gjk_run(g, q.gjk, start_guess(q, sup.a, sup.b, sup.md, guess0), r0 + r1, sup.a.kind == K_CONVEX && sup.b.kind == K_CONVEX, sup, ps);
    else
      gjk_run(g, q.gjk, guess0, r0 + r1, sup.a.kind == K_CONVEX && sup.b.kind == K_CONVEX, sup, ps);
    finish_gjk<T>(g, wk, io, q, pair, tf1, r0, r1, guess0, lig == 0, ps, true);
  }
}

// ---------------------------------------------------------------------------------------
// -inf security margin: cleared results, nothing computed (src/collision.cpp:73-76)
// ---------------------------------------------------------------------------------------
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
#if HFCL_UNIT_F64
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

#endif
// ---------------------------------------------------------------------------------------
// compact host poses (quaternion w,x,y,z + translation, 7 doubles) -> the 12-double Transform3f image the fp64
// kernels read (column-major R, then T).  Runs on the device so that only 56 bytes per pose cross the host link.
// ---------------------------------------------------------------------------------------
#if HFCL_UNIT_F64
__global__ void __launch_bounds__(256) k_expand_poses(const double* qt, double* tf, uint32_t n) {
  for (uint32_t i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) {
    const Pose<double> p = pose_from_quat<double, double>(qt + 7 * size_t(i));
    double* o = tf + 12 * size_t(i);
    o[0] = p.R.r0.x; o[1] = p.R.r1.x; o[2] = p.R.r2.x;
    o[3] = p.R.r0.y; o[4] = p.R.r1.y; o[5] = p.R.r2.y;
    o[6] = p.R.r0.z; o[7] = p.R.r1.z; o[8] = p.R.r2.z;
    o[9] = p.t.x; o[10] = p.t.y; o[11] = p.t.z;
  }
}
#endif
#if HFCL_UNIT_F64
void launch_expand_poses(hipStream_t st, const double* qt, double* tf, uint32_t n) {
  hipLaunchKernelGGL(k_expand_poses, dim3(1024), dim3(256), 0, st, qt, tf, n);
}
#endif

// =======================================================================================
// launchers (hfcl_launch.hpp)
// =======================================================================================
#if HFCL_UNIT_F64
void launch_classify(int grid, hipStream_t st, const Work& wk, const uint8_t* kinds, uint32_t n_shapes, bool distance_mode) {
  hipLaunchKernelGGL(k_classify, dim3(grid), dim3(CLS_BLOCK), 0, st, wk, kinds, n_shapes, distance_mode);
}
#endif
template <typename T>
void launch_unsupported(int grid, hipStream_t st, const Work& wk, const IO<T>& io, int bucket) {
  hipLaunchKernelGGL((k_unsupported<T>), dim3(grid), dim3(256), 0, st, wk, io, bucket);
}
#if HFCL_UNIT_F32
template void launch_unsupported<float>(int, hipStream_t, const Work&, const IO<float>&, int);
#endif
#if HFCL_UNIT_F64
template void launch_unsupported<double>(int, hipStream_t, const Work&, const IO<double>&, int);
#endif

template <typename T>
void launch_closed(int grid, hipStream_t st, const Work& wk, const LibView<T>& lv, const IO<T>& io, const QParams<T>& q, bool staged) {
#if HFCL_UNIT_F64
  if constexpr (sizeof(T) == 8) {
    if (staged) {
      hipLaunchKernelGGL(k_closed_staged, dim3(grid), dim3(256), 0, st, wk, lv, io, q);
      return;
    }
  }
#endif
  hipLaunchKernelGGL((k_closed<T>), dim3(grid), dim3(256), 0, st, wk, lv, io, q);
}
#if HFCL_UNIT_F32
template void launch_closed<float>(int, hipStream_t, const Work&, const LibView<float>&, const IO<float>&, const QParams<float>&, bool);
#endif
#if HFCL_UNIT_F64
template void launch_closed<double>(int, hipStream_t, const Work&, const LibView<double>&, const IO<double>&, const QParams<double>&, bool);
#endif

template <typename T>
void launch_gjk_prim(int grid, hipStream_t st, const Work& wk, const LibView<T>& lv, const IO<T>& io, const QParams<T>& q, bool bvg) {
  if (bvg)
    hipLaunchKernelGGL((k_gjk_prim<T, true>), dim3(grid), dim3(256), 0, st, wk, lv, io, q);
  else
    hipLaunchKernelGGL((k_gjk_prim<T, false>), dim3(grid), dim3(256), 0, st, wk, lv, io, q);
}
#if HFCL_UNIT_F32
template void launch_gjk_prim<float>(int, hipStream_t, const Work&, const LibView<float>&, const IO<float>&, const QParams<float>&, bool);
#endif
#if HFCL_UNIT_F64
template void launch_gjk_prim<double>(int, hipStream_t, const Work&, const LibView<double>&, const IO<double>&, const QParams<double>&, bool);
#endif

template <typename T, int M>
static void launch_gjk_cvx_m(int w, bool bvg, int grid, hipStream_t st, const Work& wk, const LibView<T>& lv, const IO<T>& io, const QParams<T>& q) {
  if (bvg) {  // instantiated for the widths in use only
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
void launch_gjk_cvx(int m, int w, bool bvg, int grid, hipStream_t st, const Work& wk, const LibView<T>& lv, const IO<T>& io, const QParams<T>& q) {
  if (m == 0) launch_gjk_cvx_m<T, 0>(w, bvg, grid, st, wk, lv, io, q);
  else if (m == 1) launch_gjk_cvx_m<T, 1>(w, bvg, grid, st, wk, lv, io, q);
  else launch_gjk_cvx_m<T, 2>(w, bvg, grid, st, wk, lv, io, q);
}
#if HFCL_UNIT_F32
template void launch_gjk_cvx<float>(int, int, bool, int, hipStream_t, const Work&, const LibView<float>&, const IO<float>&, const QParams<float>&);
#endif
#if HFCL_UNIT_F64
template void launch_gjk_cvx<double>(int, int, bool, int, hipStream_t, const Work&, const LibView<double>&, const IO<double>&, const QParams<double>&);
#endif

template <typename T>
void launch_gjk_large(int grid, hipStream_t st, const Work& wk, const LibView<T>& lv, const IO<T>& io, const QParams<T>& q, bool bvg) {
  if (bvg)
    hipLaunchKernelGGL((k_gjk_large<T, true>), dim3(grid), dim3(256), 0, st, wk, lv, io, q);
  else
    hipLaunchKernelGGL((k_gjk_large<T, false>), dim3(grid), dim3(256), 0, st, wk, lv, io, q);
}
#if HFCL_UNIT_F32
template void launch_gjk_large<float>(int, hipStream_t, const Work&, const LibView<float>&, const IO<float>&, const QParams<float>&, bool);
#endif
#if HFCL_UNIT_F64
template void launch_gjk_large<double>(int, hipStream_t, const Work&, const LibView<double>&, const IO<double>&, const QParams<double>&, bool);
#endif

#if HFCL_UNIT_F64
void launch_fill_skipped(hipStream_t st, hfcl_result* out, uint32_t n) {
  hipLaunchKernelGGL((k_fill_skipped<hfcl_result>), dim3(1024), dim3(256), 0, st, out, n);
}
#endif
#if HFCL_UNIT_F32
void launch_fill_skipped(hipStream_t st, hfcl_result_f32* out, uint32_t n) {
  hipLaunchKernelGGL((k_fill_skipped<hfcl_result_f32>), dim3(1024), dim3(256), 0, st, out, n);
}
#endif
