// hfcl_k_epa.hip -- the EPA kernels: tier 1 (8 polytopes per wave in small LDS blocks; batch and streaming form)
// and tier 2 (reference capacity, continues the polytopes tier 1 saved).
//
// Occupancy is what these kernels live on.  profiles/r02_a_valu_issue_peak.txt: one wave issues a VALU instruction every
// ~5 clocks at best and waits ~76 clocks for every dependent LDS round trip of the silhouette walk, so the kernel time is
// the serial time of its waves divided by the number of resident waves (1 wave per SIMD instead of 2: 1.74 -> 3.16 ms,
// profiles/r02_c).  Resident waves are bounded by the LDS block (8 polytopes) and by the registers of a wave; the
// convex x convex form of the streaming tier is built to fit three waves per SIMD on both counts.
#include <algorithm>

#include "hfcl_dev.hpp"
#include "hfcl_launch.hpp"

// Compiled twice, per precision (hfcl_k_epa32.o / hfcl_k_epa64.o; see hfcl_k_gjk.hip): the fp64 tiers without contraction
// (the reference's arithmetic: iteration counts and statuses are the oracle's in every record), the fp32 tiers with the
// contraction the source spells out (-ffp-contract=on).
#ifndef HFCL_UNIT_PRECISION
#define HFCL_UNIT_PRECISION 0
#endif
#define HFCL_UNIT_F32 (HFCL_UNIT_PRECISION != 64)
#define HFCL_UNIT_F64 (HFCL_UNIT_PRECISION != 32)

// LARGE: hulls of more than HULL_MAX vertices may occur (scanned from memory); only the
// full-capacity tier is built that way, so the fast tier keeps its register budget.
template <typename T, int WE, bool LARGE>
struct EpaSupport {  // any pair kind, evaluated by one lane group
  DShape<T> a, b;
  HullRegs<T, WE> h0, h1;
  const T* va;
  const T* vb;
  HullGraph<T> ga, gb;       // LARGE only: adjacency of a hull beyond HULL_MAX vertices (climb_support), if registered
  mutable int hint0, hint1;  // ... and where its last support call ended
  MDiff<T> md;
  int lig;
  __device__ __forceinline__ V3<T> hull(const DShape<T>& s, const HullRegs<T, WE>& h, const T* v, const HullGraph<T>& g, const V3<T>& d, int& hint) const {
    if (LARGE && s.num_points > uint32_t(HULL_MAX)) return large_hull_support<T, WE>(v, s.num_points, g, d, lig, hint);
    return h.support(d, lig);
  }
  __device__ __forceinline__ void operator()(const V3<T>& dir, V3<T>& w, V3<T>& w0) const {
    if (a.kind == K_CONVEX)
      w0 = hull(a, h0, va, ga, dir, hint0);
    else
      w0 = prim_support(a, dir);
    const V3<T> d1 = md.identity ? -dir : -tmul(md.oR1, dir);
    V3<T> s1;
    if (b.kind == K_CONVEX)
      s1 = hull(b, h1, vb, gb, d1, hint1);
    else
      s1 = prim_support(b, d1);
    s1 = md.identity ? s1 : (mul(md.oR1, s1) + md.ot1);
    w = w0 - s1;
  }
};

// Two hulls of at most HULL_MAX vertices: nothing but the register hulls and the relative pose is live across a trip,
// and the support names the shape-0 vertex it returns (the polytope keeps that tag instead of the point, V0_TAG).
template <typename T, int WE>
struct EpaSupportCC {
  HullRegs<T, WE> h0, h1;
  MDiff<T> md;
  int lig;
  __device__ __forceinline__ void operator()(const V3<T>& dir, V3<T>& w, V3<T>& w0, int& tag) const {
    w0 = h0.support(dir, lig, &tag);
    const V3<T> d1 = md.identity ? -dir : -tmul(md.oR1, dir);
    V3<T> s1 = h1.support(d1, lig);
    s1 = md.identity ? s1 : (mul(md.oR1, s1) + md.ot1);
    w = w0 - s1;
  }
};
// tag -> shape-0 support point (Epa::v0r): a vertex of shape 0's hull, or vertex -1-tag of GJK's final simplex, which
// is still in the queue entry
template <typename T>
struct SeedTags {
  const T* va;
  const EpaSeed<T>* ip;
  __device__ __forceinline__ V3<T> operator()(int tag) const {
    if (tag >= 0) return mk<T>(va[3 * tag], va[3 * tag + 1], va[3 * tag + 2]);
    return ip->w0[-1 - tag];
  }
};

template <typename T, int CAP>
__device__ __forceinline__ EpaSaved<T, CAP>* resume_slot(const Work& wk, uint32_t slot) {
  static_assert(sizeof(EpaSaved<T, CAP>) <= epa_resume_stride<T>, "hand-over slots hold any fast tier's polytope");
  return reinterpret_cast<EpaSaved<T, CAP>*>(reinterpret_cast<char*>(wk.epa_resume) + size_t(slot) * epa_resume_stride<T>);
}

// TIER: 1 reads queue 1 and may push to queue 2; 2 reads queue 2 (never overflows: CAP = 64)
// QSEL (tier 1): 0 = both queues of epa_queue one after the other, 1 = the bottom queue only, 2 = the top queue only
// (fp64: polytope pairs / pairs with a curved shape, finish_gjk).
// Waves per SIMD the compiler allocates registers for: fp64 tier 1 gets two when its LDS block admits two (16 of the 128
// units per wave), one otherwise (more registers, no spills, where the LDS would not admit a second wave anyway).
template <typename T, int WE, int CAP, int TIER>
constexpr int epa_waves_per_simd = sizeof(T) == 4 ? (TIER == 1 ? HFCL_WPE_EPA32 : 2)
                                                  : (sizeof(EpaScratch<T, CAP, TIER == 1 ? V0_BLOCK : V0_EXTERN>) * (64 / WE) <= 16 * 1280 ? 2 : HFCL_WPE_EPA64);
template <typename T, int WE, int CAP, int TIER, int QSEL = 0>
__global__ void __launch_bounds__(64) __attribute__((amdgpu_waves_per_eu(epa_waves_per_simd<T, WE, CAP, TIER>, 8)))
k_epa(Work wk, LibView<T> lib, IO<T> io, QParams<T> q) {
  constexpr int G = 64 / WE;
  // the full-capacity tier keeps the shape-0 support points in global memory: its LDS block bounds the
  // occupancy (fp64: 45.5 KB -> 3 waves per CU with them, 36.5 KB -> 4 without)
  constexpr int V0M = TIER == 1 ? V0_BLOCK : V0_EXTERN;
  __shared__ EpaScratch<T, CAP, V0M> scratch[G];
  Quad<T>* const v0_ext = TIER == 1 ? nullptr : reinterpret_cast<Quad<T>*>(wk.epa_v0) + size_t(blockIdx.x * G + threadIdx.x / WE) * (CAP + 4);
  // tier 1 walks both queues of epa_queue one after the other: the polytope pairs (slots 0 upwards), then the pairs with a
  // curved shape (slots n-1 downwards; finish_gjk) -- the G polytopes a wave steps in lockstep are of one class
  const uint32_t cnt0 = (TIER == 1 && QSEL == 2) ? 0u : wk.counts[TIER == 1 ? B_COUNT : B_COUNT + 1];
  const uint32_t cnt = cnt0 + (TIER == 1 && QSEL != 1 ? wk.counts[B_COUNT + 3] : 0u);
  const int lane = threadIdx.x & 63, grp = lane / WE, lig = lane & (WE - 1);
  const uint32_t groups = gridDim.x * G;
  const EpaItem<T>* queue = reinterpret_cast<const EpaItem<T>*>(TIER == 1 ? wk.epa_queue : wk.epa_queue2);
  for (uint32_t it = blockIdx.x * G + grp; it < cnt; it += groups) {
    // the seed stays in memory and is read where it is used (as a local copy it is spilled across the hull loads)
    const EpaItem<T>& item = queue[it < cnt0 ? it : wk.n - 1u - (it - cnt0)];
    const uint32_t pair = item.pair;
    EpaSupport<T, WE, TIER == 2> sup;
    const uint32_t sid1 = wk.shape1[pair], sid2 = wk.shape2[pair];
    sup.a = lib.shapes[sid1];
    sup.b = lib.shapes[sid2];
    sup.lig = lig;
    const T* va = lib.verts + 3 * size_t(sup.a.vertex_offset);
    const T* vb = lib.verts + 3 * size_t(sup.b.vertex_offset);
    if (TIER == 2) {
      sup.va = va;
      sup.vb = vb;
      sup.ga = sup.a.kind == K_CONVEX ? hull_graph(lib, sid1, sup.a.num_points) : HullGraph<T>{nullptr, nullptr};
      sup.gb = sup.b.kind == K_CONVEX ? hull_graph(lib, sid2, sup.b.num_points) : HullGraph<T>{nullptr, nullptr};
      sup.hint0 = sup.hint1 = -1;
    }
    if (sup.a.kind == K_CONVEX && (TIER != 2 || sup.a.num_points <= uint32_t(HULL_MAX))) sup.h0.load(va, sup.a.num_points, lig);
    if (sup.b.kind == K_CONVEX && (TIER != 2 || sup.b.num_points <= uint32_t(HULL_MAX))) sup.h1.load(vb, sup.b.num_points, lig);
    sup.md = make_mdiff(load_pose(io.tf1, pair), load_pose(io.tf2, pair));
    // (the pose of shape 1 is read again when the record is written: 12 scalars less across the expansion)
    auto tf1 = [&]() { return load_pose(io.tf1, pair); };
    const T r0 = swept_radius(sup.a), r1 = swept_radius(sup.b);
    PairOut<T> o;
    int rc = 1;
    if constexpr (TIER == 2) {
      if (item.rank & EPA_RESUME_FLAG) {  // continue what the fast tier saved for this slot (the seed's rank is not used)
        if (epa_small_cap<T> != epa_fast_cap<T> && (item.rank & EPA_RESUME_SMALL))
          epa_resume<T, LaneGroup<WE>, epa_small_cap<T>, CAP>(&scratch[grp], resume_slot<T, epa_small_cap<T>>(wk, it), item, q, tf1, r0, r1,
                                                            sup, o, v0_ext);
        else
          epa_resume<T, LaneGroup<WE>, epa_fast_cap<T>, CAP>(&scratch[grp], resume_slot<T, epa_fast_cap<T>>(wk, it), item, q, tf1, r0, r1,
                                                           sup, o, v0_ext);
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
        if (save) epa_save_block<T, LaneGroup<WE>, CAP>(&scratch[grp], resume_slot<T, CAP>(wk, slot));
        if (lig == 0) {  // queue to queue, no local copy (a local EpaItem lives in scratch memory)
          EpaItem<T>* dst = reinterpret_cast<EpaItem<T>*>(wk.epa_queue2) + slot;
          *dst = item;
          if (save) dst->rank = item.rank | EPA_RESUME_FLAG | (CAP != epa_fast_cap<T> ? EPA_RESUME_SMALL : 0);
        }
      }
    }
    LaneGroup<WE>::sync();
  }
}

// Fast tier as a stream: the 64/WE lane groups of a wave walk through the wave's share of the queue and a
// group that is done does not wait for the slowest polytope of the wave (EPA runs 1 .. CAP trips per
// polytope, mean ~7 on convex pairs: in lockstep batches of 8 only ~56 % of the trips are useful).
// The wave alternates between two uniform phases:
//   trip   : every group with a live polytope does one expansion step (Epa::step);
//   refill : once at least EPA_REFILL_MIN groups are without one (or none is live), those groups write
//            the record of the polytope they finished (or hand it over to the full-capacity tier) and
//            start the next item of the wave: seed, hulls, encloseOrigin, first tetrahedron (Epa::begin).
// Batching the refills matters: a refill costs about 1.5 trips of the whole wave whoever takes part.
//
// CC = true: the convex x convex queue (the top end of epa_queue, slots n-1 downwards; filled by k_gjk_cvx<.,0>).  Both
// shapes are hulls of <= 32 vertices, which lets the wave drop everything a trip does not need: no shape records,
// no shape-0 support points (V0_TAG), pose and radii re-read when the record is written.  That and the 1.6 KB block
// put three waves on a SIMD (two in the general form).
#ifndef HFCL_EPA_REFILL_MIN
#define HFCL_EPA_REFILL_MIN 3
#endif
#ifndef HFCL_WPE_EPA32_CC
#define HFCL_WPE_EPA32_CC 3
#endif
// TOPQ: the queue at the top end of epa_queue (fp32: convex x convex, fp64: pairs with a curved shape)
template <typename T, int WE, int CAP, bool CC, bool TOPQ = CC>
__global__ void __launch_bounds__(64) __attribute__((amdgpu_waves_per_eu(sizeof(T) == 8 ? epa_waves_per_simd<T, WE, CAP, 1> : (CC ? HFCL_WPE_EPA32_CC : HFCL_WPE_EPA32), 8)))
k_epa_stream(Work wk, LibView<T> lib, IO<T> io, QParams<T> q) {
  constexpr int G = 64 / WE;
  constexpr int V0M = CC ? V0_TAG : V0_BLOCK;
  typedef LaneGroup<WE> Grp;
  __shared__ EpaScratch<T, CAP, V0M> scratch[G];
#ifdef HFCL_EPA_PAD_LDS  // occupancy experiment: extra LDS per wave lowers the number of resident waves
  __shared__ uint32_t lds_pad[HFCL_EPA_PAD_LDS / 4];
  if (wk.n == 0xFFFFFFFFu) lds_pad[threadIdx.x] = wk.n;
#endif
  const uint32_t cnt = wk.counts[TOPQ ? B_COUNT + 3 : B_COUNT];
  const int lane = threadIdx.x & 63, grp = lane / WE, lig = lane & (WE - 1);
  const EpaItem<T>* const queue = reinterpret_cast<const EpaItem<T>*>(wk.epa_queue);
  // item i of this kernel's queue
  auto item_ptr = [&](uint32_t i) { return TOPQ ? queue + (wk.n - 1u - i) : queue + i; };
  enum { IDLE = 0, LIVE = 1, DONE = 2, HANDOVER = 3 };
  int state = IDLE;
  uint32_t it = 0;              // queue slot of this group's polytope
  uint32_t next = blockIdx.x;   // wave-uniform: the wave's items are next, next + gridDim.x, ...
  typename std::conditional<CC, EpaSupportCC<T, WE>, EpaSupport<T, WE, false>>::type sup;
  sup.lig = lig;
  Epa<T, Grp, CAP, V0M> epa;
  EpaLoop<T> L;
  // what the record needs beyond the polytope is read again when it is written (a pose and two radii held across the
  // trips are 14 registers of every lane)
  auto finish = [&](const EpaResult<T>& res, const EpaItem<T>* ip) {
    const uint32_t pair = ip->pair;
    const Pose<T> tf1 = load_pose(io.tf1, pair);
    const T r0 = swept_radius(lib.shapes[wk.shape1[pair]]), r1 = swept_radius(lib.shapes[wk.shape2[pair]]);
    PairOut<T> o;
    epa_finish(res, ip->gjk_iters, tf1, r0, r1, o);
    if (lig == 0) {
      write_out<T>(io, q, pair, o);
      write_guess<T>(io, pair, o.cached_guess, 0, 0);
    }
  };
  auto tags_of = [&](const EpaItem<T>* ip) {
    return SeedTags<T>{lib.verts + 3 * size_t(lib.shapes[wk.shape1[ip->pair]].vertex_offset), ip};
  };
  while (true) {
    const uint64_t live = __ballot(state == LIVE);
    const int n_live = __popcll(live) / WE;
    const bool more = next < cnt;
    if (n_live == 0 || (more && G - n_live >= HFCL_EPA_REFILL_MIN)) {
      // ---- refill phase (uniform decision; groups with a live polytope sit it out) ----
      if (state != LIVE) {
        if (state != IDLE) {
          const EpaItem<T>* ip = item_ptr(it);
          if (state == DONE) {
            const uint32_t pair = ip->pair;
            const T ssr = swept_radius(lib.shapes[wk.shape1[pair]]) + swept_radius(lib.shapes[wk.shape2[pair]]);
            EpaResult<T> res;
            if constexpr (CC)
              epa.loop_result(L, ssr, res, tags_of(ip));
            else
              epa.loop_result(L, ssr, res);
            finish(res, ip);
          } else {  // hand over to the full-capacity tier: the seed and, room permitting, the polytope itself
            uint32_t slot = 0;
            if (lig == 0) slot = atomicAdd(&wk.counts[B_COUNT + 1], 1u);
            slot = __shfl(slot, 0, WE);
            const bool save = epa.resumable && slot < wk.resume_cap;
            if (save) {
              if constexpr (CC)
                epa_save_block<T, Grp, CAP>(&scratch[grp], resume_slot<T, CAP>(wk, slot), tags_of(ip));
              else
                epa_save_block<T, Grp, CAP>(&scratch[grp], resume_slot<T, CAP>(wk, slot));
            }
            if (lig == 0) {  // queue to queue, no local copy (a local EpaItem lives in scratch memory)
              EpaItem<T>* dst = reinterpret_cast<EpaItem<T>*>(wk.epa_queue2) + slot;
              *dst = *ip;
              if (save) dst->rank = ip->rank | EPA_RESUME_FLAG | (CAP != epa_fast_cap<T> ? EPA_RESUME_SMALL : 0);
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
          const EpaItem<T>* ip = item_ptr(it);
          const uint32_t pair = ip->pair;
          {
            const DShape<T> a = lib.shapes[wk.shape1[pair]], b = lib.shapes[wk.shape2[pair]];
            if (CC || a.kind == K_CONVEX) sup.h0.load(lib.verts + 3 * size_t(a.vertex_offset), a.num_points, lig);
            if (CC || b.kind == K_CONVEX) sup.h1.load(lib.verts + 3 * size_t(b.vertex_offset), b.num_points, lig);
            if constexpr (!CC) {
              sup.a = a;
              sup.b = b;
            }
            sup.md = make_mdiff(load_pose(io.tf1, pair), load_pose(io.tf2, pair));
          }
          epa.reset(&scratch[grp], q.epa_max_iterations, q.epa_tolerance);
          epa.set_vert(0, ip->w[0], ip->w0[0], -1);
          epa.set_vert(1, ip->w[1], ip->w0[1], -2);
          epa.set_vert(2, ip->w[2], ip->w0[2], -3);
          epa.set_vert(3, ip->w[3], ip->w0[3], -4);
          Grp::sync();
          EpaResult<T> res;
          int closest0;
          if constexpr (CC)
            closest0 = epa.begin(ip->rank, -ip->guess, sup, res, tags_of(ip));
          else
            closest0 = epa.begin(ip->rank, -ip->guess, sup, res);
          if (closest0 != EPA_NULL) {
            epa.loop_enter(L, closest0, 0, 0);
            state = LIVE;
          } else if (epa.overflow) {
            state = HANDOVER;  // (a block too small for the first tetrahedron: not with CAP >= 1)
          } else {  // FallBack: final without a loop
            finish(res, ip);
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

// =======================================================================================
// The convex x convex fast tier in three stages (hfcl_epa.hpp: EpaReady).  k_epa_stream<.., CC> spends a quarter of its
// instructions in its refills -- seed, poses, encloseOrigin, first tetrahedron, closest face; witness points and the record of
// the polytope that ended -- with three or four of a wave's eight groups taking part and every lane of a group doing the same
// serial work (profiles/r05_a).  Here that work is done by one lane per polytope in two kernels around the loop:
//   k_epa_prepare  item of the convex x convex queue -> EpaReady block (compacted: fall-backs get their record here, the
//                  two-in-a-hundred-thousand seeds of rank < 4 go to the full-capacity tier, which starts from any seed);
//   k_epa_loop     the expansion loop alone: a refill is a block copied into LDS and two hulls;
//   k_epa_records  EpaLoopOut -> witness points, normal, record (EPAExtractWitnessPointsAndNormal, narrowphase.h:658-711).
// =======================================================================================
template <typename T>
__global__ void __launch_bounds__(256) k_epa_prepare(Work wk, LibView<T> lib, IO<T> io, QParams<T> q) {
  const uint32_t cnt = wk.counts[B_COUNT + 3];
  const EpaItem<T>* const queue = reinterpret_cast<const EpaItem<T>*>(wk.epa_queue);
  EpaReady<T>* const ready = reinterpret_cast<EpaReady<T>*>(wk.epa_ready);
  // Block i belongs to item i of the queue: no compaction (reserving a wave's slots with an atomicAdd on one counter is ~20 ns per wave
  // and trip, one after the other: 4 600 of them were 0.05 of this kernel's 0.07 ms, profiles/r05_b).  The few items without a loop --
  // fall-backs, seeds of rank < 4: a few in 100 000 -- leave a block marked EPA_READY_NONE that k_epa_loop passes over.
  for (uint32_t i = blockIdx.x * blockDim.x + threadIdx.x; i < cnt; i += gridDim.x * blockDim.x) {
    bool live = false;
    EpaReady<T> rb;
    {
      const EpaItem<T>* ip = queue + (wk.n - 1u - i);
      const int rank = ip->rank;
      if (rank != 4) {
        // encloseOrigin has supports to evaluate: the full-capacity tier starts from the seed
        const uint32_t slot = atomicAdd(&wk.counts[B_COUNT + 1], 1u);
        reinterpret_cast<EpaItem<T>*>(wk.epa_queue2)[slot] = *ip;
      } else {
        const V3<T> w[4] = {ip->w[0], ip->w[1], ip->w[2], ip->w[3]};
        int flags[4], closest = 0;
        live = epa_prepare_tetrahedron(w, q.epa_tolerance, rb.vw, rb.fn, flags, closest);
        const uint32_t pair = ip->pair;
        if (live) {
          const DShape<T> a = lib.shapes[wk.shape1[pair]], b = lib.shapes[wk.shape2[pair]];
          const MDiff<T> md = make_mdiff(load_pose(io.tf1, pair), load_pose(io.tf2, pair));
          rb.seed = i;
          rb.voff_a = a.vertex_offset;
          rb.voff_b = b.vertex_offset;
          rb.packed = (a.num_points & 63u) | ((b.num_points & 63u) << 6) | (uint32_t(closest) << 12) | (uint32_t((flags[0] >> 1) & 1) << 14) |
                      (uint32_t((flags[1] >> 1) & 1) << 15) | (uint32_t((flags[2] >> 1) & 1) << 16) | (uint32_t((flags[3] >> 1) & 1) << 17) |
                      (md.identity ? 1u << 18 : 0u);
          rb.md[0] = md.oR1.r0.x; rb.md[1] = md.oR1.r0.y; rb.md[2] = md.oR1.r0.z;
          rb.md[3] = md.oR1.r1.x; rb.md[4] = md.oR1.r1.y; rb.md[5] = md.oR1.r1.z;
          rb.md[6] = md.oR1.r2.x; rb.md[7] = md.oR1.r2.y; rb.md[8] = md.oR1.r2.z;
          rb.md[9] = md.ot1.x; rb.md[10] = md.ot1.y; rb.md[11] = md.ot1.z;
          rb.state = EPA_READY_PENDING;
          rb.pair = pair;
          rb.gjk_iters = ip->gjk_iters;
          rb.pad_ = 0u;
        } else {  // FallBack (:1299-1315): final without a loop
          EpaResult<T> res;
          res.status = EPA_FALLBACK;
          res.iterations = 0;
          PairOut<T> o;
          epa_finish(res, ip->gjk_iters, load_pose(io.tf1, pair), T(0), T(0), o);
          write_out<T>(io, q, pair, o);
          write_guess<T>(io, pair, o.cached_guess, 0, 0);
        }
      }
    }
    if (live) {
      ready[i] = rb;
    } else {
      ready[i].packed = 0u;  // (no vertices: never a live block)
      ready[i].state = EPA_READY_NONE;
    }
  }
}

#ifndef HFCL_EPA_LOOP_REFILL_MIN
#define HFCL_EPA_LOOP_REFILL_MIN 2  // a refill is cheap here: idle groups wait for fewer companions than in k_epa_stream (1: the same, 3: 4 % slower; profiles/r05_a)
#endif
#ifndef HFCL_EPA_LOOP_ROUNDS
#define HFCL_EPA_LOOP_ROUNDS 1  // persistent grid = this many times the resident waves (1 / 2 / 3: 0.979 / 0.994 / 1.016 ms; profiles/r05_a)
#endif
#ifndef HFCL_EPA_CC_RESUME
#define HFCL_EPA_CC_RESUME 1  // polytopes that outgrow the block are continued by k_epa_resume_cc (0: by the general full-capacity tier)
#endif
// the part of the hand-over area the convex x convex polytopes of k_epa_loop are saved in (EpaSaved<T, EPA_FAST_CAP> blocks, tags kept)
template <typename T>
__device__ __forceinline__ EpaSaved<T, EPA_FAST_CAP>* cc_resume_slot(const Work& wk, uint32_t slot) {
  return resume_slot<T, EPA_FAST_CAP>(wk, wk.cc_resume_base + slot);
}
template <typename T, int WE, int CAP>
__global__ void __launch_bounds__(64) __attribute__((amdgpu_waves_per_eu(HFCL_WPE_EPA32_CC, 8)))
k_epa_loop(Work wk, LibView<T> lib, QParams<T> q) {
  constexpr int G = 64 / WE;
  static_assert(WE >= 8, "a group's lanes copy the eight records of a block");
  typedef LaneGroup<WE> Grp;
  __shared__ EpaScratch<T, CAP, V0_TAG> scratch[G];
  const uint32_t cnt = wk.counts[B_COUNT + 3];  // one block per item of the convex x convex queue
  const int lane = threadIdx.x & 63, grp = lane / WE, lig = lane & (WE - 1);
  EpaReady<T>* const ready = reinterpret_cast<EpaReady<T>*>(wk.epa_ready);
  const EpaItem<T>* const queue = reinterpret_cast<const EpaItem<T>*>(wk.epa_queue);
  enum { IDLE = 0, LIVE = 1, DONE = 2, HANDOVER = 3 };
  int state = IDLE;
  uint32_t it = 0;             // block of this group's polytope
  uint32_t next = blockIdx.x;  // wave-uniform: the wave's blocks are next, next + gridDim.x, ...
  EpaSupportCC<T, WE> sup;
  sup.lig = lig;
  Epa<T, Grp, CAP, V0_TAG> epa;
  EpaLoop<T> L;
  while (true) {
    const uint64_t live = __ballot(state == LIVE);
    const int n_live = __popcll(live) / WE;
    const bool more = next < cnt;
    if (n_live == 0 || (more && G - n_live >= HFCL_EPA_LOOP_REFILL_MIN)) {
      // ---- refill phase (uniform decision; groups with a live polytope sit it out) ----
      if (state != LIVE) {
        if (state != IDLE) {
          EpaReady<T>* rb = ready + it;
          if (state == DONE) {
            EpaLoopOut<T> o;
            epa.loop_out(L, o);
            if (lig == 0) {
              *reinterpret_cast<EpaLoopOut<T>*>(rb->md) = o;
              rb->state = EPA_READY_DONE;
            }
          } else {
            // the polytope outgrew the block: saved as it is (tags kept) for k_epa_resume_cc; without room in the hand-over area,
            // or when it cannot be continued, its seed goes to the general full-capacity tier
            uint32_t slot = 0xFFFFFFFFu;
            if (HFCL_EPA_CC_RESUME && epa.resumable) {
              if (lig == 0) slot = atomicAdd(&wk.counts[CTR_EPA_CC_OVER], 1u);
              slot = __shfl(slot, 0, WE);
            }
            if (slot < wk.cc_resume_cap) {  // (slots past the area are never read: k_epa_resume_cc clamps the count)
              epa_save_block<T, Grp, CAP, V0_TAG, NoTags, true>(&scratch[grp], cc_resume_slot<T>(wk, slot));
              if (lig == 0) wk.epa_cc_over[slot] = it;
            } else {
              if (lig == 0) {  // queue to queue, no local copy
                const EpaItem<T>* ip = queue + (wk.n - 1u - rb->seed);
                const uint32_t s2 = atomicAdd(&wk.counts[B_COUNT + 1], 1u);
                reinterpret_cast<EpaItem<T>*>(wk.epa_queue2)[s2] = *ip;
              }
            }
            if (lig == 0) rb->state = EPA_READY_HANDED_OVER;
          }
          Grp::sync();
          state = IDLE;
        }
        // rank of this group among the groups taking part, in lane order
        const uint64_t lower = live | ~((uint64_t(1) << (grp * WE)) - 1);  // live lanes and lanes >= mine do not count
        const uint32_t rank = uint32_t(__popcll(~lower)) / WE;
        it = next + rank * gridDim.x;
        if (it < cnt) {
          const EpaReady<T>* rb = ready + it;
          const uint32_t packed = rb->packed;
          if ((packed & 63u) != 0u) {  // (0 vertices: EPA_READY_NONE, k_epa_prepare wrote the record or queued the seed elsewhere)
          sup.h0.load(lib.verts + 3 * size_t(rb->voff_a), packed & 63u, lig);
          sup.h1.load(lib.verts + 3 * size_t(rb->voff_b), (packed >> 6) & 63u, lig);
          sup.md.oR1.r0 = mk<T>(rb->md[0], rb->md[1], rb->md[2]);
          sup.md.oR1.r1 = mk<T>(rb->md[3], rb->md[4], rb->md[5]);
          sup.md.oR1.r2 = mk<T>(rb->md[6], rb->md[7], rb->md[8]);
          sup.md.ot1 = mk<T>(rb->md[9], rb->md[10], rb->md[11]);
          sup.md.identity = (packed >> 18) & 1u;
          epa.reset(&scratch[grp], q.epa_max_iterations, q.epa_tolerance);
          epa.loop_enter(L, epa.install(rb, packed), 0, 0);
          state = LIVE;
          }
        }
      }
      next += uint32_t(G - n_live) * gridDim.x;
      if (n_live == 0 && !more) {
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

// The polytopes k_epa_loop saved, continued in blocks of the reference's capacity by the same code -- tags, register hulls, the
// parallel horizon --, WE lanes each (the general full-capacity tier, k_epa<.., 2>, is built for any pair of kinds, walks the
// silhouette and steps four polytopes of 16 lanes in lockstep: 8 us per iteration where this one takes ~3; the hand-overs are two
// polytopes in a hundred and their kernel is as long as its longest chain of iterations).
template <typename T, int WE>
__global__ void __launch_bounds__(64) k_epa_resume_cc(Work wk, LibView<T> lib, IO<T> io, QParams<T> q) {
  constexpr int G = 64 / WE, CAP = EPA_MAX_ITER;
  typedef LaneGroup<WE> Grp;
  __shared__ EpaScratch<T, CAP, V0_TAG> scratch[G];
  const uint32_t cnt = min(wk.counts[CTR_EPA_CC_OVER], wk.cc_resume_cap);
  const int lane = threadIdx.x & 63, grp = lane / WE, lig = lane & (WE - 1);
  const EpaReady<T>* const ready = reinterpret_cast<const EpaReady<T>*>(wk.epa_ready);
  const EpaItem<T>* const queue = reinterpret_cast<const EpaItem<T>*>(wk.epa_queue);
  for (uint32_t it = blockIdx.x * G + grp; it < cnt; it += gridDim.x * G) {
    const uint32_t idx = wk.epa_cc_over[it];
    if (idx == 0xFFFFFFFFu) continue;
    const EpaReady<T>* rb = ready + idx;
    const uint32_t packed = rb->packed, pair = rb->pair;
    EpaSupportCC<T, WE> sup;
    sup.lig = lig;
    const T* va = lib.verts + 3 * size_t(rb->voff_a);
    sup.h0.load(va, packed & 63u, lig);
    sup.h1.load(lib.verts + 3 * size_t(rb->voff_b), (packed >> 6) & 63u, lig);
    sup.md.oR1.r0 = mk<T>(rb->md[0], rb->md[1], rb->md[2]);
    sup.md.oR1.r1 = mk<T>(rb->md[3], rb->md[4], rb->md[5]);
    sup.md.oR1.r2 = mk<T>(rb->md[6], rb->md[7], rb->md[8]);
    sup.md.ot1 = mk<T>(rb->md[9], rb->md[10], rb->md[11]);
    sup.md.identity = (packed >> 18) & 1u;
    const SeedTags<T> tags{va, queue + (wk.n - 1u - rb->seed)};
    const T r0 = swept_radius(lib.shapes[wk.shape1[pair]]), r1 = swept_radius(lib.shapes[wk.shape2[pair]]);
    Epa<T, Grp, CAP, V0_TAG> epa;
    epa.reset(&scratch[grp], q.epa_max_iterations, q.epa_tolerance);
    const EpaHeader h = epa.template load<EPA_FAST_CAP>(cc_resume_slot<T>(wk, it));
    EpaResult<T> res;
    epa.run_loop(h.closest, h.iterations, h.pass, r0 + r1, sup, res, tags);
    PairOut<T> o;
    epa_finish(res, rb->gjk_iters, load_pose(io.tf1, pair), r0, r1, o);
    if (lig == 0) {
      write_out<T>(io, q, pair, o);
      write_guess<T>(io, pair, o.cached_guess, 0, 0);
    }
    Grp::sync();
  }
}

template <typename T>
__global__ void __launch_bounds__(256) k_epa_records(Work wk, LibView<T> lib, IO<T> io, QParams<T> q) {
  const uint32_t cnt = wk.counts[B_COUNT + 3];
  const EpaReady<T>* const ready = reinterpret_cast<const EpaReady<T>*>(wk.epa_ready);
  const EpaItem<T>* const queue = reinterpret_cast<const EpaItem<T>*>(wk.epa_queue);
  for (uint32_t i = blockIdx.x * blockDim.x + threadIdx.x; i < cnt; i += gridDim.x * blockDim.x) {
    const EpaReady<T>* rb = ready + i;
    if (rb->state != EPA_READY_DONE) continue;  // handed over: the tier that continues the polytope writes the record
    const EpaLoopOut<T>* lo = reinterpret_cast<const EpaLoopOut<T>*>(rb->md);
    const uint32_t pair = rb->pair;
    const SeedTags<T> tags{lib.verts + 3 * size_t(rb->voff_a), queue + (wk.n - 1u - rb->seed)};
    const T r0 = swept_radius(lib.shapes[wk.shape1[pair]]), r1 = swept_radius(lib.shapes[wk.shape2[pair]]);
    EpaResult<T> res;
    res.status = lo->status;
    res.iterations = lo->iterations;
    res.normal = mk<T>(lo->nx, lo->ny, lo->nz);
    res.depth = lo->depth + (r0 + r1);
    res.rw0_ = mk<T>(lo->rw[0], lo->rw[1], lo->rw[2]);
    res.rw1_ = mk<T>(lo->rw[3], lo->rw[4], lo->rw[5]);
    res.rw2_ = mk<T>(lo->rw[6], lo->rw[7], lo->rw[8]);
    res.r00 = tags(lo->tag[0]);
    res.r01 = tags(lo->tag[1]);
    res.r02 = tags(lo->tag[2]);
    PairOut<T> o;
    epa_finish(res, rb->gjk_iters, load_pose(io.tf1, pair), r0, r1, o);
    write_out<T>(io, q, pair, o);
    write_guess<T>(io, pair, o.cached_guess, 0, 0);
  }
}

// =======================================================================================
// The same three stages for pairs of any convex kinds and for both precisions (hfcl_epa.hpp: EpaReadyG).  The lockstep fast tiers
// (k_epa<.., 1, QSEL>) step the 4 or 8 polytopes of a wave until the last of them is through: on cfg5 the curved class runs 24 iterations on
// average and 36 for the slowest of four (66 % useful trips; 77 % with the hand-over at 29), the polytope class 5.7 and 11.2 for the
// slowest of eight (53 %).  The streaming form lost in fp64 because a refill -- seed, shape records, encloseOrigin, tetrahedron; witness
// points and record -- cost more than the idle groups (profiles/r02_q, r02_u, r03_l); here a refill is twelve records copied into LDS.
//   k_epa_prepare_general  both queues of epa_queue (bottom: 0 .. counts[B_COUNT]; top: slots n-1 downwards, counts[B_COUNT + 3]) ->
//                          block i / block n-1-i of epa_ready_g; encloseOrigin (any rank) with the pair's supports, one lane per seed
//   k_epa_loop_general     one queue (QTOP), WE lanes per polytope, blocks for CAP iterations; hands over to the full-capacity tier as
//                          the lockstep kernels do (epa_save_block, queue 2)
//   k_epa_records_general  both queues
// fp64 arithmetic is the reference's (the unit is built without contraction) in every stage: the records are the lockstep form's, byte for byte.
// =======================================================================================
template <typename T>
__device__ __forceinline__ EpaReadyG<T>* ready_g(const Work& wk, bool top, uint32_t i) {
  return reinterpret_cast<EpaReadyG<T>*>(wk.epa_ready_g) + (top ? wk.n - 1u - i : i);
}
// (seeds of rank < 4 -- encloseOrigin would evaluate supports: none in 60 000 polytopes of cfg5 / cfg2 -- go to the full-capacity tier, which
// starts from any seed and reproduces the fast tier's record bit for bit (test_epa_hand_over_equals_restart); the kernel then needs no support
// function and nothing but a seed's eight points in registers)
struct NoSupportNeeded {
  template <typename T>
  __device__ __forceinline__ void operator()(const V3<T>&, V3<T>& w, V3<T>& w0) const {
    w = w0 = mk<T>(T(0), T(0), T(0));
  }
};
template <typename T>
__global__ void __launch_bounds__(256) k_epa_prepare_general(Work wk, LibView<T> lib, IO<T> io, QParams<T> q, int skip_top) {
  const uint32_t cnt0 = wk.counts[B_COUNT], cnt = cnt0 + (skip_top ? 0u : wk.counts[B_COUNT + 3]);
  const EpaItem<T>* const queue = reinterpret_cast<const EpaItem<T>*>(wk.epa_queue);
  for (uint32_t it = blockIdx.x * blockDim.x + threadIdx.x; it < cnt; it += gridDim.x * blockDim.x) {
    const bool top = it >= cnt0;
    const uint32_t i = top ? it - cnt0 : it;
    const EpaItem<T>* ip = top ? queue + (wk.n - 1u - i) : queue + i;
    EpaReadyG<T>* const rbp = ready_g<T>(wk, top, i);
    if (ip->rank != 4) {
      const uint32_t slot = atomicAdd(&wk.counts[B_COUNT + 1], 1u);
      reinterpret_cast<EpaItem<T>*>(wk.epa_queue2)[slot] = *ip;
      rbp->state = EPA_READY_NONE;
      continue;
    }
    const uint32_t pair = ip->pair;
    EpaRegStore<T> st;
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      st.w[k] = ip->w[k];
      st.w0[k] = ip->w0[k];
    }
    int flags[4], closest = 0;
    NoSupportNeeded ns;
    // (the records go straight to the block: held in registers until the end they are 100 of them in fp64)
    if (epa_prepare_general(st, 4, q.epa_tolerance, ns, rbp->vw, rbp->v0, rbp->fn, flags, closest)) {
      const uint32_t sid1 = wk.shape1[pair], sid2 = wk.shape2[pair];
      const MDiff<T> md = make_mdiff(load_pose(io.tf1, pair), load_pose(io.tf2, pair));
      rbp->seed = i | (top ? 0x80000000u : 0u);
      rbp->pair = pair;
      rbp->sid1 = sid1;
      rbp->sid2 = sid2;
      rbp->packed = uint32_t(closest) | (uint32_t((flags[0] >> 1) & 1) << 2) | (uint32_t((flags[1] >> 1) & 1) << 3) | (uint32_t((flags[2] >> 1) & 1) << 4) |
                    (uint32_t((flags[3] >> 1) & 1) << 5) | (md.identity ? 1u << 6 : 0u);
      rbp->gjk_iters = ip->gjk_iters;
      rbp->state = EPA_READY_PENDING;
      rbp->md[0] = md.oR1.r0.x; rbp->md[1] = md.oR1.r0.y; rbp->md[2] = md.oR1.r0.z;
      rbp->md[3] = md.oR1.r1.x; rbp->md[4] = md.oR1.r1.y; rbp->md[5] = md.oR1.r1.z;
      rbp->md[6] = md.oR1.r2.x; rbp->md[7] = md.oR1.r2.y; rbp->md[8] = md.oR1.r2.z;
      rbp->md[9] = md.ot1.x; rbp->md[10] = md.ot1.y; rbp->md[11] = md.ot1.z;
    } else {  // FallBack (:1299-1315): final without a loop
      EpaResult<T> res;
      res.status = EPA_FALLBACK;
      res.iterations = 0;
      PairOut<T> o;
      epa_finish(res, ip->gjk_iters, load_pose(io.tf1, pair), swept_radius(lib.shapes[wk.shape1[pair]]), swept_radius(lib.shapes[wk.shape2[pair]]), o);
      write_out<T>(io, q, pair, o);
      write_guess<T>(io, pair, o.cached_guess, 0, 0);
      rbp->state = EPA_READY_NONE;
    }
  }
}

#ifndef HFCL_EPA_LOOPG_REFILL_MIN
#define HFCL_EPA_LOOPG_REFILL_MIN 1
#endif
template <typename T, int WE, int CAP, bool QTOP>
__global__ void __launch_bounds__(64) __attribute__((amdgpu_waves_per_eu(sizeof(T) == 8 ? epa_waves_per_simd<T, WE, CAP, 1> : HFCL_WPE_EPA32, 8)))
k_epa_loop_general(Work wk, LibView<T> lib, QParams<T> q) {
  constexpr int G = 64 / WE;
  typedef LaneGroup<WE> Grp;
  __shared__ EpaScratch<T, CAP, V0_BLOCK> scratch[G];
  const uint32_t cnt = wk.counts[QTOP ? B_COUNT + 3 : B_COUNT];
  const int lane = threadIdx.x & 63, grp = lane / WE, lig = lane & (WE - 1);
  const EpaItem<T>* const queue = reinterpret_cast<const EpaItem<T>*>(wk.epa_queue);
  enum { IDLE = 0, LIVE = 1, DONE = 2, HANDOVER = 3 };
  int state = IDLE;
  uint32_t it = 0;             // item of this group's polytope
  uint32_t next = blockIdx.x;  // wave-uniform: the wave's items are next, next + gridDim.x, ...
  EpaSupport<T, WE, false> sup;
  sup.lig = lig;
  Epa<T, Grp, CAP, V0_BLOCK> epa;
  EpaLoop<T> L;
  while (true) {
    const uint64_t live = __ballot(state == LIVE);
    const int n_live = __popcll(live) / WE;
    const bool more = next < cnt;
    if (n_live == 0 || (more && G - n_live >= HFCL_EPA_LOOPG_REFILL_MIN)) {
      if (state != LIVE) {
        if (state != IDLE) {
          EpaReadyG<T>* rb = ready_g<T>(wk, QTOP, it);
          if (state == DONE) {
            EpaLoopOutG<T> o;
            epa.loop_out(L, o);
            if (lig == 0) {
              *reinterpret_cast<EpaLoopOutG<T>*>(rb->md) = o;
              rb->state = EPA_READY_DONE;
            }
          } else {  // hand over to the full-capacity tier: the seed and, room permitting, the polytope itself
            const EpaItem<T>* ip = QTOP ? queue + (wk.n - 1u - it) : queue + it;
            uint32_t slot = 0;
            if (lig == 0) slot = atomicAdd(&wk.counts[B_COUNT + 1], 1u);
            slot = __shfl(slot, 0, WE);
            const bool save = epa.resumable && slot < wk.resume_cap;
            if (save) epa_save_block<T, Grp, CAP>(&scratch[grp], resume_slot<T, CAP>(wk, slot));
            if (lig == 0) {  // queue to queue, no local copy
              EpaItem<T>* dst = reinterpret_cast<EpaItem<T>*>(wk.epa_queue2) + slot;
              *dst = *ip;
              if (save) dst->rank = ip->rank | EPA_RESUME_FLAG | (CAP != epa_fast_cap<T> ? EPA_RESUME_SMALL : 0);
              rb->state = EPA_READY_HANDED_OVER;
            }
          }
          Grp::sync();
          state = IDLE;
        }
        const uint64_t lower = live | ~((uint64_t(1) << (grp * WE)) - 1);  // live lanes and lanes >= mine do not count
        const uint32_t rank = uint32_t(__popcll(~lower)) / WE;
        it = next + rank * gridDim.x;
        if (it < cnt) {
          const EpaReadyG<T>* rb = ready_g<T>(wk, QTOP, it);
          if (rb->state == EPA_READY_PENDING) {  // (EPA_READY_NONE: k_epa_prepare_general wrote the record)
            const uint32_t packed = rb->packed;
            sup.a = lib.shapes[rb->sid1];
            sup.b = lib.shapes[rb->sid2];
            if (sup.a.kind == K_CONVEX) sup.h0.load(lib.verts + 3 * size_t(sup.a.vertex_offset), sup.a.num_points, lig);
            if (sup.b.kind == K_CONVEX) sup.h1.load(lib.verts + 3 * size_t(sup.b.vertex_offset), sup.b.num_points, lig);
            sup.md.oR1.r0 = mk<T>(rb->md[0], rb->md[1], rb->md[2]);
            sup.md.oR1.r1 = mk<T>(rb->md[3], rb->md[4], rb->md[5]);
            sup.md.oR1.r2 = mk<T>(rb->md[6], rb->md[7], rb->md[8]);
            sup.md.ot1 = mk<T>(rb->md[9], rb->md[10], rb->md[11]);
            sup.md.identity = (packed >> 6) & 1u;
            epa.reset(&scratch[grp], q.epa_max_iterations, q.epa_tolerance);
            epa.loop_enter(L, epa.install(rb, packed), 0, 0);
            state = LIVE;
          }
        }
      }
      next += uint32_t(G - n_live) * gridDim.x;
      if (n_live == 0 && !more) {
        if (__ballot(state == LIVE) == 0) break;
      }
      continue;
    }
    if (state == LIVE) {
      const int r = epa.step(L, sup);
      if (r != 0) state = r == 1 ? DONE : HANDOVER;
    }
  }
}

template <typename T>
__global__ void __launch_bounds__(256) k_epa_records_general(Work wk, LibView<T> lib, IO<T> io, QParams<T> q, int skip_top) {
  const uint32_t cnt0 = wk.counts[B_COUNT], cnt = cnt0 + (skip_top ? 0u : wk.counts[B_COUNT + 3]);
  for (uint32_t it = blockIdx.x * blockDim.x + threadIdx.x; it < cnt; it += gridDim.x * blockDim.x) {
    const bool top = it >= cnt0;
    const EpaReadyG<T>* rb = ready_g<T>(wk, top, top ? it - cnt0 : it);
    if (rb->state != EPA_READY_DONE) continue;  // handed over (the full-capacity tier writes the record) or final already
    const EpaLoopOutG<T>* lo = reinterpret_cast<const EpaLoopOutG<T>*>(rb->md);
    const uint32_t pair = rb->pair;
    const T r0 = swept_radius(lib.shapes[rb->sid1]), r1 = swept_radius(lib.shapes[rb->sid2]);
    EpaResult<T> res;
    res.status = lo->status;
    res.iterations = lo->iterations;
    res.normal = mk<T>(lo->nx, lo->ny, lo->nz);
    res.depth = lo->depth + (r0 + r1);
    res.rw0_ = mk<T>(lo->rw[0], lo->rw[1], lo->rw[2]);
    res.rw1_ = mk<T>(lo->rw[3], lo->rw[4], lo->rw[5]);
    res.rw2_ = mk<T>(lo->rw[6], lo->rw[7], lo->rw[8]);
    res.r00 = mk<T>(lo->r0[0], lo->r0[1], lo->r0[2]);
    res.r01 = mk<T>(lo->r0[3], lo->r0[4], lo->r0[5]);
    res.r02 = mk<T>(lo->r0[6], lo->r0[7], lo->r0[8]);
    PairOut<T> o;
    epa_finish(res, rb->gjk_iters, load_pose(io.tf1, pair), r0, r1, o);
    write_out<T>(io, q, pair, o);
    write_guess<T>(io, pair, o.cached_guess, 0, 0);
  }
}

// =======================================================================================
// launchers (hfcl_launch.hpp)
// =======================================================================================
// fp32 streams (two or three waves per SIMD hide the refill's global loads: k_epa<fast> 1.87 -> 1.76 ms on cfg3);
// fp64 stays with the batch form: its refills cost more than the idle groups, at one wave per SIMD (round 1: cfg5
// 1.27 -> 1.55 ms) and at two (round 2: 1.07 -> 1.22 ms, profiles/r02_u)
// The streaming kernels are persistent (every wave walks through its share of the queue), so their grid is sized from
// the number of waves the chip holds at once: HFCL_EPA_GRID_ROUNDS x that (1 round: the slowest wave decides and there
// is nothing to fill its tail with, 1.755 ms; 2 rounds: 1.679 ms on cfg3; a grid that is not a multiple -- 16 waves per
// CU with 11 or 12 resident -- runs its last round nearly empty).  The runtime's occupancy query over-estimates what
// the LDS admits: the hardware hands LDS out in 1280-byte units (160 KB / 128; tools/occupancy_probe.hip,
// profiles/r02_g: 13184 B -> 11 workgroups per CU where the API says 12), so that bound is applied here.
#ifndef HFCL_EPA64_CURVED_WE
#define HFCL_EPA64_CURVED_WE 16  // lanes per polytope of the fp64 curved-class fast tier: 4 polytopes x 5104 B = 16 LDS units per
                                 // wave = two waves per SIMD (8 lanes: 8 polytopes, 40 KB, one wave); cfg5 k_epa<fast> 1.37 -> 1.07 ms
#endif
#ifndef HFCL_EPA_GRID_ROUNDS
#define HFCL_EPA_GRID_ROUNDS 2
#endif
template <class K>
static int resident_blocks_per_cu(K kernel) {
  int nb = 0;
  if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb, reinterpret_cast<const void*>(kernel), 64, 0) != hipSuccess || nb < 1) nb = 8;
  hipFuncAttributes fa;
  if (hipFuncGetAttributes(&fa, reinterpret_cast<const void*>(kernel)) == hipSuccess && fa.sharedSizeBytes > 0) {
    const int units = int((fa.sharedSizeBytes + 1279) / 1280);
    nb = std::min(nb, 128 / units);
  }
  return std::max(nb, 1);
}
template <typename T>
void launch_epa_fast(int grid, hipStream_t st, const Work& wk, const LibView<T>& lv, const IO<T>& io, const QParams<T>& q, bool cc_queue, bool general_queue, int n_cus, bool curved_class, hipStream_t st2) {
  if constexpr (sizeof(T) == 4) {
    static const int per_cu_cc = resident_blocks_per_cu(k_epa_stream<T, EPA_WE, EPA_FAST_CAP, true>);
    static const int per_cu_gen = resident_blocks_per_cu(k_epa_stream<T, EPA_WE, EPA_FAST_CAP, false>);
    if (cc_queue)
      hipLaunchKernelGGL((k_epa_stream<T, EPA_WE, EPA_FAST_CAP, true>), dim3(std::min(grid, n_cus * per_cu_cc * HFCL_EPA_GRID_ROUNDS)), dim3(64), 0, st, wk, lv, io, q);
    if (general_queue)
      hipLaunchKernelGGL((k_epa_stream<T, EPA_WE, EPA_FAST_CAP, false>), dim3(std::min(grid, n_cus * per_cu_gen * HFCL_EPA_GRID_ROUNDS)), dim3(64), 0, st, wk, lv, io, q);
  } else {
#ifdef HFCL_EPA64_STREAM_CURVED  // A/B (profiles/r02_q, r02_u): the curved-shape queue through the streaming kernel
    static const int per_cu = resident_blocks_per_cu(k_epa_stream<T, HFCL_EPA64_CURVED_WE, epa_fast_cap<T>, false, true>);
    hipLaunchKernelGGL((k_epa_stream<T, HFCL_EPA64_CURVED_WE, epa_fast_cap<T>, false, true>), dim3(std::min(grid * (HFCL_EPA64_CURVED_WE / EPA_WE), n_cus * per_cu * HFCL_EPA_GRID_ROUNDS)), dim3(64), 0, st, wk, lv, io, q);
    hipLaunchKernelGGL((k_epa<T, EPA_WE, epa_small_cap<T>, 1, 1>), dim3(std::min(grid, n_cus * 32)), dim3(64), 0, st, wk, lv, io, q);
#else
    // one kernel per class, both sized for two waves per SIMD (20 KB of LDS per wave): the curved pairs with 16 lanes per
    // polytope and the large block (CAP 29, 4 polytopes per wave), the polytope pairs with 8 lanes and a block for 13
    // iterations (8 per wave) -- profiles/r02_u
    if (curved_class)
      hipLaunchKernelGGL((k_epa<T, HFCL_EPA64_CURVED_WE, epa_fast_cap<T>, 1, 2>), dim3(std::min(grid * (HFCL_EPA64_CURVED_WE / EPA_WE), n_cus * 16)), dim3(64), 0, st, wk, lv, io, q);
    // (st2: the two classes side by side on two streams -- each kernel's tail is the other's body; the caller forks and joins)
    hipLaunchKernelGGL((k_epa<T, EPA_WE, epa_small_cap<T>, 1, 1>), dim3(std::min(grid, n_cus * 32)), dim3(64), 0, (curved_class && st2) ? st2 : st, wk, lv, io, q);
#endif
  }
}
#if HFCL_UNIT_F32
template void launch_epa_fast<float>(int, hipStream_t, const Work&, const LibView<float>&, const IO<float>&, const QParams<float>&, bool, bool, int, bool, hipStream_t);
#endif
#if HFCL_UNIT_F64
template void launch_epa_fast<double>(int, hipStream_t, const Work&, const LibView<double>&, const IO<double>&, const QParams<double>&, bool, bool, int, bool, hipStream_t);
#endif

#if HFCL_UNIT_F32
void launch_epa_prepare(int grid, hipStream_t st, const Work& wk, const LibView<float>& lv, const IO<float>& io, const QParams<float>& q) {
  hipLaunchKernelGGL((k_epa_prepare<float>), dim3(grid), dim3(256), 0, st, wk, lv, io, q);
}
void launch_epa_loop(int grid, hipStream_t st, const Work& wk, const LibView<float>& lv, const QParams<float>& q, int n_cus) {
  static const int per_cu = resident_blocks_per_cu(k_epa_loop<float, EPA_WE, EPA_FAST_CAP>);
  hipLaunchKernelGGL((k_epa_loop<float, EPA_WE, EPA_FAST_CAP>), dim3(std::min(grid, n_cus * per_cu * HFCL_EPA_LOOP_ROUNDS)), dim3(64), 0, st, wk, lv, q);
}
void launch_epa_resume_cc(int grid, hipStream_t st, const Work& wk, const LibView<float>& lv, const IO<float>& io, const QParams<float>& q) {
  hipLaunchKernelGGL((k_epa_resume_cc<float, HFCL_EPA_CC_RESUME_WE>), dim3(grid), dim3(64), 0, st, wk, lv, io, q);
}
void launch_epa_records(int grid, hipStream_t st, const Work& wk, const LibView<float>& lv, const IO<float>& io, const QParams<float>& q) {
  hipLaunchKernelGGL((k_epa_records<float>), dim3(grid), dim3(256), 0, st, wk, lv, io, q);
}

#endif

// the general three-stage fast tier (fp64: the polytope class from the bottom queue on st, the curved class from the top queue on st2 when
// given; fp32: the bottom queue -- the top one is the convex x convex tier's)
#if !HFCL_KEEP_AB_FORMS
// (the product build does not carry the general staged kernels: the host refuses the option that would launch them)
template <typename T> void launch_epa_prepare_general(int, hipStream_t, const Work&, const LibView<T>&, const IO<T>&, const QParams<T>&, bool) {}
template <typename T> void launch_epa_records_general(int, hipStream_t, const Work&, const LibView<T>&, const IO<T>&, const QParams<T>&, bool) {}
template <typename T> void launch_epa_loop_general(int, hipStream_t, hipStream_t, const Work&, const LibView<T>&, const QParams<T>&, int, bool) {}
#else
template <typename T>
void launch_epa_prepare_general(int grid, hipStream_t st, const Work& wk, const LibView<T>& lv, const IO<T>& io, const QParams<T>& q, bool skip_top) {
  hipLaunchKernelGGL((k_epa_prepare_general<T>), dim3(grid), dim3(256), 0, st, wk, lv, io, q, skip_top ? 1 : 0);
}
template <typename T>
void launch_epa_records_general(int grid, hipStream_t st, const Work& wk, const LibView<T>& lv, const IO<T>& io, const QParams<T>& q, bool skip_top) {
  hipLaunchKernelGGL((k_epa_records_general<T>), dim3(grid), dim3(256), 0, st, wk, lv, io, q, skip_top ? 1 : 0);
}
#ifndef HFCL_EPA_LOOPG_ROUNDS
#define HFCL_EPA_LOOPG_ROUNDS 1
#endif
template <typename T>
void launch_epa_loop_general(int grid, hipStream_t st, hipStream_t st2, const Work& wk, const LibView<T>& lv, const QParams<T>& q, int n_cus, bool curved_class) {
  if constexpr (sizeof(T) == 4) {
    static const int per_cu = resident_blocks_per_cu(k_epa_loop_general<T, EPA_WE, EPA_FAST_CAP, false>);
    hipLaunchKernelGGL((k_epa_loop_general<T, EPA_WE, EPA_FAST_CAP, false>), dim3(std::min(grid, n_cus * per_cu * HFCL_EPA_LOOPG_ROUNDS)), dim3(64), 0, st, wk, lv, q);
  } else {
    static const int per_cu_c = resident_blocks_per_cu(k_epa_loop_general<T, HFCL_EPA64_CURVED_WE, epa_fast_cap<T>, true>);
    static const int per_cu_p = resident_blocks_per_cu(k_epa_loop_general<T, EPA_WE, epa_small_cap<T>, false>);
    if (curved_class)
      hipLaunchKernelGGL((k_epa_loop_general<T, HFCL_EPA64_CURVED_WE, epa_fast_cap<T>, true>), dim3(std::min(grid * (HFCL_EPA64_CURVED_WE / EPA_WE), n_cus * per_cu_c * HFCL_EPA_LOOPG_ROUNDS)), dim3(64), 0, st, wk, lv, q);
    hipLaunchKernelGGL((k_epa_loop_general<T, EPA_WE, epa_small_cap<T>, false>), dim3(std::min(grid, n_cus * per_cu_p * HFCL_EPA_LOOPG_ROUNDS)), dim3(64), 0, (curved_class && st2) ? st2 : st, wk, lv, q);
  }
}
#endif  // HFCL_KEEP_AB_FORMS
#define HFCL_INST_G(T)                                                                                                                                     \
  template void launch_epa_prepare_general<T>(int, hipStream_t, const Work&, const LibView<T>&, const IO<T>&, const QParams<T>&, bool);                   \
  template void launch_epa_records_general<T>(int, hipStream_t, const Work&, const LibView<T>&, const IO<T>&, const QParams<T>&, bool);                   \
  template void launch_epa_loop_general<T>(int, hipStream_t, hipStream_t, const Work&, const LibView<T>&, const QParams<T>&, int, bool);
#if HFCL_UNIT_F32
HFCL_INST_G(float)
#endif
#if HFCL_UNIT_F64
HFCL_INST_G(double)
#endif
#undef HFCL_INST_G

template <typename T>
void launch_epa_full(int grid, hipStream_t st, const Work& wk, const LibView<T>& lv, const IO<T>& io, const QParams<T>& q) {
  hipLaunchKernelGGL((k_epa<T, epa_we2<T>, EPA_MAX_ITER, 2>), dim3(grid), dim3(64), 0, st, wk, lv, io, q);
}
// A very small batch (option epa_direct_max): every seed of the two fast queues moves to the full-capacity tier's queue, which then runs alone -- one
// kernel as long as the batch's longest polytope instead of the fast tier (as long as ITS longest) and the full tier behind it.  One block: the queues hold
// a few thousand seeds at most.  The fast tiers are not launched for such a batch.
template <typename T>
__global__ void __launch_bounds__(1024) k_epa_requeue(Work wk) {
  __shared__ uint32_t base_s;
  const uint32_t n_bottom = min(wk.counts[B_COUNT], wk.n), n_top = min(wk.counts[B_COUNT + 3], wk.n - n_bottom);
  if (threadIdx.x == 0) base_s = wk.counts[B_COUNT + 1];
  __syncthreads();
  const uint32_t base = base_s;
  const EpaItem<T>* const src = reinterpret_cast<const EpaItem<T>*>(wk.epa_queue);
  EpaItem<T>* const dst = reinterpret_cast<EpaItem<T>*>(wk.epa_queue2);
  for (uint32_t i = threadIdx.x; i < n_bottom + n_top; i += blockDim.x) dst[base + i] = src[i < n_bottom ? i : wk.n - 1u - (i - n_bottom)];
  __syncthreads();
  if (threadIdx.x == 0) wk.counts[B_COUNT + 1] = base + n_bottom + n_top;  // (the fast queues' counters stay: what hfcl_last_bucket_counts reports as seeds)
}
template <typename T>
void launch_epa_requeue(hipStream_t st, const Work& wk) {
  hipLaunchKernelGGL((k_epa_requeue<T>), dim3(1), dim3(1024), 0, st, wk);
}
#if HFCL_UNIT_F32
template void launch_epa_requeue<float>(hipStream_t, const Work&);
template void launch_epa_full<float>(int, hipStream_t, const Work&, const LibView<float>&, const IO<float>&, const QParams<float>&);
#endif
#if HFCL_UNIT_F64
template void launch_epa_requeue<double>(hipStream_t, const Work&);
template void launch_epa_full<double>(int, hipStream_t, const Work&, const LibView<double>&, const IO<double>&, const QParams<double>&);
#endif
