// hfcl_k_bvhd.hip -- BVHModel<OBBRSS> x BVHModel<OBBRSS> distance(): the lane walk (k_bvh_distance), its continuations
// (k_bvh_distance_pool; k_bvh_distance_coop, the ordered wave-per-walk form of round 3) and their launcher.
//
// This unit is compiled with -ffp-contract=off (Makefile: FLAGS_k_bvhd): rectDistance, sqrTriDistance and the products in
// front of them are then the reference's operations one for one (its default build does not contract either, and the
// oracle is built the same way), so which of several triangle pairs at the minimal distance is reported -- decided by
// comparisons between values that agree to the last bit or differ in it -- is the reference's choice
// (tests/test_gpu_parity.py::test_bvh_distance*: triangle ids equal to the oracle's).
#include <algorithm>

#include "hfcl_dev.hpp"
#include "hfcl_launch.hpp"

// ---------------------------------------------------------------------------------------
// k_bvh_distance: BVHModel<OBBRSS> x BVHModel<OBBRSS> distance().  distanceRecurse
// (src/traversal/traversal_recurse.cpp:153-203) flattened: both child pairs get their RSS lower
// bound, the farther one is pushed first (with its bound), the nearer one on top; a popped entry is
// skipped when its bound can no longer beat the current minimum (canStop, rel_err = abs_err = 0 as
// latched by the reference's traversal node, traversal_node_bvhs.h:409-410).  Leaves =
// sqrTriDistance in model 1's frame; the result is seeded with triangle 0 x triangle 0 (preprocess).
// ---------------------------------------------------------------------------------------


// Two waves per SIMD (256 VGPRs, 36 B per lane of scratch in fp64) with the 20 KB stack: the compiler's own allocation
// (264 registers) left one.  cfg4's distance() variant, 1M queries: 0.75 -> 1.85 M q/s with the stack and this (profiles/r03_g).
#ifndef HFCL_WPE_BVH_DISTANCE
#define HFCL_WPE_BVH_DISTANCE 2
#endif
template <typename T, bool WIDE>
__global__ void __launch_bounds__(BVHD_BLOCK) __attribute__((amdgpu_waves_per_eu(HFCL_WPE_BVH_DISTANCE, 8))) k_bvh_distance(Work wk, LibView<T> lib, BvhView<T> bv, IO<T> io, QParams<T> q, BvhSpill spill) {
  typedef BvhEntry<WIDE> EN;
  typedef typename EN::E E;
  constexpr int STACK = WIDE ? (BVHD_STACK * 3) / 4 : BVHD_STACK, HALF = STACK / 2;
  __shared__ E stack_e[STACK][BVHD_BLOCK];
  // The bound travels in 4 bytes, rounded DOWN: an entry is skipped when its bound cannot beat the current minimum
  // (canStop), and a bound that is a little too small only means an entry is looked at that the exact bound would have
  // skipped -- it cannot lower the minimum, and the order of the walk was decided on the exact values when it was pushed.
  // fp64: the upper word of the double (sign, exponent, 20 mantissa bits; truncation = rounding down for the bounds, which
  // are >= 0, and exact for the root's -1): the full exponent range, so scenes of any scale keep their pruning (a float
  // would flush the bounds of a 1e-40-sized scene to zero and the walk would visit every pair).
  // Where the truncated bound falls short of the minimum by less than its own resolution (the exact bound may still reach
  // it: exact ties are what prunes a mesh against a shifted copy of itself, tests/test_gpu_parity.py::
  // test_bvh_degenerate_deep_tree) the exact bound is evaluated again -- same inputs, same value -- and decides.
  typedef typename std::conditional<sizeof(T) == 8, uint32_t, float>::type BD;
  __shared__ BD stack_d[STACK][BVHD_BLOCK];
  auto bound_down = [](T d) -> BD {
    if constexpr (sizeof(T) == 8)
      return uint32_t(__double2hiint(d));
    else
      return d;
  };
  auto bound_value = [](BD b) -> T {
    if constexpr (sizeof(T) == 8)
      return __hiloint2double(int(b), 0);
    else
      return b;
  };
  // this lane's slab of spilled (entry, bound) records (WIDE only): entries first, bounds behind them
  E* const slab_e = WIDE && spill.slab ? reinterpret_cast<E*>(spill.slab) + size_t(blockIdx.x * BVHD_BLOCK + threadIdx.x) * spill.cap * 2 : nullptr;
  BD* const slab_d = reinterpret_cast<BD*>(slab_e + spill.cap);
  uint32_t nspill = 0, steps = 0;
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
  auto reload = [&]() {  // the LDS part ran empty: take spilled records back (WIDE)
    if (!WIDE || nspill == 0) return;
    const uint32_t m = min(nspill, uint32_t(HALF));
    for (uint32_t k = 0; k < m; ++k) {
      stack_e[k][tid] = slab_e[nspill - m + k];
      stack_d[k][tid] = slab_d[nspill - m + k];
    }
    nspill -= m;
    sp = int(m);
  };
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
    if (live && sp == 0) reload();
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
          nspill = 0;
          steps = 0;
          leaf(0u, 0u);  // preprocess()
          sp = 1;
          stack_e[0][tid] = 0u;
          stack_d[0][tid] = bound_down(T(-1));
          live = true;
        }
      }
      if (base + uint32_t(n_need) >= cnt) exhausted = true;
      continue;
    }
    // (Triangle pairs -- 6 % of the steps -- are evaluated where they are popped.  Parking them until several lanes of the
    // wave wait, as k_bvh_collide does, loses here: a BV step is two rectangle distances, as heavy as a triangle pair, and
    // the parked lanes miss them: 8 lanes 0.75, 24 lanes 0.56 against 0.83 M q/s; profiles/r03_g.)
    for (;;) {
      if (!WIDE && spill.budget && live && sp > 0 && steps >= spill.budget) {
        // this walk is a long one: its state and stack go to a record, a wave takes it over (k_bvh_distance_coop)
        DistSusp<T>* r = reinterpret_cast<DistSusp<T>*>(spill.susp) + atomicAdd(spill.susp_count, 1u);
        r->pair = pair;
        r->sp = uint32_t(sp);
        r->fb1 = fb1;
        r->fb2 = fb2;
        r->mind = mind;
        r->np1 = np1;
        r->np2 = np2;
        for (int k = 0; k < sp; ++k) {
          r->entry[k] = uint32_t(stack_e[k][tid]);
          r->bound[k] = bound_value(stack_d[k][tid]);
        }
        sp = 0;
        live = false;  // (no record from this lane)
      }
      const bool run = live && sp > 0;
      const int n_run = __popcll(__ballot(run || (live && nspill > 0)));
      if (n_run == 0 || (!exhausted && 64 - n_run >= BVH_REFILL_MIN)) break;
      if (!run) {
        if (live) reload();
        continue;
      }
      ++steps;
      --sp;
      const E e = stack_e[sp][tid];
      const BD dc = stack_d[sp][tid];
      const T de = bound_value(dc);
      if (de >= T(0) && de >= mind) continue;  // canStop(d)
      const uint32_t b1 = EN::first(e), b2 = EN::second(e);
      if constexpr (sizeof(T) == 8) {
        if (de >= T(0) && __hiloint2double(int(dc) + 1, 0) > mind) {  // the exact bound may reach the minimum: ask it
          const T exact = rss_lower_bound(RT_R, RT_T, bv.nodes[m1.node_off + b1], bv.rss[m1.node_off + b1], bv.nodes[m2.node_off + b2],
                                          bv.rss[m2.node_off + b2]);
          if (exact >= mind) continue;
        }
      }
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
      if (WIDE && sp + 2 > STACK && slab_e && nspill + uint32_t(HALF) <= spill.cap) {  // lower half -> the lane's slab
        for (int k = 0; k < HALF; ++k) {
          slab_e[nspill + k] = stack_e[k][tid];
          slab_d[nspill + k] = stack_d[k][tid];
        }
        nspill += uint32_t(HALF);
        for (int k = HALF; k < sp; ++k) {
          stack_e[k - HALF][tid] = stack_e[k][tid];
          stack_d[k - HALF][tid] = stack_d[k][tid];
        }
        sp -= HALF;
      }
      if (sp + 2 > STACK) {
        overflow = true;
        sp = 0;
        nspill = 0;
        continue;
      }
      const E ea = EN::pack(a1, a2), ec = EN::pack(c1, c2);
      const bool c_first = d2 < d1;  // visit (c1,c2) first when it is strictly nearer
      stack_e[sp][tid] = c_first ? ea : ec;
      stack_d[sp][tid] = bound_down(c_first ? d1 : d2);
      ++sp;
      stack_e[sp][tid] = c_first ? ec : ea;
      stack_d[sp][tid] = bound_down(c_first ? d2 : d1);
      ++sp;
    }
  }
}

// ---------------------------------------------------------------------------------------
// k_bvh_distance_coop: the long mesh x mesh distance() walks, a wave per query, 64 stack entries per trip (the scheme of
// k_bvh_coop applied to branch and bound).  Entries of the window whose bound cannot beat the minimum are dropped; box pairs
// are replaced by their two successors with their bounds (nearer one on top) wherever they stand -- deciding that with the
// minimum of the moment can only keep a pair the sequential walk would have skipped, never drop one it would have kept; the
// triangle pairs IN FRONT of the first pair that is split are evaluated together and applied in stack order (the minimum is
// lowered by strictly smaller distances only, so the first triangle pair in DFS order that attains it is reported, as in
// distanceRecurse): the same minimum, triangle ids and witness points as the lane's walk.
// ---------------------------------------------------------------------------------------
constexpr int COOPD_CAP = 960, COOPD_SLACK = 64;
template <typename T>
__global__ void __launch_bounds__(64) __attribute__((amdgpu_waves_per_eu(2, 8)))
k_bvh_distance_coop(Work wk, LibView<T> lib, BvhView<T> bv, IO<T> io, BvhSpill spill) {
  typedef BvhEntry<false> EN;
  __shared__ uint32_t stack_e[COOPD_CAP + COOPD_SLACK];
  __shared__ T stack_d[COOPD_CAP + COOPD_SLACK];
  const int lane = threadIdx.x;
  const uint32_t n_susp = *spill.susp_count;
  const T big = Lim<T>::max(), nanv = Lim<T>::nan();
  auto vtx = [](const T* v, uint32_t i) { return mk<T>(v[3 * size_t(i)], v[3 * size_t(i) + 1], v[3 * size_t(i) + 2]); };
  for (uint32_t qi = blockIdx.x; qi < n_susp; qi += gridDim.x) {
    const DistSusp<T>* const r = reinterpret_cast<const DistSusp<T>*>(spill.susp) + qi;
    const uint32_t pair = r->pair;
    const DMesh m1 = bv.meshes[lib.shapes[wk.shape1[pair]].bvh_index], m2 = bv.meshes[lib.shapes[wk.shape2[pair]].bvh_index];
    const Pose<T> tf1 = load_pose(io.tf1, pair);
    M3<T> RT_R;
    V3<T> RT_T;
    {
      const Pose<T> tf2 = load_pose(io.tf2, pair);
      RT_R = tmul(tf1.R, tf2.R);
      RT_T = tmul(tf1.R, tf2.t - tf1.t);
    }
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    __builtin_amdgcn_wave_barrier();
    int sp = int(r->sp);
    if (lane < sp) {
      stack_e[lane] = r->entry[lane];
      stack_d[lane] = r->bound[lane];
    }
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    __builtin_amdgcn_wave_barrier();
    T mind = r->mind;
    int fb1 = r->fb1, fb2 = r->fb2;
    V3<T> np1 = r->np1, np2 = r->np2;
    bool overflow = false;
    while (sp > 0) {
      const int w = min(64, min(sp, max(COOPD_CAP - sp, 1)));
      const bool act = lane < w;
      const uint32_t e = act ? stack_e[sp - 1 - lane] : 0u;
      const T db = act ? stack_d[sp - 1 - lane] : big;
      sp -= w;
      const bool alive = act && !(db >= T(0) && db >= mind);  // canStop(d), with the minimum of the moment
      const uint32_t b1 = EN::first(e), b2 = EN::second(e);
      const DNode<T>* const p1n = bv.nodes + m1.node_off + b1;
      const DNode<T>* const p2n = bv.nodes + m2.node_off + b2;
      const int32_t fc1 = alive ? p1n->first_child : 0, fc2 = alive ? p2n->first_child : 0;
      const bool l1 = fc1 < 0, l2 = fc2 < 0;
      const bool is_leaf = alive && l1 && l2, split = alive && !(l1 && l2);
      uint32_t ea = 0, ec = 0;
      T d1 = big, d2 = big;
      if (split) {
        uint32_t a1, a2, c1, c2;
        if (l2 || (!l1 && (sqnorm(p1n->extent) > sqnorm(p2n->extent)))) {
          a1 = uint32_t(fc1);
          a2 = b2;
          c1 = a1 + 1;
          c2 = b2;
        } else {
          a1 = b1;
          a2 = uint32_t(fc2);
          c1 = b1;
          c2 = a2 + 1;
        }
        d1 = rss_lower_bound(RT_R, RT_T, bv.nodes[m1.node_off + a1], bv.rss[m1.node_off + a1], bv.nodes[m2.node_off + a2], bv.rss[m2.node_off + a2]);
        d2 = rss_lower_bound(RT_R, RT_T, bv.nodes[m1.node_off + c1], bv.rss[m1.node_off + c1], bv.nodes[m2.node_off + c2], bv.rss[m2.node_off + c2]);
        ea = EN::pack(a1, a2);
        ec = EN::pack(c1, c2);
      }
      const uint64_t smask = __ballot(split);
      const int f = smask ? __ffsll((unsigned long long)smask) - 1 : 64;  // the triangle pairs in front of it are visited now
      const bool visit = is_leaf && lane < f;
      T val = big;
      V3<T> P = mk<T>(nanv, nanv, nanv), Q = P;
      const uint32_t lb1 = uint32_t(-(fc1 + 1)), lb2 = uint32_t(-(fc2 + 1));
      if (visit) {
        const T* v1 = bv.verts + 3 * size_t(m1.vert_off);
        const T* v2 = bv.verts + 3 * size_t(m2.vert_off);
        const uint32_t* t1 = bv.tris + 3 * size_t(m1.tri_off + lb1);
        const uint32_t* t2 = bv.tris + 3 * size_t(m2.tri_off + lb2);
        const T dd = sqr_tri_distance(vtx(v1, t1[0]), vtx(v1, t1[1]), vtx(v1, t1[2]), mul(RT_R, vtx(v2, t2[0])) + RT_T,
                                      mul(RT_R, vtx(v2, t2[1])) + RT_T, mul(RT_R, vtx(v2, t2[2])) + RT_T, P, Q);
        val = hsqrt(dd);
      }
      // The evaluated triangle pairs, in stack order, exactly as the lane's walk takes them: a pair counts only if its bound does
      // not let it be skipped at ITS turn (canStop with the minimum as it stands then), and the minimum is lowered by strictly
      // smaller distances only (DistanceResult::update)
      {
        int start = 0, src = -1;
        T run = mind;
        for (;;) {
          const bool cand = visit && lane >= start && !(db >= T(0) && db >= run) && val < run;
          const uint64_t m = __ballot(cand);
          if (!m) break;
          src = __ffsll((unsigned long long)m) - 1;
          run = __shfl(val, src);
          start = src + 1;
        }
        if (src >= 0) {
          mind = run;
          fb1 = __shfl(int(lb1), src);
          fb2 = __shfl(int(lb2), src);
          np1 = mk<T>(__shfl(P.x, src), __shfl(P.y, src), __shfl(P.z, src));
          np2 = mk<T>(__shfl(Q.x, src), __shfl(Q.y, src), __shfl(Q.z, src));
        }
      }
      // the stack again, in order: visited triangle pairs and dropped entries are gone, a split pair is its two successors
      // (the nearer one on top), a triangle pair behind the first split stays
      const int cnt = split ? 2 : ((is_leaf && lane >= f) ? 1 : 0);
      const uint64_t m2b = __ballot(cnt == 2), m1b = __ballot(cnt == 1);
      const uint64_t deeper = ~((uint64_t(2) << lane) - 1);
      const int pos = sp + 2 * __popcll(m2b & deeper) + __popcll(m1b & deeper);
      if (cnt == 2) {
        const bool c_first = d2 < d1;  // visit (c1, c2) first when it is strictly nearer
        stack_e[pos] = c_first ? ea : ec;
        stack_d[pos] = c_first ? d1 : d2;
        stack_e[pos + 1] = c_first ? ec : ea;
        stack_d[pos + 1] = c_first ? d2 : d1;
      } else if (cnt == 1) {
        stack_e[pos] = e;
        stack_d[pos] = db;
      }
      sp += 2 * __popcll(m2b) + __popcll(m1b);
      if (sp > COOPD_CAP + COOPD_SLACK - 2) {
        overflow = true;
        sp = 0;
      }
      __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
      __builtin_amdgcn_wave_barrier();
    }
    if (lane == 0) {
      PairOut<T> o;
      o.distance = mind;
      o.normal = mk<T>(nanv, nanv, nanv);  // not set by the reference on this path (traversal_node_bvhs.h:454,465)
      o.p1 = xform(tf1, np1);              // postprocess(): model-1 frame -> world
      o.p2 = xform(tf1, np2);
      o.gjk_status = GJK_DID_NOT_RUN;
      o.epa_status = EPA_DID_NOT_RUN;
      o.gjk_iters = o.epa_iters = 0;
      store_bvh_record(io, pair, o, mind <= T(0) ? 0x80000000u : 0u, fb1, fb2, overflow);
    }
  }
}


// ---------------------------------------------------------------------------------------
// k_bvh_distance_pool: the mesh x mesh distance() walks past the lanes' step budget, POOL_Q walks per wave with their tests
// pooled (round 4; replaces k_bvh_distance_coop as the default continuation).
//
// What the measurements of the wave-per-walk form said (profiles/r04_b_mesh_pmc.txt): 0.47 of the fp64 issue peak with 16 of 64 lanes
// active on average -- a walk's window holds 10-20 box pairs to split and each of their lanes ran both children's rectangle
// distances one after the other while the rest idled, and triangle pairs were evaluated a handful at a time.  Here
//   * a wave owns POOL_Q walks (slots), each with its stack in LDS, in DFS order (top = next in the reference's order);
//   * every trip, each slot offers the top POOL_SEG * POOL_E entries of its stack; entries whose bound cannot beat the slot's
//     minimum are dropped, box pairs to split put their TWO child tests into one list for the whole wave, and the list is
//     worked off 64 tests at a time, one rss_lower_bound per lane (the packed 128-B DNodeD records: one line per node).  Full
//     rounds only, plus a last partial one when it is the only one or fills `part_min` lanes: what is left over stays on its
//     stack and is offered again;
//   * triangle pairs stay on the stack until the wave holds `leaf_min` of them in its windows (or has too few box tests
//     to fill half of its lanes), then up to 64 of them are evaluated together, one per lane, whichever slot they belong to;
//   * the window is written back in order: a split pair becomes its two children (nearer one on top, the reference's
//     `d2 < d1` rule), a deferred triangle pair or box pair stays, everything else is gone.
// An entry is one 32-bit word and its bound: what the next trip needs to know about it (triangle ids, or which node is split
// and its first child), filled in by the lane that tested it from the records it had loaded anyway: a trip has no dependent
// gather before its tests.
// What bounds it now (profiles/r04_d): a walk's frontier is ~6 box pairs per trip however wide the window, so a wave of four
// walks fills 42 of 64 lanes per round; a round runs as many candidate blocks of rectDistance as its slowest lane needs (3-4
// against 1.2 on average); more walks per wave fill the rounds but lengthen every walk, and the batch ends with its longest
// walks (18 600 box tests against 3 400 on average).  Decoupling the tests from the trips (queues of tests and an "in flight"
// state for the entries that wait for them) was built and measured 3-5x slower (profiles/r04_f): every level of the
// trees then takes two trips.
//
// Order.  The reference visits triangle pairs in DFS order and keeps the FIRST one that attains the minimum
// (DistanceResult::update lowers on `<` only), and it skips an entry when its bound is >= the minimum of that moment.  The
// minimum itself does not depend on the order of the visits; the reported pair is the first in DFS order among the visited
// ones that attain it.  So each slot keeps, with its minimum, a marker `p` = how many entries of its (DFS-ordered) stack
// come AFTER the pair that set it: a triangle pair replaces the minimum when it is strictly smaller, or equal and in front
// of it (index >= p); an entry is dropped when its bound exceeds the minimum, or equals it and stands behind it (index < p:
// the sequential walk would already hold that minimum at the entry's turn) -- an entry in front of it with an equal bound
// is kept, as the sequential walk (whose minimum was still larger there) would have.  Writing the window back updates `p`
// by counting what the entries behind the marker became.  The lane phase hands a walk over with everything on its stack
// behind its minimum (p = sp).  Measured (profiles/r04_c): the oracle's distance, triangle ids and witness points, bit for bit,
// in 100 000 of 100 000 cfg4d queries; 95 % of the separated queries have several triangle pairs at exactly the minimal
// distance (shared vertices), so the marker is what decides the reported ids.
//
// Verification (round 6).  The rule stands on one premise: a bound never exceeds a distance beneath it.  In floating point it does, by
// an ulp of the scene's size (rectDistance against sqrTriDistance), and then the sequential walk may DROP an entry -- its bound against
// the minimum it holds at that entry's turn -- that the pool, which cannot know that minimum for entries in front of its own, keeps: the
// pool then reports a pair the reference never visits (one query in 20 000 without the `margin` below, one in 450 000 with a wide one).
// Any such case needs a triangle pair that sets the slot's minimum while standing under a bound -- its own or an ancestor's, the
// hierarchy's bounds are not monotone -- ABOVE its distance (else every entry of its chain passes the sequential walk's test whatever
// the minimum).  Every entry therefore carries the largest bound of its chain (`st_c`); a walk in which a pair with such a chain sets
// the minimum is `flagged` and, instead of being written, walked AGAIN by its slot from the suspended record in "ordered" mode: the
// pooled minimum tells which box pairs the sequential walk splits whatever its minimum (bound below it: split wherever they stand, in
// parallel) and beyond which bound nothing matters (`cut`); the few entries between, and every triangle pair, are decided at their turn
// with the minimum of that turn; the walk stops at the first pair that reaches the pooled minimum.  0.6 % of cfg4d's walks at 0.5 % of
// the kernel's time; every byte of every record equal to the lanes' sequential walk (tools/distance_order_soak.py; HFCL_POOL_RERUN=2
// sends every walk through the ordered mode, = 0 is the round-5 behaviour).
// ---------------------------------------------------------------------------------------
#ifndef HFCL_POOL_Q
#define HFCL_POOL_Q 4
#endif
#ifndef HFCL_POOL_E
#define HFCL_POOL_E 1
#endif
// POOL_Q slots per wave, POOL_SEG lanes per slot, POOL_E window entries per lane: a slot offers POOL_SEG * POOL_E entries a trip
constexpr int POOL_Q = HFCL_POOL_Q, POOL_SEG = 64 / POOL_Q, POOL_E = HFCL_POOL_E, POOL_WIN = POOL_SEG * POOL_E;
static_assert(POOL_E >= 1 && POOL_E <= 3, "per-lane counts travel as three ballots (values up to 2 * POOL_E <= 7)");
// a slot's stack: windows narrow once POOL_CAPW entries are in use, down to plain DFS, which adds at most the lanes'
// stack depth (BVHD_STACK >= depth1 + depth2 + 2, make_bvh_spill) on top
#ifndef HFCL_POOL_CAPW
#define HFCL_POOL_CAPW (640 / HFCL_POOL_Q)
#endif
#ifndef HFCL_POOL_CAPW8
#define HFCL_POOL_CAPW8 72
#endif
constexpr int POOL_CAPW = HFCL_POOL_CAPW;
#ifndef HFCL_POOL_BIG_BATCH
#define HFCL_POOL_BIG_BATCH 400000
#endif
constexpr int POOL_BIG_BATCH = HFCL_POOL_BIG_BATCH;  // queries from which a wave takes 8 walks instead of POOL_Q
#ifndef HFCL_POOL_ROUNDS
#define HFCL_POOL_ROUNDS 2
#endif
constexpr int POOL_ROUNDS = HFCL_POOL_ROUNDS, POOL_TESTS = 64 * POOL_ROUNDS;  // child tests of a trip: full rounds of 64
// HFCL_POOL_CHEAP (variant builds; default 0): a trip first puts rss_cheap_bound() to its candidate tests -- up to POOL_CAND of them, a round more
// than it runs exactly -- and only those the bound does not decide go through rectDistance, packed into full rounds again (hfcl_bvh.hpp: the pairs
// it decides are pairs the exact value would drop as well, so the walk is the same).  Exact, and decides 30 % of the tests, but the pass that
// forms the bounds reads both nodes and multiplies the frames as the exact test does: cfg4d 25.4 -> 30.9 ms alone, 21.5 -> 25.6 ms on top of
// HFCL_POOL_DROP_DEAD (profiles/r05_g).
#ifndef HFCL_POOL_CHEAP
#define HFCL_POOL_CHEAP 0
#endif
constexpr int POOL_CAND = HFCL_POOL_CHEAP ? 64 * (POOL_ROUNDS + 1) : POOL_TESTS;
#ifndef HFCL_POOL_DROP_DEAD
#define HFCL_POOL_DROP_DEAD 1
#endif
constexpr uint32_t POOL_MAX_NODES = 32767;

// A stack entry is ONE 32-bit word next to its bound -- what the next trip needs to know about the node pair (n1, n2):
// bit 31 = two leaves -> bits 0-14 / 15-29 the triangle ids; else bit 30 = node 1 is the one to split (distanceRecurse's descent
// rule: node 2 a leaf, or node 1 not a leaf and larger), bits 0-14 = the node that is kept, bits 15-29 = the first child of the
// node that is split.  15-bit ids: models of up to POOL_MAX_NODES nodes (larger narrow-form models take k_bvh_distance_coop).
__device__ __forceinline__ uint32_t pool_entry_info(uint32_t n1, int32_t fc1, uint32_t rk1, uint32_t n2, int32_t fc2, uint32_t rk2) {
  const bool l1 = fc1 < 0, l2 = fc2 < 0;
  if (l1 && l2) return 0x80000000u | uint32_t(-(fc1 + 1)) | (uint32_t(-(fc2 + 1)) << 15);
  const bool side1 = l2 || (!l1 && rk1 > rk2);
  return side1 ? (0x40000000u | n2 | (uint32_t(fc1) << 15)) : (n1 | (uint32_t(fc2) << 15));
}

#ifndef HFCL_WPE_BVHD_POOL
#define HFCL_WPE_BVHD_POOL 2
#endif
// HFCL_POOL_PROF (variant builds only, tools/build_variant.sh): wave clocks per phase and event counts into pool_prof[16]
// (0 scan / 1 box tests / 2 triangle tests / 3 write-back / 4 refill clocks; 8 trips, 9 box rounds, 10 box tests, 11 triangle
// passes, 12 triangle tests, 13 walks), read back by tools/pool_prof.py through hfcl_debug_pool_prof
#ifdef HFCL_POOL_PROF
__device__ unsigned long long pool_prof[16];
#define POOL_T0() const unsigned long long t0_ = __builtin_readcyclecounter()
#define POOL_T(i)                                                                      \
  do {                                                                                 \
    if (lane == 0) atomicAdd(&pool_prof[i], __builtin_readcyclecounter() - tq_);       \
    tq_ = __builtin_readcyclecounter();                                                \
  } while (0)
#define POOL_C(i, v)                                                  \
  do {                                                                \
    if (lane == 0) atomicAdd(&pool_prof[i], (unsigned long long)(v)); \
  } while (0)
#else
#define POOL_T(i)
#define POOL_C(i, v)
#endif
// PQ: walks per wave.  POOL_Q (4) for batches up to POOL_BIG_BATCH queries; 8 (8-entry windows, shorter stacks) beyond: fuller rounds,
// and the longer life of every walk no longer shows at the end of the batch (1M queries: 224 -> 215 ms; 100k: 25.9 -> 27.3 ms)
template <typename T, int PQ>
__global__ void __launch_bounds__(64) __attribute__((amdgpu_waves_per_eu(HFCL_WPE_BVHD_POOL, 8))) k_bvh_distance_pool(Work wk, LibView<T> lib, BvhView<T> bv, IO<T> io, BvhSpill spill) {
  constexpr int Q = PQ, SEG = 64 / PQ;
  // (PQ = 8: 72 entries before the windows narrow -- it was 104 --: with the chain word per entry the block is 20 320 B, eight waves per CU.
  // 1M queries 183 -> 198 ms; more entries at 7 or 6 waves per CU: 215 / 239 ms; the chain in 16 bits -- 90 entries -- flags half the walks:
  // 228 ms; four walks per wave at this size: 201 ms.  profiles/r06_a)
  constexpr int CAPW = PQ == POOL_Q ? POOL_CAPW : HFCL_POOL_CAPW8, CAP = CAPW + BVHD_STACK + 8;
  __shared__ uint32_t st_x[Q][CAP];
  __shared__ T st_d[Q][CAP];
  // The largest bound among an entry's ancestors (since the walk came to the pool) where it exceeds the entry's own: 0 = none does, else that
  // bound rounded UP to a float (bounds of a hierarchy are not monotone, and the sequential walk drops a subtree on its ROOT's bound).  Read
  // when a triangle pair becomes the slot's minimum: if the chain's largest bound exceeds the pair's distance -- the roundings of
  // rectDistance and sqrTriDistance, an ulp of the scene's size -- the sequential walk may never have come to this pair, and the walk is
  // re-run in order (BvhSpill::flag_list).
  __shared__ float st_c[Q][CAP];
  __shared__ T q_rt[Q][12];   // RT_R (row-major), RT_T of the slot's query
  __shared__ T q_wit[Q][6];   // witness points of the slot's minimum, model-1 frame
  __shared__ int q_fb[Q][2];
  __shared__ uint32_t q_noff[Q][2];  // node_off of the slot's two models
  __shared__ uint32_t q_off[Q][4];  // vert_off, tri_off of the slot's two models (the lane that evaluates a triangle pair may belong to another slot)
  __shared__ int q_win[Q];          // index in the triangle list of the pair that set the slot's minimum in this trip (-1: none)
  __shared__ uint32_t t_n1[POOL_CAND], t_n2[POOL_CAND], t_x[POOL_CAND];
  __shared__ uint8_t t_q[POOL_CAND];
  __shared__ uint8_t t_live[POOL_CAND];  // the candidates rss_cheap_bound left undecided, in order
  __shared__ T t_res[POOL_CAND];
  __shared__ T q_bar[Q];  // what a bound must exceed to be dropped wherever its entry stands: the slot's minimum + margin
  __shared__ uint32_t l_x[64];  // the triangle pairs evaluated in this trip (at most one per lane): the entry's info word ...
  __shared__ uint8_t l_q[64];   // ... and its slot
  __shared__ T l_val[64];
  constexpr int E = POOL_E, WIN = SEG * E;
  const int lane = threadIdx.x, q = lane / SEG, j = lane % SEG;
  const uint64_t lt_mask = (uint64_t(1) << lane) - 1;
  const uint64_t segm = (SEG == 64 ? ~uint64_t(0) : ((uint64_t(1) << SEG) - 1)) << (q * SEG);
  const uint64_t deeper = segm & ~((uint64_t(2) << lane) - 1);  // the lanes of my slot that hold entries further down
  // sum of a small per-lane count c (0 ... 7) over the lanes of mask m
  auto lane_sum = [](int c, uint64_t m) -> int {
    return __popcll(__ballot((c & 1) != 0) & m) + 2 * __popcll(__ballot((c & 2) != 0) & m) + 4 * __popcll(__ballot((c & 4) != 0) & m);
  };
  const uint32_t n_susp = *spill.susp_count;
  const int leaf_min = int(spill.pool_leaf_min), starve = int(spill.pool_starve), part_min = int(spill.pool_part_min);
  uint32_t trip = 0;
  const T big = Lim<T>::max(), nanv = Lim<T>::nan();
  auto sync = []() {
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    __builtin_amdgcn_wave_barrier();
  };
  auto vtx = [](const T* v, uint32_t i) { return mk<T>(v[3 * size_t(i)], v[3 * size_t(i) + 1], v[3 * size_t(i) + 2]); };
  // slot state, identical in the SEG lanes of a slot
  // ordered: the slot walks a flagged record again, in the reference's order (below); redo: it is about to take that record again
  bool active = false, exhausted = false, flagged = false, ordered = false, redo = false;
  int sp = 0, p = 0;
  T mind = big, margin = T(0), pooled = T(0), cut = big;
  uint32_t pair = 0, off1 = 0, off2 = 0, susp_it = 0;
  DMesh m1 = {0, 0, 0, 0}, m2 = {0, 0, 0, 0};
#ifdef HFCL_POOL_PROF
  unsigned long long tq_ = __builtin_readcyclecounter();
#endif
  for (;;) {
    // ---- slots without a walk take the next suspended ones
    const uint64_t idle = __ballot(!active && !redo && j == 0);
    const bool any_redo = __ballot(redo) != 0;
    if ((idle && !exhausted) || any_redo) {
      uint32_t it = n_susp;
      if (idle && !exhausted) {
        const int n_need = __popcll(idle);
        uint32_t base = 0;
        if (lane == 0) base = atomicAdd(spill.pool_ticket, uint32_t(n_need));
        base = __builtin_amdgcn_readfirstlane(base);
        if (base + uint32_t(n_need) >= n_susp) exhausted = true;
        it = base + uint32_t(__popcll(idle & ((uint64_t(1) << (q * SEG)) - 1)));  // (bits of idle sit at the slots' first lanes)
      }
      if (redo) it = susp_it;  // the slot's own record once more
      if ((!active && it < n_susp) || redo) {
        const DistSusp<T>* const r = reinterpret_cast<const DistSusp<T>*>(spill.susp) + it;
        pair = r->pair;
        susp_it = it;
        ordered = redo;
        redo = false;
        flagged = false;  // (the lane's minimum is one the sequential walk holds: the lane IS that walk up to here)
        m1 = bv.meshes[lib.shapes[wk.shape1[pair]].bvh_index];
        m2 = bv.meshes[lib.shapes[wk.shape2[pair]].bvh_index];
        off1 = m1.node_off;
        off2 = m2.node_off;
        sp = int(r->sp);
        p = sp;  // everything on the stack comes after what the lane has visited
        mind = r->mind;
        const Pose<T> ltf1 = load_pose(io.tf1, pair), ltf2 = load_pose(io.tf2, pair);
        const M3<T> lR = tmul(ltf1.R, ltf2.R);
        const V3<T> lt = tmul(ltf1.R, ltf2.t - ltf1.t);
        {
          // what a bound may exceed a distance beneath it by: both are differences of coordinates of the size of the scene, so the
          // slack is a few ulps of THAT (the models' root volumes and their offset), not of the distance -- a pair 0.006 apart in a
          // scene of size 3 had its bound 81 eps of the distance above it (profiles/r04_c section 3)
          const DNodeD<T>* const a0 = bv.dnodes + off1;
          const DNodeD<T>* const b0 = bv.dnodes + off2;
          const V3<T> dt = ltf2.t - ltf1.t;
          const T scale = habs(dt.x) + habs(dt.y) + habs(dt.z) + a0->l0 + a0->l1 + T(2) * a0->r + habs(a0->Tr.x) + habs(a0->Tr.y) + habs(a0->Tr.z) +
                          b0->l0 + b0->l1 + T(2) * b0->r + habs(b0->Tr.x) + habs(b0->Tr.y) + habs(b0->Tr.z);
          // (16 eps of the scene's size: ~300x the slack that was observed, 1.1e-16 in a scene of size ~10.  Within the margin the rule
          // can also go wrong the other way -- the sequential walk DROPS an entry in front whose bound exceeds the minimum it holds at
          // that turn by an ulp, and with it a pair that ties the final minimum (seen once in 450 000 queries with a margin of 256 eps,
          // profiles/r04_c section 3): those walks are the flagged ones, st_c)
          margin = T(16) * Lim<T>::eps() * scale;
        }
        for (int k = j; k < sp; k += SEG) {
          const uint32_t e = r->entry[k];
          const DNodeD<T>* const a = bv.dnodes + off1 + (e & 0xFFFFu);
          const DNodeD<T>* const b = bv.dnodes + off2 + (e >> 16);
          st_x[q][k] = pool_entry_info(e & 0xFFFFu, a->first_child, a->rank, e >> 16, b->first_child, b->rank);
          // fp64: the lane's stack carries the bounds rounded down to 32 bits; here they are the exact values again (the same inputs, the
          // same operations), which is what the chain test compares distances with (the root's -1 stays)
          T bk = r->bound[k];
          if constexpr (sizeof(T) == 8) {
            if (bk >= T(0)) bk = rss_lower_bound(lR, lt, *a, *b);
          }
          st_d[q][k] = bk;
          st_c[q][k] = 0.0f;  // (its ancestors were decided by the lane, in the reference's order)
        }
        if (j == 0) {
          const M3<T>& R = lR;
          const V3<T>& t = lt;
          T* o = q_rt[q];
          o[0] = R.r0.x; o[1] = R.r0.y; o[2] = R.r0.z; o[3] = R.r1.x; o[4] = R.r1.y; o[5] = R.r1.z;
          o[6] = R.r2.x; o[7] = R.r2.y; o[8] = R.r2.z; o[9] = t.x; o[10] = t.y; o[11] = t.z;
          q_fb[q][0] = r->fb1;
          q_fb[q][1] = r->fb2;
          q_noff[q][0] = m1.node_off; q_noff[q][1] = m2.node_off;
          q_off[q][0] = m1.vert_off; q_off[q][1] = m1.tri_off; q_off[q][2] = m2.vert_off; q_off[q][3] = m2.tri_off;
          T* w6 = q_wit[q];
          w6[0] = r->np1.x; w6[1] = r->np1.y; w6[2] = r->np1.z; w6[3] = r->np2.x; w6[4] = r->np2.y; w6[5] = r->np2.z;
        }
        active = true;
      }
      sync();
    }
    if (__ballot(active) == 0) break;  // (only once the records have run out)
    POOL_T(4);
    // ---- the windows: lane j of a slot holds entries j * E ... j * E + E - 1 from the top of the slot's stack
    const int w = active ? min(WIN, min(sp, max(CAPW - sp, 1))) : 0;
    const int base_i = sp - w;
    uint32_t x[E];
    T db[E];
    float ch[E];
    int idx[E];
    bool is_leaf[E], split[E], held[E];
    int ns_l = 0, nl_l = 0;
#pragma unroll
    for (int u = 0; u < E; ++u) {
      const int wj = j * E + u;
      const bool act = wj < w;
      idx[u] = sp - 1 - wj;
      x[u] = 0u;
      db[u] = big;
      ch[u] = 0.0f;
      if (act) {
        x[u] = st_x[q][idx[u]];
        db[u] = st_d[q][idx[u]];
        ch[u] = st_c[q][idx[u]];
      }
      // canStop(bound) with the slot's minimum.  Behind the pair that set the minimum (idx < p) this is the sequential walk's
      // test at the entry's turn.  In front of it the sequential walk held a larger minimum at the entry's turn, and a bound
      // can exceed a distance below it by an ulp (a leaf pair's rectangles against its triangles): such an entry may hold
      // the first pair at the minimal distance, so entries in front are only dropped when their bound is clear of the minimum
      // by `margin` (4 eps of the scene's size, set when the slot takes the walk)
      // (ordered mode: what lies above the cut is gone, everything else is decided below)
      const bool alive = act && (ordered ? !(db[u] > cut) : !(db[u] >= T(0) && (idx[u] < p ? db[u] >= mind : db[u] > mind + margin)));
      held[u] = false;
      is_leaf[u] = alive && (x[u] >> 31) != 0u;
      split[u] = alive && (x[u] >> 31) == 0u;
    }
    // ---- Ordered mode (a flagged walk, again from its record): the reference's decisions, each at its turn.  The pooled pass left two
    // numbers: `pooled`, its minimum -- no triangle pair the sequential walk visits lies below it, so a box pair whose bound is below
    // it is split by that walk whatever minimum it holds at its turn: those are split here wherever they stand --, and `cut` above it,
    // beyond which nothing matters.  What lies between (a handful of entries) hangs on the minimum of ITS turn, and so does every
    // triangle pair: a box pair of the band is decided only at the top of the stack, the triangle pairs in front of the first box pair
    // of the window are evaluated together and applied in stack order (as k_bvh_distance_coop does), everything else waits.  The
    // minimum is then the sequential walk's at every decision, and the walk is over as soon as it reaches `pooled`: that pair is the
    // first the reference meets at the final distance.
    if constexpr (E == 1) {
      if (__ballot(ordered) != 0) {
        const uint64_t boxm = __ballot(split[0]) & segm;
        const int f = boxm ? __ffsll((unsigned long long)boxm) - 1 - q * SEG : SEG;  // the slot's first box pair from the top
        if (ordered) {
          if (split[0] && !(db[0] < pooled)) {  // a box pair of the band
            if (j == 0) {
              if (db[0] >= mind) split[0] = false;  // canStop at its turn: gone
            } else {
              split[0] = false;
              held[0] = true;
            }
          } else if (is_leaf[0]) {
            if (j > f) {
              is_leaf[0] = false;  // behind a box pair: not its turn yet
              held[0] = true;
            } else if (db[0] >= T(0) && db[0] >= mind) {
              is_leaf[0] = false;  // canStop with a minimum that is not above the one of its turn: gone
            }
          }
        }
      }
    }
#pragma unroll
    for (int u = 0; u < E; ++u) {
      ns_l += split[u] ? 1 : 0;
      nl_l += is_leaf[u] ? 1 : 0;
    }
    // ---- the children's tests of every split pair of the wave in one list
    const int ns = lane_sum(ns_l, ~uint64_t(0));
    // Only full rounds of 64 tests are run (at most POOL_ROUNDS of them), and a last partial one when nothing else would run or
    // it fills `part_min` lanes: the split pairs beyond stay as they are and are offered again in the next trip, together with
    // the children of the pairs that were split -- a round costs the same with 3 lanes as with 64 (profiles/r04_d: 42 of 64
    // lanes per round before).  Which end of the wave's windows is served first alternates, so no slot waits for long.
    const bool up = (trip & 1u) == 0u;
    ++trip;
    const int t_all = 2 * ns;
    int t_run = min(t_all, POOL_CAND);
    if (!HFCL_POOL_CHEAP && t_run > 64 && (t_run & 63) != 0 && (t_run & 63) < part_min) t_run &= ~63;
    if (HFCL_POOL_CHEAP && j == 0) q_bar[q] = mind + margin;
    int k2[E];
    {
      int k = lane_sum(ns_l, up ? lt_mask : ~((uint64_t(2) << lane) - 1));
#pragma unroll
      for (int uu = 0; uu < E; ++uu) {
        const int u = up ? uu : E - 1 - uu;
        k2[u] = 2 * k;
        if (split[u]) {
          if (2 * k < t_run) {
            const uint32_t keep = x[u] & 0x7FFFu, fc = (x[u] >> 15) & 0x7FFFu;
            const bool side1 = (x[u] & 0x40000000u) != 0u;
            const uint32_t a1 = side1 ? fc : keep, a2 = side1 ? keep : fc, c1 = side1 ? fc + 1 : keep, c2 = side1 ? keep : fc + 1;
            t_n1[2 * k] = a1;
            t_n2[2 * k] = a2;
            t_q[2 * k] = uint8_t(q);
            t_n1[2 * k + 1] = c1;
            t_n2[2 * k + 1] = c2;
            t_q[2 * k + 1] = uint8_t(q);
          } else {
            split[u] = false;  // not in this trip: the entry stays as it is
            held[u] = true;
          }
          ++k;
        }
      }
    }
    sync();
    POOL_T(0);
    POOL_C(8, 1);
        int n_live = t_run;
    if constexpr (HFCL_POOL_CHEAP != 0) {
      // the cheap bound on every candidate; the undecided ones listed in order (t_live), marked -1 until their exact value is there
      n_live = 0;
      for (int tb = 0; tb < t_run; tb += 64) {
        const int t = tb + lane;
        bool live = false;
        if (t < t_run) {
          const uint32_t ts = t_q[t];
          const T* const rt = q_rt[ts];
          M3<T> R0;
          R0.r0 = mk<T>(rt[0], rt[1], rt[2]);
          R0.r1 = mk<T>(rt[3], rt[4], rt[5]);
          R0.r2 = mk<T>(rt[6], rt[7], rt[8]);
          const V3<T> T0 = mk<T>(rt[9], rt[10], rt[11]);
          const DNodeD<T> A = bv.dnodes[q_noff[ts][0] + t_n1[t]], B = bv.dnodes[q_noff[ts][1] + t_n2[t]];
          live = !(rss_cheap_bound(R0, T0, A, B) > q_bar[ts]);
          t_res[t] = live ? T(-1) : big;  // (a bound of `big` is dropped wherever the entry stands)
          t_x[t] = 0u;
        }
        const uint64_t lm = __ballot(live);
        if (live) t_live[n_live + __popcll(lm & lt_mask)] = uint8_t(t);
        n_live += __popcll(lm);
      }
      POOL_C(14, t_run - n_live);
      sync();
      // full rounds of the undecided ones (and a last partial one when it is the only one or fills `part_min` lanes); the rest keep their mark
      if (n_live > 64 * POOL_ROUNDS) n_live = 64 * POOL_ROUNDS;
      if (n_live > 64 && (n_live & 63) != 0 && (n_live & 63) < part_min) n_live &= ~63;
    }
    POOL_C(10, n_live);
    for (int tb = 0; tb < n_live; tb += 64) {
      POOL_C(9, 1);
      if (tb + lane < n_live) {
        const int t = HFCL_POOL_CHEAP ? int(t_live[tb + lane]) : tb + lane;
        const uint32_t ts = t_q[t];
        const T* const rt = q_rt[ts];
        M3<T> R0;
        R0.r0 = mk<T>(rt[0], rt[1], rt[2]);
        R0.r1 = mk<T>(rt[3], rt[4], rt[5]);
        R0.r2 = mk<T>(rt[6], rt[7], rt[8]);
        const V3<T> T0 = mk<T>(rt[9], rt[10], rt[11]);
        const uint32_t n1 = t_n1[t], n2 = t_n2[t];
        const DNodeD<T> A = bv.dnodes[q_noff[ts][0] + n1], B = bv.dnodes[q_noff[ts][1] + n2];
        t_res[t] = rss_lower_bound(R0, T0, A, B);
        t_x[t] = pool_entry_info(n1, A.first_child, A.rank, n2, B.first_child, B.rank);
      }
    }
    sync();
    T d1[E], d2[E];
    uint32_t xa[E], xc[E];
#pragma unroll
    for (int u = 0; u < E; ++u) {
      d1[u] = d2[u] = big;
      xa[u] = xc[u] = 0u;
      if (split[u]) {
        d1[u] = t_res[k2[u]];
        d2[u] = t_res[k2[u] + 1];
        xa[u] = t_x[k2[u]];
        xc[u] = t_x[k2[u] + 1];
        if (HFCL_POOL_CHEAP && (d1[u] < T(0) || d2[u] < T(0))) {  // a child's exact test did not fit into this trip's rounds: the entry stays as it is
          split[u] = false;
          held[u] = true;
        }
      }
    }
    POOL_T(1);
    // ---- the triangle pairs, once they are worth a pass: the first 64 of the wave's windows, one per lane
    const int nl = lane_sum(nl_l, ~uint64_t(0));
    const bool do_leaves = nl > 0 && (nl >= leaf_min || 2 * ns < starve || ns == 0);  // (nothing but triangles in the windows: they run, whatever the knobs say)
    int jw = -1;  // the window entry (j * E + u) that set a new minimum in this trip (slot-uniform)
    bool leaf_eval[E];
#pragma unroll
    for (int u = 0; u < E; ++u) leaf_eval[u] = false;
    if (do_leaves) {
      int li[E];
      {
        int l = lane_sum(nl_l, lt_mask);
#pragma unroll
        for (int u = 0; u < E; ++u) {
          li[u] = l;
          if (is_leaf[u]) {
            if (l < 64) {
              leaf_eval[u] = true;
              l_x[l] = x[u];
              l_q[l] = uint8_t(q);
            }
            ++l;
          }
        }
      }
      if (j == 0) q_win[q] = -1;
      sync();
      const int nle = min(nl, 64);
      T val = big;
      V3<T> P = mk<T>(nanv, nanv, nanv), Qp = P;
      uint32_t my_slot = 0u, lb1 = 0u, lb2 = 0u;
      if (lane < nle) {
        my_slot = l_q[lane];
        lb1 = l_x[lane] & 0x7FFFu;
        lb2 = (l_x[lane] >> 15) & 0x7FFFu;
        const T* const rt = q_rt[my_slot];
        M3<T> R0;
        R0.r0 = mk<T>(rt[0], rt[1], rt[2]);
        R0.r1 = mk<T>(rt[3], rt[4], rt[5]);
        R0.r2 = mk<T>(rt[6], rt[7], rt[8]);
        const V3<T> T0 = mk<T>(rt[9], rt[10], rt[11]);
        const T* v1 = bv.verts + 3 * size_t(q_off[my_slot][0]);
        const T* v2 = bv.verts + 3 * size_t(q_off[my_slot][2]);
        const uint32_t* t1 = bv.tris + 3 * size_t(q_off[my_slot][1] + lb1);
        const uint32_t* t2 = bv.tris + 3 * size_t(q_off[my_slot][3] + lb2);
        const T dd = sqr_tri_distance(vtx(v1, t1[0]), vtx(v1, t1[1]), vtx(v1, t1[2]), mul(R0, vtx(v2, t2[0])) + T0,
                                      mul(R0, vtx(v2, t2[1])) + T0, mul(R0, vtx(v2, t2[2])) + T0, P, Qp);
        val = hsqrt(dd);
        l_val[lane] = val;
      }
      sync();
      // DistanceResult::update over the slot's evaluated pairs: the smallest value, the first in DFS order among equals (window
      // index 0 is the top of the stack), and against the standing minimum a tie wins only in front of it
      T bestv = big;
      int bw = 64 * E, bl = -1;  // window index and list index of the lane's / the slot's best candidate
      int bviol = 0;             // ... and whether a bound of its chain exceeds its distance
#pragma unroll
      for (int u = 0; u < E; ++u) {
        if (leaf_eval[u]) {
          const T v = l_val[li[u]];
          const bool cand = v < mind || (v == mind && idx[u] >= p);
          if (cand && (bl < 0 || v < bestv)) {  // (u ascending = DFS order: a later equal value does not replace)
            bestv = v;
            bw = j * E + u;
            bl = li[u];
            bviol = (ch[u] != 0.0f ? T(ch[u]) > v : db[u] > v) ? 1 : 0;
          }
        }
      }
#pragma unroll
      for (int m = 1; m < SEG; m <<= 1) {
        const T ov = __shfl_xor(bestv, m);
        const int ow = __shfl_xor(bw, m), ol = __shfl_xor(bl, m), oviol = __shfl_xor(bviol, m);
        if (ol >= 0 && (bl < 0 || ov < bestv || (ov == bestv && ow < bw))) {
          bestv = ov;
          bw = ow;
          bl = ol;
          bviol = oviol;
        }
      }
      if constexpr (E == 1) {
        if (__ballot(ordered) != 0) {
          // ordered mode: DistanceResult::update over the evaluated pairs in stack order -- a pair is visited if its bound is below the
          // minimum of its turn, and lowers it only when strictly smaller
          T run = mind;
          int win = -1, winl = -1;
          const T v0 = leaf_eval[0] ? l_val[li[0]] : big;
          for (int s = 0; s < SEG; ++s) {
            const int src = q * SEG + s;
            const bool ev = __shfl(int(leaf_eval[0]), src) != 0;
            const T vs = __shfl(v0, src), ds = __shfl(db[0], src);
            const int ls = __shfl(li[0], src);
            if (ev && !(ds >= T(0) && ds >= run) && vs < run) {
              run = vs;
              win = s;
              winl = ls;
            }
          }
          if (ordered) {
            bestv = run;
            bw = win;
            bl = winl;
            bviol = 0;
          }
        }
      }
      if (bl >= 0) {
        jw = bw;
        mind = bestv;
        // Every decision of the pool is the sequential walk's as long as the pair of the minimum is one that walk visits, or at least
        // stands under no bound above its own distance (then the minimum the sequential walk holds behind it is not larger).  Otherwise:
        flagged = flagged || bviol != 0;
        if (j == 0) q_win[q] = bl;
      }
      sync();
      if (lane < nle && q_win[my_slot] == lane) {  // the lane that evaluated the new minimum's pair still holds its witness
        q_fb[my_slot][0] = int(lb1);
        q_fb[my_slot][1] = int(lb2);
        T* w6 = q_wit[my_slot];
        w6[0] = P.x; w6[1] = P.y; w6[2] = P.z; w6[3] = Qp.x; w6[4] = Qp.y; w6[5] = Qp.z;
      }
      POOL_C(11, 1);
      POOL_C(12, nle);
    }
    POOL_T(2);
    // ---- the windows written back, in order: deeper entries first, a split pair as its two children (nearer one on top)
    // (HFCL_POOL_DROP_DEAD, default 1) A child whose bound cannot beat the minimum any more is not written at all: the test is the one the
    // next trip would put to it at the head of its window -- with this trip's minimum, and behind the pair of the minimum exactly when its
    // parent is --, so the walk is the same and the windows hold live entries only.
    int cnt[E], cnt_l = 0, later_l = 0;
    bool keep_a[E], keep_c[E];
#pragma unroll
    for (int u = 0; u < E; ++u) {
      const bool later = (j * E + u) < w && (jw >= 0 ? (j * E + u) > jw : idx[u] < p);  // stands behind the pair of the minimum
      keep_a[u] = keep_c[u] = split[u];
#if HFCL_POOL_DROP_DEAD
      if (split[u]) {
        keep_a[u] = ordered ? !(d1[u] > cut) : !(later ? d1[u] >= mind : d1[u] > mind + margin);
        keep_c[u] = ordered ? !(d2[u] > cut) : !(later ? d2[u] >= mind : d2[u] > mind + margin);
      }
#endif
      cnt[u] = split[u] ? (keep_a[u] ? 1 : 0) + (keep_c[u] ? 1 : 0) : (((is_leaf[u] && !leaf_eval[u]) || held[u]) ? 1 : 0);
      cnt_l += cnt[u];
      later_l += later ? cnt[u] : 0;
    }
    {
      int pos = base_i + lane_sum(cnt_l, deeper);
#pragma unroll
      for (int u = E - 1; u >= 0; --u) {
        if (split[u]) {
          const bool c_first = d2[u] < d1[u];  // visit (c1, c2) first when it is strictly nearer: the other one lies deeper
          // the children's chains: the parent's largest bound (its own, or what it inherited) where the child's own is below it
          const T pc = ch[u] != 0.0f ? T(ch[u]) : db[u];
          const float pcw = ch[u] != 0.0f ? ch[u] : chain_up(db[u]);
          const float ca = d1[u] >= pc ? 0.0f : pcw, cc = d2[u] >= pc ? 0.0f : pcw;
          int at = pos;
          if (c_first ? keep_a[u] : keep_c[u]) {
            st_x[q][at] = c_first ? xa[u] : xc[u];
            st_d[q][at] = c_first ? d1[u] : d2[u];
            st_c[q][at] = c_first ? ca : cc;
            ++at;
          }
          if (c_first ? keep_c[u] : keep_a[u]) {
            st_x[q][at] = c_first ? xc[u] : xa[u];
            st_d[q][at] = c_first ? d2[u] : d1[u];
            st_c[q][at] = c_first ? cc : ca;
          }
        } else if (cnt[u] == 1) {
          st_x[q][pos] = x[u];
          st_d[q][pos] = db[u];
          st_c[q][pos] = ch[u];
        }
        pos += cnt[u];
      }
    }
    // the marker: what stands behind the pair of the minimum now
    p = (jw >= 0 ? base_i : min(p, base_i)) + lane_sum(later_l, segm);
    sp = base_i + lane_sum(cnt_l, segm);
    if (ordered && mind <= pooled) sp = 0;  // the first pair at the final distance in the reference's order: nothing behind it replaces it
    sync();
    POOL_T(3);
    if (active && sp == 0) {  // this walk is over
      if (E == 1 && (flagged || spill.rerun_all) && !ordered && spill.rerun_count) {
        // not written: the slot takes the record again and walks it in the reference's order.  What the pooled pass found bounds that
        // walk: the sequential minimum is not below this one and within the arithmetic's slack of it, so no entry whose bound exceeds
        // it by a multiple of that slack can hold the reference's pair.
        pooled = mind;
        cut = mind + T(64) * margin;
        redo = true;
        if (j == 0) atomicAdd(spill.rerun_count, 1u);
      } else if (j == 0) {
#ifdef HFCL_POOL_PROF
        atomicAdd(&pool_prof[13], 1ull);
#endif
        const Pose<T> tf1 = load_pose(io.tf1, pair);
        const T* w6 = q_wit[q];
        PairOut<T> o;
        o.distance = mind;
        o.normal = mk<T>(nanv, nanv, nanv);  // not set by the reference on this path (traversal_node_bvhs.h:454,465)
        o.p1 = xform(tf1, mk<T>(w6[0], w6[1], w6[2]));  // postprocess(): model-1 frame -> world
        o.p2 = xform(tf1, mk<T>(w6[3], w6[4], w6[5]));
        o.gjk_status = GJK_DID_NOT_RUN;
        o.epa_status = EPA_DID_NOT_RUN;
        o.gjk_iters = o.epa_iters = 0;
        store_bvh_record(io, pair, o, mind <= T(0) ? 0x80000000u : 0u, q_fb[q][0], q_fb[q][1], false);
      }
      active = false;
      ordered = false;
    }
  }
}

// =======================================================================================
// launcher (hfcl_launch.hpp)
// =======================================================================================
template <typename T>
void launch_bvh_distance(int grid, hipStream_t st, const Work& wk, const LibView<T>& lv, const BvhView<T>& bv, const IO<T>& io, const QParams<T>& q, BvhSpill spill) {
  if (spill.wide) {
    if (spill.slab) grid = std::min(grid, int(spill.max_blocks) * (BVH_BLOCK / BVHD_BLOCK));
    hipLaunchKernelGGL((k_bvh_distance<T, true>), dim3(grid), dim3(BVHD_BLOCK), 0, st, wk, lv, bv, io, q, spill);
  } else {
    hipLaunchKernelGGL((k_bvh_distance<T, false>), dim3(grid), dim3(BVHD_BLOCK), 0, st, wk, lv, bv, io, q, spill);
    const int cgrid = std::max(1, int(std::min<uint32_t>(wk.n, spill.max_blocks)));  // a wave per suspended walk, up to what the chip holds
    // the pool: a wave per POOL_Q walks, up to what the chip holds (`grid` is the lanes' grid, a block of BVHD_BLOCK queries each:
    // with it a 100k-query batch started 1 563 of the 2 048 waves the chip holds -- 30.5 ms instead of 25.4, profiles/r04_h)
    const int n_est = int(std::min<uint32_t>(wk.n, 0x7FFFFFFFu));  // pairs of the batch (the lanes' grid is capped, hfcl_host.hip: blocks_for)
    if (spill.budget && spill.pool && n_est > POOL_BIG_BATCH)
      hipLaunchKernelGGL((k_bvh_distance_pool<T, 8>), dim3(std::max(1, std::min((n_est + 7) / 8, int(spill.max_blocks)))), dim3(64), 0, st, wk, lv, bv, io, spill);
    else if (spill.budget && spill.pool)
      hipLaunchKernelGGL((k_bvh_distance_pool<T, POOL_Q>), dim3(std::max(1, std::min((n_est + POOL_Q - 1) / POOL_Q, int(spill.max_blocks)))), dim3(64), 0, st, wk, lv, bv, io, spill);
    else if (spill.budget)
      hipLaunchKernelGGL((k_bvh_distance_coop<T>), dim3(cgrid), dim3(64), 0, st, wk, lv, bv, io, spill);
  }
}
#ifdef HFCL_POOL_PROF
extern "C" int hfcl_debug_pool_prof(unsigned long long* out16, int reset) {
  unsigned long long z[16] = {0};
  if (hipMemcpyFromSymbol(out16, HIP_SYMBOL(pool_prof), sizeof(z)) != hipSuccess) return -1;
  if (reset && hipMemcpyToSymbol(HIP_SYMBOL(pool_prof), z, sizeof(z)) != hipSuccess) return -1;
  return 0;
}
#endif
template void launch_bvh_distance<float>(int, hipStream_t, const Work&, const LibView<float>&, const BvhView<float>&, const IO<float>&, const QParams<float>&, BvhSpill);
template void launch_bvh_distance<double>(int, hipStream_t, const Work&, const LibView<double>&, const BvhView<double>&, const IO<double>&, const QParams<double>&, BvhSpill);
