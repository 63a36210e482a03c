// hfcl_k_bvh.hip -- BVHModel<OBBRSS> kernels (mesh x mesh collide / distance, mesh x solid) and top-level TriangleP pairs.
#include <algorithm>

#include "hfcl_dev.hpp"
#include "hfcl_launch.hpp"

// This file is compiled three times (Makefile): hfcl_k_bvh.o, HFCL_BVH_PART = 1 -- mesh x mesh collide(), top-level triangle pairs and their
// launchers --, hfcl_k_bvhc.o, HFCL_BVH_PART = 3 -- mesh x solid collide(), its leaves' solvers without contraction (HFCL_LEAF_CONTRACT_OFF,
// hfcl_bvh.hpp: every inlined copy of a leaf is then the same arithmetic; the box tests keep hipcc's default) -- and
// hfcl_k_bvhs.o, HFCL_BVH_PART = 2 -- mesh x solid distance() -- WITHOUT contraction of a*b+c: the reported triangle hangs on
// comparisons of distances that are equal or an ulp apart (triangles that share the closest vertex or edge), and with the
// reference's arithmetic the ids, distances and witness points are the oracle's (profiles/r05_c: ids 77-93 % -> 100 % equal,
// distances 29-71 % -> 100 % bit-equal), as for mesh x mesh (hfcl_k_bvhd.hip).  Kernels both parts launch carry the part as a
// template argument (one host stub per object).  0: everything in one object (tools).
#ifndef HFCL_BVH_PART
#define HFCL_BVH_PART 0
#endif
#define HFCL_BVH_COLLIDE_PART (HFCL_BVH_PART != 2)                          // either collide() part
#define HFCL_BVH_MESH_PART (HFCL_BVH_PART == 0 || HFCL_BVH_PART == 1)      // mesh x mesh collide(), triangle pairs
#define HFCL_BVH_SOLID_PART (HFCL_BVH_PART == 0 || HFCL_BVH_PART == 3)     // mesh x solid collide()
#define HFCL_BVH_DISTANCE_PART (HFCL_BVH_PART == 0 || HFCL_BVH_PART == 2)  // mesh x solid distance()

// ---------------------------------------------------------------------------------------
// k_bvh_collide: BVHModel<OBBRSS> x BVHModel<OBBRSS> collide().
// Traversal = collisionRecurse (src/traversal/traversal_recurse.cpp:44-85) with the recursion
// flattened into a per-lane LDS stack; children are pushed right-then-left so they pop in the
// reference's order, and the walk ends as soon as num_max_contacts contacts exist (canStop()).
//
// Long traversals are cut into tasks.  Queries differ by two orders of magnitude in length (cfg4: 135 BV tests on
// average, > 2000 for the longest), and with 1.5 queries per resident lane the kernel used to last as long as its longest
// query.  Now a unit of work (level 0: a query; deeper levels: a task = one pending (b1, b2) subtree pair) that has used
// up its step budget -- or whose LDS stack is full -- *suspends*: every entry of its stack becomes a task of the next
// level (in DFS order: top of the stack first), its state so far is parked as a summary, and the lane takes new work.
// The levels run as separate launches; afterwards k_bvh_combine folds the children of every suspended unit back, deepest
// level first, exactly as the sequential walk would have seen them:
//   * the walk's running lower bound is a minimum over all its BV / leaf events: order-free;
//   * the reported witness points are those of the LAST leaf that lowered the bound when it was visited.  Within a task
//     those leaves have decreasing values, so only its last one can also lie below the bound accumulated before the task:
//     a task reports that leaf (cand_val, np1, np2, nn) and its overall minimum (dlb, rec_dist), nothing else;
//   * a contact ends the walk: children after the first one with a contact are ignored (their work was speculative).
// Contact lists (num_max_contacts > 1) need the running count of the whole query and run unsplit.
// ---------------------------------------------------------------------------------------
template <typename T>
__device__ __forceinline__ BvhSum<T>* bvh_sum(const BvhSplit& sp, uint32_t slot) { return reinterpret_cast<BvhSum<T>*>(sp.sums) + slot; }

// ---------------------------------------------------------------------------------------
// Mesh x solid with one query per LANE: pieces of the SOLID form of k_bvh_collide (below) and of k_bvh_shape_finish.
// ---------------------------------------------------------------------------------------
#ifndef HFCL_LANE_SOLID_BATCH
#define HFCL_LANE_SOLID_BATCH 4
#endif
template <typename T>
struct LaneSolid {  // support of the solid in its own frame, by one lane (ConvexBase: serial scan, first maximum wins)
  DShape<T> s;
  const T* v;
  __device__ __forceinline__ V3<T> operator()(const V3<T>& d) const {
    if (s.kind != K_CONVEX) return prim_support(s, d);
    uint32_t best = 0;
    T bd = v[0] * d.x + v[1] * d.y + v[2] * d.z;
    uint32_t i = 1;
#if HFCL_LANE_SOLID_BATCH
    // HFCL_LANE_SOLID_BATCH vertices (four: 96 consecutive bytes in fp64) fetched together, then their products in the scan's order: one vertex per trip of a loop
    // of unknown length was one exposed load latency per vertex, 32 per support call of a convex solid
    constexpr uint32_t LB = HFCL_LANE_SOLID_BATCH;
    for (; i + LB <= s.num_points; i += LB) {
      T a[3 * LB];
#pragma unroll
      for (int j = 0; j < int(3 * LB); ++j) a[j] = v[3 * size_t(i) + size_t(j)];
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int j = 0; j < int(LB); ++j) {
        const T x = a[3 * j] * d.x + a[3 * j + 1] * d.y + a[3 * j + 2] * d.z;
        if (x > bd) {
          bd = x;
          best = i + uint32_t(j);
        }
      }
    }
#endif
    for (; i < s.num_points; ++i) {
      const T x = v[3 * size_t(i)] * d.x + v[3 * size_t(i) + 1] * d.y + v[3 * size_t(i) + 2] * d.z;
      if (x > bd) {
        bd = x;
        best = i;
      }
    }
    return mk<T>(v[3 * size_t(best)], v[3 * size_t(best) + 1], v[3 * size_t(best) + 2]);
  }
};
template <typename T>
__device__ __forceinline__ void emit_shape_contact(const BvhParams& bp, uint32_t pair, bool swapped, int prim, T distance,
                                                   const V3<T>& p1, const V3<T>& p2, const V3<T>& nn) {
  if (!bp.contacts) return;
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
}
template <typename T>
__device__ __forceinline__ void store_unsupported(const IO<T>& io, uint32_t pair) {
  auto r = io.out[pair];
  memset(&r, 0, sizeof(r));
  r.status = 0x80000000u;
  io.out[pair] = r;
}
// Is the work of the task (parent slot p, position o) still needed?  Not if an earlier sibling -- of it or of any of its
// ancestors -- has found a contact: the sequential walk would have ended there.
template <typename T>
__device__ __forceinline__ bool bvh_moot(const BvhSplit& split, uint32_t p, uint32_t o) {
  for (int hop = 0; hop < BVH_MAX_LEVELS + 1 && p != 0xFFFFFFFFu; ++hop) {
    const BvhSum<T>* ps = bvh_sum<T>(split, p);
    if (__hip_atomic_load(&ps->contact_order, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < o) return true;
    o = ps->order;
    p = ps->parent;
  }
  return false;
}

#ifndef HFCL_BVH_PREFETCH
#define HFCL_BVH_PREFETCH 1  // cfg4 100k: 7.5 -> 6.7 ms, 1M: 61.6 -> 67.6 M q/s (profiles/r02_r); touching the sibling as well gave nothing more
#endif
#ifndef HFCL_WPE_BVH_COLLIDE
#define HFCL_WPE_BVH_COLLIDE 2  // two waves per SIMD: the walk waits for its node gathers most of the time (profiles/r02_m)
#endif
// FILT (fp64 only): the separating-axis test runs as an fp32 filter on 64-byte node records (hfcl_bvh.hpp: obb_filter, with
// its error analysis) and the fp64 test of the reference -- on the 128-byte records, with the relative pose rebuilt from the
// query's poses -- only where the filter cannot prove what that test would do: 4 % of the steps on cfg4 (3.9 %: a
// disjoint pair whose value could lower the running bound, i.e. the record lows of the bound; 0.4 %: a quantity within the
// error bound of its threshold).  A lane that needs the fp64 test parks (like a lane waiting for its leaf test) and the wave
// runs it for all parked lanes at once.  Decisions and reported numbers are the fp64 test's throughout; the walk's
// persistent state is the fp32 relative pose (13 registers instead of 24).
// MEASURED SLOWER than the plain fp64 kernel and therefore off by default (HFCL_BVH_FILTER=1 selects it; profiles/r03_b):
// cfg4 100k queries 6.9 against 5.3 ms, 1M queries 53 against 68 M q/s.  The fp32 test is ~1.3x the instructions of the
// fp64 one (bounds, rebuilt third axes) at 1.8x the issue rate, the kernel keeps the registers of its fp64 leaf phase (256:
// two waves per SIMD either way; forced to three it spills 396 B per lane and takes 10.8 ms), and the step is a gather
// round trip in both forms.  The walk ALONE (no leaf tests, no fp64 re-tests) fits 152 registers = three waves per SIMD
// and does 29 G steps/s against the 12 G/s of the full kernel: the headroom is in separating the walk from the fp64
// phases, not in the arithmetic of the test.
#ifndef HFCL_WPE_BVH_FILT
#define HFCL_WPE_BVH_FILT 2
#endif
// The leaf of the SOLID form as a call: inlined (HFCL_SOLID_LEAF_OUTLINE=0), the leaf's GJK over every support function
// shares the register allocation of the walk -- 7 % slower on tools/mesh_solid_bench.py (profiles/r03_i).
#ifndef HFCL_SOLID_LEAF_OUTLINE
#define HFCL_SOLID_LEAF_OUTLINE 1
#endif
template <typename T>
struct SolidLeafOut {
  T distance;
  V3<T> p1, p2, n, guess;
};
template <typename T>
struct SolidLeafIn {
  const T* mesh_verts;      // of this model
  const uint32_t* tri;      // the triangle's three vertex ids
  const DShape<T>* shapes;
  const T* lib_verts;
  const decltype(IO<T>::tf1) pose_m;  // pose arrays of the mesh / the solid
  const decltype(IO<T>::tf1) pose_s;
  ShapeDeferItem<T>* defer;
  uint32_t* defer_count;
  uint32_t defer_cap;
  uint32_t pair, solid_id, prim, parent, order;
  T bound;            // distance(): min_distance before this leaf, and the triangle it belongs to
  int32_t prev_prim;
};
template <typename T, class PS>
__device__ __noinline__ bool solid_leaf_call(const SolidLeafIn<T> in, const QParams<T>* qp, const PS ps, const V3<T> guess_in, SolidLeafOut<T>* out,
                                             uint32_t* slot_out = nullptr) {
  const QParams<T> q = *qp;
  auto vtx = [&](uint32_t i) { return mk<T>(in.mesh_verts[3 * size_t(i)], in.mesh_verts[3 * size_t(i) + 1], in.mesh_verts[3 * size_t(i) + 2]); };
  const V3<T> ta = vtx(in.tri[0]), tb = vtx(in.tri[1]), tc = vtx(in.tri[2]);
  LaneSolid<T> solid;
  solid.s = in.shapes[in.solid_id];
  solid.v = in.lib_verts + 3 * size_t(solid.s.vertex_offset);
  auto tfm_of = [&]() { return load_pose(in.pose_m, in.pair); };
  auto tfs_of = [&]() { return load_pose(in.pose_s, in.pair); };
  const MDiff<T> sMt = make_mdiff(tfs_of(), tfm_of());
  V3<T> guess = guess_in;
  ShapeDeferItem<T> item;
  const bool to_epa = mesh_shape_leaf_lane(ta, tb, tc, sMt, tfm_of, tfs_of, solid.s, solid, swept_radius(solid.s), q, guess, ps, out->distance,
                                           out->p1, out->p2, out->n, item);
  if (to_epa && in.defer) {  // (defer == nullptr: the caller only asks whether the leaf needs EPA)
    item.seed.pair = in.pair;
    item.prim = in.prim;
    item.parent = in.parent;
    item.order = in.order;
    item.bound = in.bound;
    item.prev_prim = in.prev_prim;
    const uint32_t slot = atomicAdd(in.defer_count, 1u);
    if (slot < in.defer_cap) in.defer[slot] = item;  // (the host sizes the queue for one item per unit of the batch; never past its end)
    if (slot_out) *slot_out = slot;
  }
  out->guess = guess;
  return to_epa;
}

// the triangle pair of a mesh x mesh leaf as a call (k_bvh_coop; k_bvh_collide with HFCL_BVH_LEAF_OUTLINE=1)
#ifndef HFCL_BVH_LEAF_OUTLINE
#define HFCL_BVH_LEAF_OUTLINE 0
#endif
template <typename T>
struct TriLeafOut {
  T distance;
  V3<T> p1, p2, n;
};
template <typename T, class PS>
__device__ __noinline__ void tri_leaf_call(const T* v1, const uint32_t* t1, const T* v2, const uint32_t* t2, const decltype(IO<T>::tf1) pose1,
                                           const decltype(IO<T>::tf1) pose2, uint32_t pair, const QParams<T>* qp, const PS ps, TriLeafOut<T>* out) {
  const QParams<T> q = *qp;
  auto vtx = [](const T* v, uint32_t i) { return mk<T>(v[3 * size_t(i)], v[3 * size_t(i) + 1], v[3 * size_t(i) + 2]); };
  TriSupport<T> tri;
  {
    const Pose<T> tf1 = load_pose(pose1, pair);
    tri.p1 = xform(tf1, vtx(v1, t1[0]));
    tri.p2 = xform(tf1, vtx(v1, t1[1]));
    tri.p3 = xform(tf1, vtx(v1, t1[2]));
  }
  {
    const Pose<T> tf2 = load_pose(pose2, pair);
    tri.q1 = xform(tf2, vtx(v2, t2[0]));
    tri.q2 = xform(tf2, vtx(v2, t2[1]));
    tri.q3 = xform(tf2, vtx(v2, t2[2]));
  }
  int gst, git;
  out->distance = tri_tri_distance(tri, q.gjk, q.guess_mode == HFCL_GUESS_CACHED, mk<T>(q.guess[0], q.guess[1], q.guess[2]), out->p1, out->p2,
                                   out->n, gst, git, (V3<T>*)nullptr, ps);
}
#ifndef HFCL_BVH_PARK_MAX
#define HFCL_BVH_PARK_MAX 32  // lanes waiting for a leaf test or an fp64 re-test that end the BV phase of a wave
#endif
// SOLID: BVHModel<OBBRSS> x convex solid (bucket B_BVHSHAPE, either operand order) walked by the same machinery, one query
// per lane.  The second "tree" is the solid's fitted OBB (k_shape_obb wrote its products with the mesh pose, ObbQuery, per
// query), so an entry is a node of the mesh, every step splits the mesh node (collisionRecurse with a leaf second node) and
// a leaf is ShapeShapeDistance<TriangleP, S>: closed form or per-lane GJK.  A leaf that needs EPA is a contact on the
// host's word (mesh_shape_lane_request) and ends the unit like any contact; its numbers come later, from
// k_bvh_shape_finish.  The 16-lane group kernel this replaces where the request allows walked the long queries of a
// batch alone (the steps per query have a heavy tail: median 1, mean 60, maximum > 3000 with > 1000 leaf tests on
// tools/mesh_solid_bench.py's scenes) -- here they suspend into tasks like the long mesh x mesh queries.
template <typename T, bool WIDE, bool FILT = false, bool SOLID = false>
__global__ void __launch_bounds__(BVH_BLOCK) __attribute__((amdgpu_waves_per_eu(FILT ? HFCL_WPE_BVH_FILT : HFCL_WPE_BVH_COLLIDE, 8))) k_bvh_collide(Work wk, LibView<T> lib, BvhView<T> bv, IO<T> io, QParams<T> q,
                                                          BvhParams bp, T break_distance2, BvhSplit split, BvhSpill spill) {
  static_assert(!FILT || sizeof(T) == 8, "the fp32 filter stands in front of the fp64 test");
  static_assert(!SOLID || (!WIDE && !FILT), "mesh x solid: 32-bit node ids in the narrow entry, plain tests");
  typedef BvhEntry<WIDE> EN;
  typedef typename EN::E E;
  // the filter form is built for three waves per SIMD: six 128-thread blocks per CU = 20 LDS allocation units (25 600 B) each
  constexpr int STACK = FILT ? (WIDE ? BVH_STACK_FILT / 2 : BVH_STACK_FILT) : EN::STACK, HALF = STACK / 2;
  __shared__ E stack[STACK][BVH_BLOCK];
  __shared__ T w0_slab[W0Lds<T, BVH_BLOCK>::WORDS];  // witness payload of the leaf tests' GJK simplex (hfcl_dev.hpp: W0Lds)
  const W0Lds<T, BVH_BLOCK> leaf_ps{w0_slab + threadIdx.x};
  // this lane's slab of spilled entries (WIDE only)
  E* const slab = WIDE && spill.slab ? reinterpret_cast<E*>(spill.slab) + size_t(blockIdx.x * BVH_BLOCK + threadIdx.x) * spill.cap : nullptr;
  uint32_t nspill = 0;
  const uint32_t level = split.level;
  const uint32_t unit0 = level ? split.ctr[BVH_CTR_LEVEL0 + level - 1] : 0u;  // first task of this level
  const uint32_t cnt = level ? min(split.ctr[BVH_CTR_LEVEL0 + level], split.cap) - min(unit0, split.cap) : wk.counts[SOLID ? B_BVHSHAPE : B_BVH];
  if (cnt == 0u) return;  // (no unit of this kind in the batch / on this level: no wave draws a ticket)
  uint32_t* const ticket = &wk.counts[SOLID ? CTR_SHAPE_TICKET : B_COUNT + 2];
  auto ent_first = [](E e) -> uint32_t { return SOLID ? uint32_t(e) : EN::first(e); };  // SOLID: the entry is the mesh node
  const uint32_t budget = split.budget;  // steps a unit may take before it suspends (0: never)
  const int tid = threadIdx.x, lane = tid & 63;
  const T nanv = Lim<T>::nan();
  // Per-lane unit state.  The lanes of a wave do not advance through the batch in lockstep: a lane whose traversal is
  // over parks its result (`pending`) and, as soon as BVH_REFILL_MIN lanes of the wave are idle, all of them write
  // their records and take the next units from a global ticket counter.
  constexpr int refill_min = BVH_REFILL_MIN;
  bool live = false, pending = false, exhausted = false;  // exhausted is wave-uniform
  uint32_t pair = 0, unit = 0, steps = 0;
  // What a lane keeps in registers across a walk is what the BV tests need: the relative pose and the running bounds.
  // The poses themselves (leaf tests only: ~5 per query) are re-read there, the witness of the bound (p1, p2, normal:
  // updated a handful of times) lives where it will be read -- the query's record, or the task's summary.
  DMesh m1 = {0, 0, 0, 0}, m2 = {0, 0, 0, 0};
  typedef typename std::conditional<FILT, float, T>::type TR;  // precision the walk keeps the relative pose in
  M3<TR> RT_R;
  V3<TR> RT_T;
  float t0mag = 0.f;        // FILT: >= |RT_T|_1 (error bound of the filter)
  // FILT: a lane that needs the fp64 test parks with need_exact = 1 (the pair `pend`, which the filter could not decide)
  // or 2 (the candidate `cand`: a disjoint pair whose value may be the minimum of the bound, known to [cand_lo, cand_hi];
  // its fp64 value is computed only when something is compared with it -- a leaf test, a second candidate it cannot be
  // told apart from, the end of a contact-free walk, a suspension)
  int need_exact = 0;
  bool pend_first = false, has_cand = false;
  E pend = 0, cand = 0;
  float cand_lo = 0.f, cand_hi = 0.f;
  const float marginf = float(q.security_margin), bd2f = float(break_distance2);
  int sp = 0;
  bool overflow = false;
  uint32_t ncontacts = 0;
  T dlb = Lim<T>::max(), rec_dist = Lim<T>::max(), cand_val = Lim<T>::max();
  int fb1 = -1, fb2 = -1;
  bool have_leaf = false;
  uint32_t lb1 = 0, lb2 = 0;
  uint32_t my_parent = 0xFFFFFFFFu, my_order = 0;  // (tasks) where this unit hangs
  // SOLID: RT_R / RT_T hold the ObbQuery's M / V, q_ext its extent; the solid, the operand order, the solver's cached guess
  V3<T> q_ext = mk<T>(T(0), T(0), T(0)), guess = mk<T>(T(1), T(0), T(0));
  uint32_t solid_id = 0;
  bool swapped = false;
  auto witness_store = [&](const V3<T>& p1, const V3<T>& p2, const V3<T>& n) {  // the witness of the bound, in place
    if (level) {
      BvhSum<T>* sm = bvh_sum<T>(split, split.n_queries + unit);
      sm->np1 = p1;
      sm->np2 = p2;
      sm->nn = n;
    } else {
      store_witness(io, pair, p1, p2, n);
    }
  };
  auto write_sum_head = [&](uint32_t slot, uint32_t first_child, uint32_t n_child, uint32_t flags) {  // all but np1 / np2 / nn
    BvhSum<T>* sm = bvh_sum<T>(split, slot);
    sm->contact_order = 0xFFFFFFFFu; sm->parent = my_parent; sm->order = my_order; sm->pad_ = 0;
    sm->dlb = dlb; sm->rec_dist = rec_dist; sm->cand_val = cand_val;
    sm->fb1 = fb1; sm->fb2 = fb2;
    sm->ncontacts = ncontacts; sm->first_child = first_child; sm->n_child = n_child; sm->flags = flags;
  };
  auto flush = [&]() {  // the unit this lane finished: a query's record, or a task's summary (the witness is in place)
    if (level)
      write_sum_head(split.n_queries + unit, 0u, 0u, overflow ? BVH_SUM_OVERFLOW : 0u);
    else {
      store_bvh_record_head(io, pair, rec_dist, ncontacts, fb1, fb2, overflow);
      if constexpr (SOLID) write_guess<T>(io, pair, guess, 0, 0);
    }
  };
  auto moot = [&](uint32_t p, uint32_t o) -> bool { return bvh_moot<T>(split, p, o); };
  // a contact in this task ends the walk of every unit above it at this child's position
  auto report_contact = [&](uint32_t p, uint32_t o) {
    for (int hop = 0; hop < BVH_MAX_LEVELS + 1 && p != 0xFFFFFFFFu; ++hop) {
      BvhSum<T>* ps = bvh_sum<T>(split, p);
      atomicMin(&ps->contact_order, o);
      o = ps->order;
      p = ps->parent;
    }
  };
  // Turn the `n_extra` entries in `extra` (children about to be pushed: first one on top) and the whole stack into tasks
  // of the next level and park this unit.  false: no room in the task table (the unit then simply goes on).
  // SOLID: every entry is EXPANDED ONCE on the way out -- its box is tested here, and its two children (or the bound, if the
  // boxes are disjoint: a child that is already finished) take its place among the tasks.  A suspended stack holds subtrees
  // of geometrically decreasing size (the bottom entry is the sibling of the walk's first step: half the tree), so plain
  // suspension halves the longest chain per level at best; with the expansion it is quartered (profiles/r03_i: mesh x solid
  // 6.2 -> 4.6 ms per 100k sphere queries, nine levels -> six; the same on mesh x mesh stacks LOSES, 4.97 -> 5.28 ms per
  // 100k cfg4 queries: twice the tasks, and a task costs its set-up).
  auto suspend = [&](uint32_t ea, uint32_t eb, int n_extra) -> bool {
    const uint32_t n_ent = uint32_t(sp + n_extra);
    if (WIDE || !split.can_suspend || n_ent == 0) return false;  // (tasks carry 16-bit node ids)
    const bool expand = SOLID && !split.coop;  // (a suspended stack that k_bvh_shape_coop continues stays as it is)
    const uint32_t n_slots = expand ? 2u * n_ent : n_ent;
    const uint32_t first = atomicAdd(&split.ctr[BVH_CTR_TASKS], n_slots);
    if (first + n_slots > split.cap) {  // table full: the slots taken become no-ops for the next level
      for (uint32_t j = first; j < min(first + n_slots, split.cap); ++j) split.tasks[j] = BvhTask{0u, 0u, 0xFFFFFFFFu, 0u};
      steps = 0;  // the unit goes on; it asks again after another budget of steps, not at every step (a full table stays full)
      return false;
    }
    uint32_t my_slot;
    if (level) {
      my_slot = split.n_queries + unit;
    } else {
      my_slot = atomicAdd(&split.ctr[BVH_CTR_SUSPENDED], 1u);  // < n_queries: one per query at most
      split.suspended[my_slot] = pair;
    }
    uint32_t j = first, o = 0;
    auto as_they_are = [&]() {
      if (n_extra > 0) split.tasks[j++] = BvhTask{pair, my_slot, ea, o++};
      if (n_extra > 1) split.tasks[j++] = BvhTask{pair, my_slot, eb, o++};
      for (int k = sp - 1; k >= 0; --k) split.tasks[j++] = BvhTask{pair, my_slot, uint32_t(stack[k][tid]), o++};  // DFS order: top first
    };
    if (SOLID && !expand) {
      as_they_are();
    } else if constexpr (SOLID) {
      ObbQuery<T> oq;
      oq.M = RT_R;
      oq.V = RT_T;
      oq.ext = q_ext;
      for (int k = int(n_ent) - 1; k >= 0; --k) {  // DFS order: the children about to be pushed, then the stack from the top
        const uint32_t e = k >= sp ? (k == int(n_ent) - 1 ? ea : eb) : uint32_t(stack[k][tid]);
        const DNode<T>* const np = bv.nodes + m1.node_off + e;
        const int32_t fc = np->first_child;
        if (fc < 0) {
          split.tasks[j++] = BvhTask{pair, my_slot, e, o++};
          continue;
        }
        const DNode<T> n1 = *np;
        T sq;
        if (obb_disjoint_q(oq, n1, q.security_margin, break_distance2, sq)) {
          // a child that is finished: its summary is the bound (updateDistanceLowerBoundFromBV, folded in at its place)
          BvhSum<T>* sm = bvh_sum<T>(split, split.n_queries + j);
          const T nd = hsqrt(sq);
          sm->dlb = nd;
          sm->rec_dist = nd + q.security_margin;
          sm->cand_val = Lim<T>::max();
          sm->fb1 = sm->fb2 = -1;
          sm->ncontacts = 0; sm->first_child = 0; sm->n_child = 0; sm->flags = 0;
          sm->contact_order = 0xFFFFFFFFu; sm->parent = my_slot; sm->order = o; sm->pad_ = 0;
          split.tasks[j++] = BvhTask{pair, my_slot, 0xFFFFFFFFu, o++};
        } else {
          split.tasks[j++] = BvhTask{pair, my_slot, uint32_t(fc), o++};
          split.tasks[j++] = BvhTask{pair, my_slot, uint32_t(fc) + 1u, o++};
        }
      }
      for (uint32_t u = j; u < first + n_slots; ++u) split.tasks[u] = BvhTask{0u, 0u, 0xFFFFFFFFu, 0u};  // slots not needed
    } else {
      as_they_are();
    }
    write_sum_head(my_slot, first, j - first, BVH_SUM_SUSPENDED);
    if (!level) {  // a query's witness so far sits in its record: the summary needs a copy
      BvhSum<T>* sm = bvh_sum<T>(split, my_slot);
      load_witness(io, pair, sm->np1, sm->np2, sm->nn);
    }
    sp = 0;
    live = false;  // nothing to flush: the summary is written
    return true;
  };
  for (;;) {
    if (WIDE && live && !have_leaf && !need_exact && sp == 0 && nspill > 0) {  // the LDS part ran empty: take spilled entries back
      const uint32_t m = min(nspill, uint32_t(HALF));
      for (uint32_t k = 0; k < m; ++k) stack[k][tid] = slab[nspill - m + k];
      nspill -= m;
      sp = int(m);
    }
    if (FILT && live && !have_leaf && !need_exact && sp == 0 && has_cand) {  // the walk is over: does its bound matter?
      if (ncontacts == 0 && !overflow)
        need_exact = 2;  // a contact-free walk reports its bound: the candidate's value is needed
      else
        has_cand = false;
    }
    if (live && !have_leaf && !need_exact && sp == 0) {  // traversal over
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
          E entry = 0u;  // (b1 = 0, b2 = 0)
          bool valid = true;
          if (level) {
            unit = unit0 + it;
            const BvhTask t = split.tasks[unit];
            pair = t.pair;
            entry = t.entry;
            my_parent = t.parent;
            my_order = t.order;
            valid = t.entry != 0xFFFFFFFFu;
            if (valid && moot(my_parent, my_order)) {  // speculative work nobody will read: an empty summary
              valid = false;
              dlb = rec_dist = cand_val = Lim<T>::max();
              ncontacts = 0;
              overflow = false;
              write_sum_head(split.n_queries + unit, 0u, 0u, 0u);
            }
          } else {
            pair = wk.lists[size_t(SOLID ? B_BVHSHAPE : B_BVH) * wk.n + it];
          }
          if constexpr (SOLID) {
            if (valid) {
              const uint32_t id1 = wk.shape1[pair], id2 = wk.shape2[pair];
              swapped = lib.kinds[id1] != uint8_t(K_BVH);  // (shape, BVH): collide(o2, o1) then swapObjects (collision.cpp:93-108)
              solid_id = swapped ? id1 : id2;
              m1 = bv.meshes[lib.shapes[swapped ? id2 : id1].bvh_index];
              const ObbQuery<T> oq = reinterpret_cast<const ObbQuery<T>*>(wk.shape_oq)[pair];
              if (!(oq.ext.x == oq.ext.x)) {  // k_shape_obb: no OBB for this solid (swept-sphere radius, < 4 bound vertices)
                valid = false;
                store_unsupported(io, pair);
              }
              RT_R = oq.M;
              RT_T = oq.V;
              q_ext = oq.ext;
              guess = initial_guess<T>(io, q, pair);
            }
          }
          if (valid) {
            if constexpr (!SOLID) {
            const DShape<T> a = lib.shapes[wk.shape1[pair]], b = lib.shapes[wk.shape2[pair]];
            m1 = bv.meshes[a.bvh_index];
            m2 = bv.meshes[b.bvh_index];
            }
            if constexpr (!SOLID) {
              const Pose<T> tf1 = load_pose(io.tf1, pair), tf2 = load_pose(io.tf2, pair);
              const M3<T> R = tmul(tf1.R, tf2.R);  // traversal_node_setup.h:560-563
              const V3<T> t = tmul(tf1.R, tf2.t - tf1.t);
              RT_R.r0 = mk<TR>(TR(R.r0.x), TR(R.r0.y), TR(R.r0.z));
              RT_R.r1 = mk<TR>(TR(R.r1.x), TR(R.r1.y), TR(R.r1.z));
              RT_R.r2 = mk<TR>(TR(R.r2.x), TR(R.r2.y), TR(R.r2.z));
              RT_T = mk<TR>(TR(t.x), TR(t.y), TR(t.z));
              if (FILT) t0mag = (habs(float(RT_T.x)) + habs(float(RT_T.y)) + habs(float(RT_T.z))) * (1.f + 4.f * OBBF_U);
            }
            stack[0][tid] = entry;
            sp = 1;
            nspill = 0;
            steps = 0;
            overflow = false;
            ncontacts = 0;
            dlb = rec_dist = cand_val = Lim<T>::max();
            witness_store(mk<T>(nanv, nanv, nanv), mk<T>(nanv, nanv, nanv), mk<T>(nanv, nanv, nanv));
            fb1 = fb2 = -1;
            have_leaf = false;
            need_exact = 0;
            has_cand = false;
            live = true;
          }
        }
      }
      if (base + uint32_t(n_need) >= cnt) exhausted = true;
      continue;
    }
    // ---- BV phase: advance every lane that has no leaf test (or fp64 re-test) pending, until half the wave waits for
    // one, nobody can advance, or enough lanes ran out of work to make a refill due
    // what a BV test's outcome does to the walk (the same for the filter's verdict and the fp64 test's):
    auto on_disjoint = [&](T sq) {  // updateDistanceLowerBoundFromBV
      if (!(dlb <= T(0))) {
        const T nd = hsqrt(sq);
        if (nd < dlb) {
          dlb = nd;
          rec_dist = nd + q.security_margin;
        }
      }
    };
    auto on_overlap = [&](bool first, uint32_t b1, uint32_t b2, int32_t fc1, int32_t fc2) {
      E ea, eb;
      if (first) {
        const uint32_t c1 = uint32_t(fc1);
        ea = EN::pack(c1, b2);
        eb = EN::pack(c1 + 1, b2);
      } else {
        const uint32_t c1 = uint32_t(fc2);
        ea = EN::pack(b1, c1);
        eb = EN::pack(b1, c1 + 1);
      }
      if (WIDE && sp + 2 > STACK && slab && nspill + uint32_t(HALF) <= spill.cap) {
        // the LDS stack is full: its lower half (the entries needed last) moves to the lane's slab
        for (int k = 0; k < HALF; ++k) slab[nspill + k] = stack[k][tid];
        nspill += uint32_t(HALF);
        for (int k = HALF; k < sp; ++k) stack[k - HALF][tid] = stack[k][tid];
        sp -= HALF;
      }
      if (sp + 2 > STACK) {
        // the LDS stack is full: the whole stack (and the two children) go on as tasks; only where that is not
        // possible (contact lists, last level, task table full, slab full) the unit is flagged as overflowed
        if (!suspend(uint32_t(ea), uint32_t(eb), 2)) {
          overflow = true;
          sp = 0;
          nspill = 0;
        }
      } else {
        stack[sp++][tid] = eb;  // second child below
        stack[sp++][tid] = ea;  // first child on top
      }
    };
    for (;;) {
      const bool can_bv = live && !have_leaf && !need_exact && sp > 0;
      if (!__any(can_bv)) break;
      if (__popcll(__ballot(have_leaf || need_exact)) >= HFCL_BVH_PARK_MAX) break;
      if (!exhausted && 64 - __popcll(__ballot(live && (have_leaf || need_exact || sp > 0 || nspill > 0))) >= refill_min) break;
      if (can_bv) {
        if (FILT && has_cand && budget && steps >= budget) {
          // about to suspend (step budget): a summary carries an exact bound, so the candidate is resolved first
          need_exact = 2;
          continue;
        }
        if (budget && steps >= budget && suspend(0u, 0u, 0)) continue;
        if (level && (steps & 15u) == 15u && moot(my_parent, my_order)) {  // an earlier sibling ended the walk meanwhile
          sp = 0;
          has_cand = false;
          continue;
        }
        ++steps;
        const E e = stack[--sp][tid];
        const uint32_t b1 = ent_first(e), b2 = SOLID ? 0u : EN::second(e);
        if constexpr (FILT) {
          const DNodeF f1 = bv.fnodes[m1.node_off + b1];
          const DNodeF f2 = bv.fnodes[m2.node_off + b2];
          const bool l1 = f1.first_child < 0, l2 = f2.first_child < 0;
          if (l1 && l2) {
            have_leaf = true;
            lb1 = uint32_t(-(f1.first_child + 1));
            lb2 = uint32_t(-(f2.first_child + 1));
            if (has_cand) need_exact = 2;  // the leaf's distance is compared with the bound: the bound must be exact
          } else {
            const bool first = l2 || (!l1 && ((f1.rank & OBBF_RANK_MASK) > (f2.rank & OBBF_RANK_MASK)));  // firstOverSecond
#if HFCL_BVH_PREFETCH
            const DNodeF* const next = bv.fnodes + (first ? m1.node_off + uint32_t(f1.first_child) : m2.node_off + uint32_t(f2.first_child));
            const int32_t touched = next[0].first_child;
#endif
            float nd_lo, nd_hi;
            // argument order of the reference: overlap(RT.R, RT.T, model2.bv(b2), model1.bv(b1))
            const int verdict = obb_filter(RT_R, RT_T, t0mag, f2, f1, marginf, bd2f, nd_lo, nd_hi);
#if HFCL_BVH_PREFETCH
            asm volatile("" ::"v"(touched));
#endif
            if (verdict == OBBF_OVERLAP) {
              if (sp + 2 > STACK && has_cand && !WIDE) {  // this expansion will suspend the unit: candidate first, then again
                stack[sp++][tid] = e;
                need_exact = 2;
              } else {
                on_overlap(first, b1, b2, f1.first_child, f2.first_child);
              }
            } else if (verdict == OBBF_DISJOINT) {
              if (dlb <= T(0) || T(nd_lo) >= dlb || (has_cand && nd_lo >= cand_hi)) {
                // the value the fp64 test would report cannot lower the bound: nothing changes
              } else if (!has_cand || nd_hi < cand_lo) {
                has_cand = true;  // (a candidate that is certainly above this one is dropped)
                cand = e;
                cand_lo = nd_lo;
                cand_hi = nd_hi;
              } else {  // two candidates that cannot be told apart: the old one is resolved, this pair is looked at again
                stack[sp++][tid] = e;
                need_exact = 2;
              }
            } else {
              need_exact = 1;
              pend = e;
              pend_first = first;
            }
          }
        } else if constexpr (SOLID) {
          const DNode<T>* const np = bv.nodes + m1.node_off + b1;
          const int32_t fc = np->first_child;
          if (fc < 0) {
            have_leaf = true;
            lb1 = uint32_t(-(fc + 1));
          } else {
            const DNode<T> n1 = *np;
#if HFCL_BVH_PREFETCH
            const int32_t touched = np[fc - int32_t(b1)].first_child;  // the node popped next if the boxes overlap
#endif
            ObbQuery<T> oq;
            oq.M = RT_R;
            oq.V = RT_T;
            oq.ext = q_ext;
            T sq;
            const bool disjoint = obb_disjoint_q(oq, n1, q.security_margin, break_distance2, sq);
#if HFCL_BVH_PREFETCH
            asm volatile("" ::"v"(touched));
#endif
            if (disjoint)
              on_disjoint(sq);
            else
              on_overlap(true, b1, 0u, fc, 0);
          }
        } else {
        const DNode<T> n1 = bv.nodes[m1.node_off + b1];
        const DNode<T> n2 = bv.nodes[m2.node_off + b2];
        const bool l1 = n1.first_child < 0, l2 = n2.first_child < 0;
        if (l1 && l2) {
          have_leaf = true;
          lb1 = uint32_t(-(n1.first_child + 1));
          lb2 = uint32_t(-(n2.first_child + 1));
        } else {
          const T sz1 = sqnorm(n1.extent), sz2 = sqnorm(n2.extent);
          const bool first = l2 || (!l1 && (sz1 > sz2));  // firstOverSecond
#if HFCL_BVH_PREFETCH
          // The pair this lane pops next, if the boxes overlap, is (first child of the node that gets split, other node):
          // touch that child's record (one 128-byte line in fp64) before the separating-axis test, so that the gather of
          // the next step finds it on its way up the cache hierarchy instead of starting after the test.
          const DNode<T>* const next = bv.nodes + (first ? m1.node_off + uint32_t(n1.first_child) : m2.node_off + uint32_t(n2.first_child));
          const int32_t touched = next[0].first_child;
#endif
          T sq;
          // argument order of the reference: overlap(RT.R, RT.T, model2.bv(b2), model1.bv(b1))
          const bool disjoint = obb_disjoint(RT_R, RT_T, n2, n1, q.security_margin, break_distance2, sq);
#if HFCL_BVH_PREFETCH
          asm volatile("" ::"v"(touched));
#endif
          if (disjoint)
            on_disjoint(sq);
          else
            on_overlap(first, b1, b2, n1.first_child, n2.first_child);
        }
        }
      }
    }
    // ---- fp64 tests (FILT): pairs the filter could not decide, and candidates whose exact value is needed now
    if constexpr (FILT) {
      if (need_exact) {
        const bool is_cand = need_exact == 2;
        need_exact = 0;
        const E e = is_cand ? cand : pend;
        const uint32_t b1 = EN::first(e), b2 = EN::second(e);
        const DNode<T> n1 = bv.nodes[m1.node_off + b1];
        const DNode<T> n2 = bv.nodes[m2.node_off + b2];
        const Pose<T> tf1 = load_pose(io.tf1, pair), tf2 = load_pose(io.tf2, pair);
        const M3<T> R = tmul(tf1.R, tf2.R);  // traversal_node_setup.h:560-563
        const V3<T> t = tmul(tf1.R, tf2.t - tf1.t);
        T sq;
        const bool disjoint = obb_disjoint(R, t, n2, n1, q.security_margin, break_distance2, sq);
        if (is_cand) {
          has_cand = false;
          on_disjoint(sq);  // (the filter proved "disjoint")
        } else if (disjoint) {
          on_disjoint(sq);
        } else if (sp + 2 > STACK && has_cand && !WIDE) {  // this expansion will suspend the unit: candidate first, then again
          stack[sp++][tid] = e;
          need_exact = 2;
        } else {
          on_overlap(pend_first, b1, b2, n1.first_child, n2.first_child);
        }
      }
    }
    // ---- leaf phase (leafCollides, traversal_node_bvhs.h:184-233)
    if constexpr (SOLID) {
      if (have_leaf) {
        have_leaf = false;
        const uint32_t* t3 = bv.tris + 3 * size_t(m1.tri_off + lb1);
        const T* mv = bv.verts + 3 * size_t(m1.vert_off);
        const uint32_t kind = lib.kinds[solid_id];
        // what the leaf costs in BV tests (the unit of the step budget): a GJK run against the solid, or a closed form
        steps += (kind == uint32_t(K_SPHERE) || kind_is_flat(int(kind))) ? max(split.leaf_cost / 8u, 1u) : split.leaf_cost;
        T distance;
        V3<T> p1, p2, n;
        bool to_epa;
#if HFCL_SOLID_LEAF_OUTLINE
        {
          SolidLeafIn<T> in{mv, t3, lib.shapes, lib.verts, swapped ? io.tf2 : io.tf1, swapped ? io.tf1 : io.tf2,
                            reinterpret_cast<ShapeDeferItem<T>*>(wk.shape_defer), &wk.counts[CTR_SHAPE_DEFER], wk.shape_defer_cap, pair, solid_id, lb1, my_parent, my_order,
                            T(0), -1};
          SolidLeafOut<T> lo;
          to_epa = solid_leaf_call<T>(in, &q, leaf_ps, guess, &lo);
          distance = lo.distance;
          p1 = lo.p1;
          p2 = lo.p2;
          n = lo.n;
          guess = lo.guess;
        }
#else
        {
        auto vtx = [&](uint32_t i) { return mk<T>(mv[3 * size_t(i)], mv[3 * size_t(i) + 1], mv[3 * size_t(i) + 2]); };
        const V3<T> ta = vtx(t3[0]), tb = vtx(t3[1]), tc = vtx(t3[2]);
        LaneSolid<T> solid;
        solid.s = lib.shapes[solid_id];
        solid.v = lib.verts + 3 * size_t(solid.s.vertex_offset);
        auto tfm_of = [&]() { return load_pose(swapped ? io.tf2 : io.tf1, pair); };
        auto tfs_of = [&]() { return load_pose(swapped ? io.tf1 : io.tf2, pair); };
        const MDiff<T> sMt = make_mdiff(tfs_of(), tfm_of());  // Transform3f::inverseTimes: the mesh frame in the solid's frame
        ShapeDeferItem<T> item;
        to_epa = mesh_shape_leaf_lane(ta, tb, tc, sMt, tfm_of, tfs_of, solid.s, solid, swept_radius(solid.s), q, guess, leaf_ps,
                                      distance, p1, p2, n, item);
        if (to_epa) {
          item.seed.pair = pair;
          item.prim = lb1;
          item.parent = my_parent;
          item.order = my_order;
          const uint32_t slot = atomicAdd(&wk.counts[CTR_SHAPE_DEFER], 1u);
          if (slot < wk.shape_defer_cap) reinterpret_cast<ShapeDeferItem<T>*>(wk.shape_defer)[slot] = item;
        }
        }
#endif
        bool contact = to_epa;
        if (!to_epa) {
          const T dtc = distance - q.security_margin;
          if (dtc < dlb) {  // updateDistanceLowerBoundFromLeaf
            dlb = dtc;
            cand_val = dtc;
            rec_dist = distance;
            witness_store(swapped ? p2 : p1, swapped ? p1 : p2, swapped ? -n : n);
          }
          contact = dtc <= q.collision_distance_threshold;
        }
        if (contact) {
          if (ncontacts < bp.num_max_contacts) {
            if (ncontacts == 0) {
              fb1 = swapped ? -1 : int(lb1);
              fb2 = swapped ? int(lb1) : -1;
              if (level) report_contact(my_parent, my_order);
            }
            ++ncontacts;
            if (!to_epa) emit_shape_contact(bp, pair, swapped, int(lb1), distance, p1, p2, n);
          }
          if (to_epa || ncontacts >= bp.num_max_contacts) {  // canStop()
            sp = 0;
            nspill = 0;
          }
        }
      }
    } else
    if (have_leaf) {
      have_leaf = false;
      steps += 8;  // a triangle pair costs about as much as eight BV tests
      const uint32_t* t1 = bv.tris + 3 * size_t(m1.tri_off + lb1);
      const uint32_t* t2 = bv.tris + 3 * size_t(m2.tri_off + lb2);
      const T* v1 = bv.verts + 3 * size_t(m1.vert_off);
      const T* v2 = bv.verts + 3 * size_t(m2.vert_off);
#if HFCL_BVH_LEAF_OUTLINE
      TriLeafOut<T> tlo;
      tri_leaf_call<T>(v1, t1, v2, t2, io.tf1, io.tf2, pair, &q, leaf_ps, &tlo);
      const T distance = tlo.distance;
      const V3<T> p1 = tlo.p1, p2 = tlo.p2, n = tlo.n;
#else
      auto vtx = [](const T* v, uint32_t i) { return mk<T>(v[3 * size_t(i)], v[3 * size_t(i) + 1], v[3 * size_t(i) + 2]); };
      TriSupport<T> tri;
      {
        const Pose<T> tf1 = load_pose(io.tf1, pair);
        tri.p1 = xform(tf1, vtx(v1, t1[0]));
        tri.p2 = xform(tf1, vtx(v1, t1[1]));
        tri.p3 = xform(tf1, vtx(v1, t1[2]));
      }
      {
        const Pose<T> tf2 = load_pose(io.tf2, pair);
        tri.q1 = xform(tf2, vtx(v2, t2[0]));
        tri.q2 = xform(tf2, vtx(v2, t2[1]));
        tri.q3 = xform(tf2, vtx(v2, t2[2]));
      }
      V3<T> p1, p2, n;
      int gst, git;
      const T distance = tri_tri_distance(tri, q.gjk, q.guess_mode == HFCL_GUESS_CACHED,
                                          mk<T>(q.guess[0], q.guess[1], q.guess[2]), p1, p2, n, gst, git, (V3<T>*)nullptr, leaf_ps);
#endif
      const T dtc = distance - q.security_margin;
      if (dtc < dlb) {  // updateDistanceLowerBoundFromLeaf
        dlb = dtc;
        cand_val = dtc;
        rec_dist = distance;
        witness_store(p1, p2, n);
      }
      if (dtc <= q.collision_distance_threshold) {
        if (ncontacts < bp.num_max_contacts) {
          if (ncontacts == 0) {
            fb1 = int(lb1);
            fb2 = int(lb2);
            if (level) report_contact(my_parent, my_order);
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
        if (ncontacts >= bp.num_max_contacts) {  // canStop(): nothing else is visited
          sp = 0;
          nspill = 0;
        }
      }
    }
  }
}

// Between two levels: the tasks made so far are the next level's units; the ticket counter starts over.
#if HFCL_BVH_COLLIDE_PART
template <int PART>
__global__ void k_bvh_level_mark(Work wk, BvhSplit split, int ticket) {
  if (threadIdx.x == 0 && blockIdx.x == 0) {
    split.ctr[BVH_CTR_LEVEL0 + split.level + 1] = split.ctr[BVH_CTR_TASKS];
    wk.counts[ticket] = 0u;
    if (ticket == CTR_SHAPE_TICKET && split.level == 1) wk.counts[CTR_SHAPE_DEFER_MARK] = min(wk.counts[CTR_SHAPE_DEFER], wk.shape_defer_cap);  // (the first launch of k_bvh_shape_coop is over)
  }
}
#endif

// Fold the children of the suspended units of level `split.level` back (the children's own children are folded already).
template <typename T, int PART>
__global__ void __launch_bounds__(256) k_bvh_combine(Work wk, IO<T> io, BvhSplit split) {
  const uint32_t level = split.level;
  const uint32_t unit0 = level ? min(split.ctr[BVH_CTR_LEVEL0 + level - 1], split.cap) : 0u;
  const uint32_t cnt = level ? min(split.ctr[BVH_CTR_LEVEL0 + level], split.cap) - unit0 : split.ctr[BVH_CTR_SUSPENDED];
  for (uint32_t i = blockIdx.x * blockDim.x + threadIdx.x; i < cnt; i += gridDim.x * blockDim.x) {
    const uint32_t slot = level ? split.n_queries + unit0 + i : i;
    BvhSum<T> s = *bvh_sum<T>(split, slot);
    if (!(s.flags & BVH_SUM_SUSPENDED)) continue;  // (level 0: a suspended query that k_bvh_coop / k_bvh_shape_coop walked to its end)
    if (level && split.tasks[unit0 + i].entry == 0xFFFFFFFFu) continue;
    bool overflow = (s.flags & BVH_SUM_OVERFLOW) != 0;
    for (uint32_t j = 0; j < s.n_child; ++j) {
      const BvhSum<T> c = *bvh_sum<T>(split, split.n_queries + s.first_child + j);
      overflow = overflow || (c.flags & BVH_SUM_OVERFLOW);
      if (c.cand_val < s.dlb) {  // the child's last bound-lowering leaf also lowers the bound as it stood before the child
        s.cand_val = c.cand_val;
        s.np1 = c.np1;
        s.np2 = c.np2;
        s.nn = c.nn;
      }
      if (c.dlb < s.dlb) {
        s.dlb = c.dlb;
        s.rec_dist = c.rec_dist;
      }
      if (c.ncontacts) {  // the walk ended here
        s.ncontacts = c.ncontacts;
        s.fb1 = c.fb1;
        s.fb2 = c.fb2;
        break;
      }
    }
    if (level) {
      s.flags = overflow ? BVH_SUM_OVERFLOW : 0u;
      s.n_child = 0;
      *bvh_sum<T>(split, slot) = s;
    } else {
      PairOut<T> o;
      o.distance = s.rec_dist;
      o.normal = s.nn;
      o.p1 = s.np1;
      o.p2 = s.np2;
      o.gjk_status = GJK_DID_NOT_RUN;
      o.epa_status = EPA_DID_NOT_RUN;
      o.gjk_iters = o.epa_iters = 0;
      store_bvh_record(io, split.suspended[i], o, s.ncontacts, s.fb1, s.fb2, overflow);
    }
  }
}

// ---------------------------------------------------------------------------------------
// k_bvh_shape: BVHModel<OBBRSS> x convex solid collide(), either operand order.  One query per BS_W-lane
// group (hfcl_bvh_shape.hpp: sequential traversal, the group's lanes share support scans and EPA face
// work); DFS stack and the full-capacity polytope of each group in LDS.
// ---------------------------------------------------------------------------------------

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
  __shared__ uint32_t stacks[G][BS_STACK];  // 32-bit node ids: models of any size
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
// Mesh x solid, one query per lane (k_bvh_collide<T, false, false, SOLID>): the two kernels around the walk.
// k_shape_obb: computeBV<OBBRSS, S>(solid, tf) per query -- the reference's PCA fit of the solid's bound vertices in
// world frame (Jacobi sweeps; hfcl_bvh_shape.hpp: shape_obb) -- and its products with the mesh pose (ObbQuery), by
// pair, so that the walk (and every task a long query is cut into) starts from 15 loaded numbers.
// k_bvh_shape_finish: the leaves that needed EPA, one per BS_W-lane group with the full-capacity polytope in LDS, after
// the walk and its fold-back: a leaf whose unit was not overtaken by an earlier contact (bvh_moot) is the query's contact;
// its depth is compared with the bound the record holds (updateDistanceLowerBoundFromLeaf) and patched in.
// ---------------------------------------------------------------------------------------
// distance(): the solid's OBBRSS itself (rss_lower_bound takes the mesh pose as an argument)
template <typename T>
__global__ void __launch_bounds__(256) k_shape_obbrss(Work wk, LibView<T> lib, IO<T> io) {
  const uint32_t cnt = wk.counts[B_BVHSHAPE];
  RssQuery<T>* const table = reinterpret_cast<RssQuery<T>*>(wk.shape_oq);
  for (uint32_t it = blockIdx.x * blockDim.x + threadIdx.x; it < cnt; it += gridDim.x * blockDim.x) {
    const uint32_t pair = wk.lists[size_t(B_BVHSHAPE) * wk.n + it];
    const uint32_t id1 = wk.shape1[pair], id2 = wk.shape2[pair];
    const bool swapped = lib.kinds[id1] != uint8_t(K_BVH);
    const DShape<T> shape = lib.shapes[swapped ? id1 : id2];
    const Pose<T> tfs = load_pose(swapped ? io.tf1 : io.tf2, pair);
    DNode<T> bv2;
    DRss<T> rss2;
    RssQuery<T> rq;
    if (shape_obbrss(shape, lib.verts, tfs, bv2, &rss2)) {
      rq.axes = bv2.axes;
      rq.Tr = rss2.Tr;
      rq.l0 = rss2.l0;
      rq.l1 = rss2.l1;
      rq.r = rss2.r;
    } else {
      const T nanv = Lim<T>::nan();
      rq.axes.r0 = rq.axes.r1 = rq.axes.r2 = rq.Tr = mk<T>(nanv, nanv, nanv);
      rq.l0 = rq.l1 = rq.r = nanv;
    }
    table[pair] = rq;
  }
}

template <typename T>
__global__ void __launch_bounds__(256) k_shape_obb(Work wk, LibView<T> lib, IO<T> io) {
  const uint32_t cnt = wk.counts[B_BVHSHAPE];
  ObbQuery<T>* const table = reinterpret_cast<ObbQuery<T>*>(wk.shape_oq);
  for (uint32_t it = blockIdx.x * blockDim.x + threadIdx.x; it < cnt; it += gridDim.x * blockDim.x) {
    const uint32_t pair = wk.lists[size_t(B_BVHSHAPE) * wk.n + it];
    const uint32_t id1 = wk.shape1[pair], id2 = wk.shape2[pair];
    const bool swapped = lib.kinds[id1] != uint8_t(K_BVH);
    const DShape<T> shape = lib.shapes[swapped ? id1 : id2];
    const Pose<T> tf1 = load_pose(io.tf1, pair), tf2 = load_pose(io.tf2, pair);
    DNode<T> bv2;
    ObbQuery<T> oq;
    if (shape_obb(shape, lib.verts, swapped ? tf1 : tf2, bv2)) {
      oq = make_obb_query(swapped ? tf2 : tf1, bv2);
    } else {
      const T nanv = Lim<T>::nan();
      oq.M.r0 = oq.M.r1 = oq.M.r2 = oq.V = oq.ext = mk<T>(nanv, nanv, nanv);
    }
    table[pair] = oq;
  }
}

// Two tiers (HFCL_SHAPE_FINISH_TIERS, default on), as k_epa's: CAP = SHAPE_FINISH_FAST_CAP first -- a block a third the size, two waves
// per SIMD where the full-capacity block admits one -- and the polytopes that outgrow it listed in Work::shape_finish_over for a second
// launch at EPA_MAX_ITER that starts them again (tier: 0 the only launch, 1 the fast one, 2 the one over the list).  What a polytope that
// fits computes does not depend on the block's capacity (tests/test_gpu_parity.py: test_epa_hand_over_equals_restart is the same property of
// k_epa's tiers), so the records are those of the one-tier form.
#ifndef HFCL_SHAPE_FINISH_FAST_CAP
#define HFCL_SHAPE_FINISH_FAST_CAP 21
#endif
constexpr int SHAPE_FINISH_FAST_CAP = HFCL_SHAPE_FINISH_FAST_CAP;
template <typename T, int PART = HFCL_BVH_PART, int CAP = EPA_MAX_ITER>
// range: 0 every item; 1 the items in front of CTR_SHAPE_DEFER_MARK (whole walks: final once the first launch of k_bvh_shape_coop is over); 2 the
// items behind it (those of chunks).  Ranges 1 and 2 run on two streams and list their second tier in the two halves of shape_finish_over.
__global__ void __launch_bounds__(64) k_bvh_shape_finish(Work wk, LibView<T> lib, IO<T> io, QParams<T> q, BvhParams bp, BvhSplit split, int distance_mode, int tier, int range) {
  constexpr int G = 64 / BS_W;
  __shared__ EpaScratch<T, CAP> scratch[G];
  const int half = range == 2 ? 1 : 0;
  uint32_t* const over = wk.shape_finish_over + size_t(half) * wk.shape_defer_cap;
  const uint32_t first = tier != 2 && range == 2 ? wk.counts[CTR_SHAPE_DEFER_MARK] : 0u;
  const uint32_t cnt = tier == 2 ? min(wk.counts[CTR_SHAPE_FINISH_OVER + half], wk.shape_defer_cap)
                                 : (range == 1 ? wk.counts[CTR_SHAPE_DEFER_MARK] : min(wk.counts[CTR_SHAPE_DEFER], wk.shape_defer_cap));
  const ShapeDeferItem<T>* const items = reinterpret_cast<const ShapeDeferItem<T>*>(wk.shape_defer);
  const int lane = threadIdx.x & 63, grp = lane / BS_W, lig = lane & (BS_W - 1);
  for (uint32_t k = first + blockIdx.x * G + grp; k < cnt; k += gridDim.x * G) {
    const uint32_t it = tier == 2 ? over[k] : k;
    const ShapeDeferItem<T> item = items[it];
    if (split.tasks && bvh_moot<T>(split, item.parent, item.order)) continue;  // (group-uniform) an earlier contact ended the walk
    const uint32_t pair = item.seed.pair;
    const DShape<T> a = lib.shapes[wk.shape1[pair]], b = lib.shapes[wk.shape2[pair]];
    const bool swapped = a.kind != K_BVH;
    GroupSolid<T> solid;
    solid.s = swapped ? a : b;
    solid.v = lib.verts + 3 * size_t(solid.s.vertex_offset);
    solid.lig = lig;
    if (solid.s.kind == K_CONVEX && solid.s.num_points <= uint32_t(HULL_MAX)) solid.h.load(solid.v, solid.s.num_points, lig);
    const Pose<T> tfs = load_pose(swapped ? io.tf1 : io.tf2, pair);
    V3<T> p1, p2, n, guess;
    bool done = true;
    const T distance = mesh_shape_leaf_finish<T, LaneGroup<BS_W>, GroupSolid<T>, CAP>(item, tfs, solid, swept_radius(solid.s), q, &scratch[grp], p1, p2, n, guess, &done);
    if (CAP != EPA_MAX_ITER && !done) {  // (group-uniform) to the full-capacity launch
      if (lig == 0) over[atomicAdd(&wk.counts[CTR_SHAPE_FINISH_OVER + half], 1u)] = it;
      LaneGroup<BS_W>::sync();
      continue;
    }
    if (lig == 0 && distance_mode) {
      // distance(): the walk ended at this leaf (every bound left on the stack is >= 0); DistanceResult::update
      T mind = item.bound;
      int prim = item.prev_prim;
      if (mind > distance) {
        mind = distance;
        prim = int(item.prim);
        store_witness(io, pair, swapped ? p2 : p1, swapped ? p1 : p2, swapped ? -n : n);
      }
      store_bvh_record_head(io, pair, mind, mind <= T(0) ? 0x80000000u : 0u, prim, -1, false);
      write_guess<T>(io, pair, guess, 0, 0);
    } else if (lig == 0) {
      // the record holds the walk's result without this leaf: recorded distance = bound + margin where a leaf set it (the
      // same subtraction as updateDistanceLowerBoundFromLeaf's), a positive OBB bound otherwise (always above a penetration)
      auto* r = &io.out[pair];
      const T dtc = distance - q.security_margin;
      if (dtc < T(r->distance) - q.security_margin) {
        r->distance = distance;
        store_witness(io, pair, swapped ? p2 : p1, swapped ? p1 : p2, swapped ? -n : n);
      }
      emit_shape_contact(bp, pair, swapped, int(item.prim), distance, p1, p2, n);
      write_guess<T>(io, pair, guess, 0, 0);
    }
    LaneGroup<BS_W>::sync();
  }
}

template <typename T>
static void launch_shape_finish(int grid, hipStream_t st, const Work& wk, const LibView<T>& lv, const IO<T>& io, const QParams<T>& q, const BvhParams& bp, const BvhSplit& split,
                                int distance_mode, int range = 0) {
  if (wk.shape_finish_over && q.epa_max_iterations > SHAPE_FINISH_FAST_CAP) {
    hipLaunchKernelGGL((k_bvh_shape_finish<T, HFCL_BVH_PART, SHAPE_FINISH_FAST_CAP>), dim3(grid), dim3(64), 0, st, wk, lv, io, q, bp, split, distance_mode, 1, range);
    hipLaunchKernelGGL((k_bvh_shape_finish<T, HFCL_BVH_PART, EPA_MAX_ITER>), dim3(std::max(1, grid / 4)), dim3(64), 0, st, wk, lv, io, q, bp, split, distance_mode, 2, range);
  } else {
    hipLaunchKernelGGL((k_bvh_shape_finish<T, HFCL_BVH_PART, EPA_MAX_ITER>), dim3(grid), dim3(64), 0, st, wk, lv, io, q, bp, split, distance_mode, 0, range);
  }
}

// ---------------------------------------------------------------------------------------
// k_bvh_shape_coop: the long walks of a mesh x solid batch, ONE LANE GROUP PER QUERY, COOP_W entries of the stack at a time.
// k_bvh_collide<SOLID> suspends a query after its step budget and leaves its stack (DFS order) and its state (bounds, witness)
// behind; here a group of COOP_W lanes takes the query over and goes on with the SAME sequential walk, COOP_W entries wide:
// the top entries of the (ordered, LDS) stack are popped together, each lane tests its entry's box or runs its triangle's
// leaf, and the results are applied in stack order by scans over the group --
//   * the bound after entry i is the minimum of the bound before the window and the values of entries <= i (exclusive
//     prefix minimum): "this leaf lowered the bound when it was visited" is val_i < that, the witness is the LAST such leaf,
//     the recorded distance belongs to the FIRST entry that reaches the window's minimum (later ties do not lower it);
//   * the first entry with a contact ends the walk, entries behind it are void (their EPA items are marked so);
//   * the children of the overlapping boxes take their parents' places, in order -- and only the entries in FRONT of the
//     window's first overlapping box are visited in a trip (what the others do to the state depends on that box's subtree).
// So the walk visits exactly the reference's sequence, up to COOP_W steps per trip instead of one, and ends with the query's
// record: no task levels, no fold-back.  A step of a lone lane costs ~2 us and a level of tasks its budget times that; a
// trip of this kernel costs one step's latency for COOP_W of them.  Groups take the next suspended query as they finish.
// ---------------------------------------------------------------------------------------
// A walk's state is the same in every lane of its group; when the group is the wave (W == 64) it belongs in SGPRs, not in 64 copies
// that stay live across the leaf's GJK (k_bvh_shape_coop<double>: 624 B/lane of scratch before, profiles/r05_f): wave_uniform() says so
// to the compiler (v_readfirstlane), a no-op for narrower groups.
template <int W>
__device__ __forceinline__ uint32_t wave_uniform(uint32_t x) {
  if constexpr (W == 64) return uint32_t(__builtin_amdgcn_readfirstlane(int(x)));
  return x;
}
template <int W>
__device__ __forceinline__ int wave_uniform(int x) {
  if constexpr (W == 64) return __builtin_amdgcn_readfirstlane(x);
  return x;
}
template <int W>
__device__ __forceinline__ float wave_uniform(float x) {
  if constexpr (W == 64) return __int_as_float(__builtin_amdgcn_readfirstlane(__float_as_int(x)));
  return x;
}
template <int W>
__device__ __forceinline__ double wave_uniform(double x) {
  if constexpr (W == 64) return __hiloint2double(__builtin_amdgcn_readfirstlane(__double2hiint(x)), __builtin_amdgcn_readfirstlane(__double2loint(x)));
  return x;
}
template <int W, typename T>
__device__ __forceinline__ V3<T> wave_uniform(const V3<T>& v) {
  return mk<T>(wave_uniform<W>(v.x), wave_uniform<W>(v.y), wave_uniform<W>(v.z));
}
template <typename T, int W>
__device__ __forceinline__ T group_min_excl_scan(T v, int lig, T identity) {  // exclusive prefix minimum over the lanes of a group
#pragma unroll
  for (int d = 1; d < W; d <<= 1) {
    const T o = __shfl_up(v, d, W);
    if (lig >= d) v = o < v ? o : v;
  }
  const T up = __shfl_up(v, 1, W);
  return lig == 0 ? identity : up;
}
template <typename T, int W>
__device__ __forceinline__ T group_min_all(T v) {
#pragma unroll
  for (int d = W / 2; d >= 1; d >>= 1) {
    const T o = __shfl_xor(v, d, W);
    v = o < v ? o : v;
  }
  return v;
}
// COOP_W lanes per query (a wave walks 64 / COOP_W queries side by side, each with its own stack).  Narrower groups were
// expected to pay -- the triangles that can be visited in one trip rarely come 64 in a row -- and LOSE: the kernel's time
// is that of its longest walks, which need more trips the narrower the window (100k ellipsoid queries: 16.5 / 12 / 9.8 /
// 7.7 ms at 8 / 16 / 32 / 64 lanes, profiles/r03_i).
#ifndef HFCL_COOP_W
#define HFCL_COOP_W 64
#endif
constexpr int COOP_W = HFCL_COOP_W;
#ifndef HFCL_COOP_CAP
#define HFCL_COOP_CAP 448
#endif
constexpr int COOP_CAP = HFCL_COOP_CAP, COOP_SLACK = 64;  // entries of a query's stack: a full stack narrows the window down to plain DFS, which needs the tree's depth more
// What is known about a stack entry travels with it (two bits of the entry word + a value): a box found disjoint keeps its
// bound, a triangle its distance -- its result does not depend on the walk's state (the leaf solver starts from the
// request's guess).  A trip with a triangle in it costs a GJK run whatever the number of lanes that have one, so triangles
// are EVALUATED when HFCL_COOP_LEAF_BATCH lanes of the window hold an unevaluated one or the window has no box left to
// split, wherever they stand -- and VISITED (applied to the walk's state) later, when everything in front of them is done.
// 100k queries per kind, batch 16 / 32 / 64: ellipsoid 7.1 / 6.2 / 5.5 ms, convex32 5.3 / 4.5 / 3.9, cylinder 3.6 / 3.2 / 2.7
// (evaluated where they are visited: 8.1 / 6.2 / 3.8; evaluating whenever the TOP entry is an unevaluated triangle, the first
// form of this idea, gained nothing: nearly every trip then has one) -- profiles/r03_k.
#ifndef HFCL_COOP_LEAF_BATCH
#define HFCL_COOP_LEAF_BATCH 64
#endif
constexpr uint32_t COOP_NODE = 0x3FFFFFFFu, COOP_DISJOINT = 1u << 30, COOP_LEAF = 2u << 30, COOP_LEAF_EPA = 3u << 30;
// HFCL_COOP_PROF (variant builds only, tools/build_variant.sh ... k_bvh -DHFCL_COOP_PROF): how long the walks of k_bvh_shape_coop /
// k_bvh_coop and their waves live, in clock ticks -- [0] longest walk, [1] sum over the walks, [2] walks, [3] longest wave, [4] sum over
// the waves that had a walk, [5] their number, [6] walks cut -- read back by tools/coop_prof.py through hfcl_debug_coop_prof
#ifdef HFCL_COOP_PROF
__device__ unsigned long long coop_prof[32];
#define COOP_WALK_END(first_lane, t_begin)                                                         \
  do {                                                                                             \
    if (first_lane) {                                                                              \
      const unsigned long long dt_ = __builtin_readcyclecounter() - (t_begin);                     \
      atomicMax(&coop_prof[0], dt_);                                                               \
      atomicAdd(&coop_prof[1], dt_);                                                               \
      atomicAdd(&coop_prof[2], 1ull);                                                              \
    }                                                                                              \
  } while (0)
#define COOP_WAVE_END(t_begin, n_walks)                                                            \
  do {                                                                                             \
    if (threadIdx.x == 0 && (n_walks) > 0) {                                                       \
      const unsigned long long dt_ = __builtin_readcyclecounter() - (t_begin);                     \
      atomicMax(&coop_prof[3], dt_);                                                               \
      atomicAdd(&coop_prof[4], dt_);                                                               \
      atomicAdd(&coop_prof[5], 1ull);                                                              \
    }                                                                                              \
  } while (0)
#define COOP_CUT_COUNT(first_lane)                      \
  do {                                                  \
    if (first_lane) atomicAdd(&coop_prof[6], 1ull);     \
  } while (0)
extern "C" int hfcl_debug_coop_prof(unsigned long long* out16, int reset) {
  if (hipDeviceSynchronize() != hipSuccess) return -1;
  if (hipMemcpyFromSymbol(out16, HIP_SYMBOL(coop_prof), sizeof(coop_prof)) != hipSuccess) return -1;
  if (reset) {
    unsigned long long z[32] = {0};
    if (hipMemcpyToSymbol(HIP_SYMBOL(coop_prof), z, sizeof(z)) != hipSuccess) return -1;
  }
  return 0;
}
#define COOP_PROF_DECL unsigned long long pf_[18] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0}, pt_ = 0
#define COOP_PROF_T0 (pt_ = __builtin_readcyclecounter())
#define COOP_PROF_ADD(k, x) (pf_[k] += (unsigned long long)(x))
#define COOP_PROF_DT(k) (pf_[k] += __builtin_readcyclecounter() - pt_)
#define COOP_PROF_FLUSH                                                 \
  do {                                                                  \
    if (threadIdx.x == 0)                                               \
      for (int k_ = 0; k_ < 18; ++k_) atomicAdd(&coop_prof[7 + k_], pf_[k_]); \
  } while (0)
#else
#define COOP_WALK_END(first_lane, t_begin)
#define COOP_WAVE_END(t_begin, n_walks)
#define COOP_CUT_COUNT(first_lane)
#define COOP_PROF_DECL
#define COOP_PROF_T0
#define COOP_PROF_ADD(k, x)
#define COOP_PROF_DT(k)
#define COOP_PROF_FLUSH
#endif
// Cutting a long walk (BvhSplit::cut_ticks).  A batch of 100 000 queries used to end with one wave on its longest walk (mesh x solid:
// 1.7 of the kernel's 2.9 ms, the waves busy 54 % of the time; mesh x mesh: 79 %; profiles/r04_j).  A walk that has had its time is
// therefore cut: its stack (DFS order, top first) goes to BvhSplit::cut_words / cut_vals as it is -- tags and values included -- in
// chunks of COOP_CHUNK entries, each chunk a BvhTask of the next launch of the same kernel, whose unit starts from those entries with an
// empty state and ends with a summary (BvhSum) instead of a record; the cut walk's own state is the summary its chunks hang under, and
// k_bvh_combine folds them back in DFS order exactly as it folds k_bvh_collide's task levels (bound: a minimum; witness: the last
// triangle of a chunk that lowered the chunk's own bound, if it also lies below the bound as it stood before the chunk; a contact ends
// the fold).  A chunk can be cut again (two more launches); where it is cut never changes a record, only when its parts are walked.
// What a lane group draws from the launch's ticket: a suspended query (level 0) or a chunk of a cut walk.
struct CoopUnit {
  bool take;       // there is a unit to walk (not: the ticket ran out, a slot of a table that was full, a chunk behind a contact)
  bool exhausted;  // the ticket has run past the launch's units
  uint32_t index;  // of the unit among the launch's units
  uint32_t pair, parent, order;  // the query; (chunks) the summary slot of the walk it was cut from and its place among that walk's chunks
  uint32_t first, count;         // chunks: its entries are cut_words / cut_vals [first, first + count)
};
template <typename T, int W>
__device__ __forceinline__ CoopUnit coop_draw(const BvhSplit& split, uint32_t* ticket, uint32_t level, uint32_t unit0, uint32_t n_units, int lig, int grp) {
  CoopUnit u = {false, false, 0u, 0u, 0xFFFFFFFFu, 0u, 0u, 0u};
  uint32_t qi = 0;
  if (lig == 0) qi = atomicAdd(ticket, 1u);
  qi = uint32_t(__shfl(int(qi), grp * W));
  u.index = qi;
  u.exhausted = qi >= n_units;
  if (u.exhausted) return u;
  if (!level) {  // (unit0: where this launch's part of the suspended list starts, BvhSplit::coop_range)
    u.index = split.order ? split.order[unit0 + qi] : unit0 + qi;
    u.pair = split.suspended[u.index];
    u.take = true;
    return u;
  }
  const BvhTask task = split.tasks[unit0 + qi];
  if (task.entry == 0xFFFFFFFFu) return u;  // (a slot of a table that was full)
  if (bvh_moot<T>(split, task.parent, task.order & 0xFFFFu)) {  // it stands behind a contact: nobody reads its summary
    if (lig == 0) bvh_sum<T>(split, split.n_queries + unit0 + qi)->flags = 0u;
    return u;
  }
  u.pair = task.pair;
  u.parent = task.parent;
  u.order = task.order & 0xFFFFu;
  u.first = task.entry;
  u.count = task.order >> 16;
  u.take = true;
  return u;
}
// a contact in a chunk ends the walk of every unit above it at that chunk's position (k_bvh_collide: report_contact)
template <typename T>
__device__ __forceinline__ void coop_report_contact(const BvhSplit& split, uint32_t p, uint32_t o) {
  for (int hop = 0; hop < BVH_MAX_LEVELS + 1 && p != 0xFFFFFFFFu; ++hop) {
    BvhSum<T>* ps = bvh_sum<T>(split, p);
    atomicMin(&ps->contact_order, o);
    o = ps->order;
    p = ps->parent;
  }
}
template <typename T>
__device__ __forceinline__ void coop_write_sum(const BvhSplit& split, uint32_t slot, T dlb, T rec_dist, T cand_val, const V3<T>& np1, const V3<T>& np2,
                                               const V3<T>& nn, int fb1, int fb2, uint32_t ncontacts, uint32_t first_child, uint32_t n_child,
                                               uint32_t flags, uint32_t parent, uint32_t order) {
  BvhSum<T>* sm = bvh_sum<T>(split, slot);
  sm->dlb = dlb; sm->rec_dist = rec_dist; sm->cand_val = cand_val;
  sm->np1 = np1; sm->np2 = np2; sm->nn = nn;
  sm->fb1 = fb1; sm->fb2 = fb2;
  sm->ncontacts = ncontacts; sm->first_child = first_child; sm->n_child = n_child; sm->flags = flags;
  sm->contact_order = 0xFFFFFFFFu; sm->parent = parent; sm->order = order; sm->pad_ = 0;
}
// The cut itself, by the lanes of the unit's group: the stack (sp entries, top first; `value`: what is known about them, or nullptr) to
// cut_words / cut_vals in chunks of COOP_CHUNK, a BvhTask per chunk, the unit's state as the summary they hang under.  false: the tables
// have no room (the slots taken are no-ops): the walk goes on.
template <typename T, int W>
__device__ __forceinline__ bool coop_cut(const BvhSplit& split, uint32_t my_slot, uint32_t pair, uint32_t my_parent, uint32_t my_order, const uint32_t* stack,
                                         const T* value, int sp, int lig, T dlb, T rec_dist, T cand_val, const V3<T>& np1, const V3<T>& np2, const V3<T>& nn,
                                         bool overflow) {
  const uint32_t n_ent = uint32_t(sp), n_chunks = (n_ent + COOP_CHUNK - 1u) / COOP_CHUNK;
  uint32_t first_task = 0, first_word = 0;
  if (lig == 0) {
    first_task = atomicAdd(&split.ctr[BVH_CTR_TASKS], n_chunks);
    first_word = atomicAdd(&split.ctr[BVH_CTR_CUT], n_ent);
  }
  first_task = uint32_t(__shfl(int(first_task), 0, W));
  first_word = uint32_t(__shfl(int(first_word), 0, W));
  __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");  // (the stack was written back by the lanes of this trip)
  __builtin_amdgcn_wave_barrier();
  if (first_task + n_chunks > min(split.cap, split.cut_task_cap) || first_word + n_ent > split.cut_cap) {
    for (uint32_t c = first_task + uint32_t(lig); c < min(first_task + n_chunks, split.cap); c += uint32_t(W)) split.tasks[c] = BvhTask{0u, 0u, 0xFFFFFFFFu, 0u};
    return false;
  }
  T* const cut_vals = reinterpret_cast<T*>(split.cut_vals);
  for (uint32_t k = uint32_t(lig); k < n_ent; k += uint32_t(W)) {
    split.cut_words[first_word + k] = stack[sp - 1 - int(k)];
    if (value) cut_vals[first_word + k] = value[sp - 1 - int(k)];
  }
  for (uint32_t c = uint32_t(lig); c < n_chunks; c += uint32_t(W))
    split.tasks[first_task + c] = BvhTask{pair, my_slot, first_word + c * COOP_CHUNK, c | (min(uint32_t(COOP_CHUNK), n_ent - c * COOP_CHUNK) << 16)};
  if (lig == 0)
    coop_write_sum<T>(split, my_slot, dlb, rec_dist, cand_val, np1, np2, nn, -1, -1, 0u, first_task, n_chunks,
                      BVH_SUM_SUSPENDED | (overflow ? BVH_SUM_OVERFLOW : 0u), my_parent, my_order);
  return true;
}
#ifndef HFCL_WPE_SHAPE_COOP
#define HFCL_WPE_SHAPE_COOP 2
#endif
// What a walk of k_bvh_shape_coop and the leaf it calls share, in LDS (round 6).  The leaf's GJK takes 248 registers, so whatever the
// walk holds across the call goes to scratch memory and back: as first built -- the walk's state in registers, the leaf's arguments a
// by-value block, its result a block behind a pointer, the request's parameters behind another -- that was 608 B per lane, and 100 000
// queries moved 6.4 GB through it (a read-only tree walk whose records are 9.6 MB; profiles/r05 traffic_cfg4s.json): more than the
// XCD's L2 holds for its resident waves, so the kernel ran at half the chip's memory bandwidth.  Here the call passes two words per lane
// (the triangle, a flag) and returns three; everything that is the same for the lanes of a walk -- where the model's arrays are, the
// query, the solid, the request, the solid's OBB products, the witness of the walk's bound -- is in this block, and a leaf's witness
// goes to a lane-minor slab beside it.
template <typename T>
struct ShapeCoopCtx {
  const T* mesh_verts;      // of the walk's model
  const uint32_t* tris;
  const DShape<T>* shapes;
  const T* lib_verts;
  decltype(IO<T>::tf1) pose_m;  // pose arrays of the mesh / the solid
  decltype(IO<T>::tf1) pose_s;
  ShapeDeferItem<T>* defer;
  uint32_t* defer_count;
  T* wit;                   // [9][64]: p1, p2, n of the leaf each lane ran last
  uint32_t defer_cap;
  uint32_t pair, solid_id, parent, order;
  V3<T> guess0;
  QParams<T> q;
  ObbQuery<T> oq;
  V3<T> np1, np2, nn;       // witness of the walk's bound (record orientation)
};
template <typename T>
struct CoopLeafRet {
  T distance;
  uint32_t to_epa, slot;
};
// (not_tail_called: LLVM marks a call that hands no pointer into the caller's frame as a possible tail call, and a function with such a call
// site keeps the convention's callee-saved registers -- 50 of them here, saved and restored around every leaf, 200 B of scratch per lane;
// without the mark the register allocation is interprocedural and the callee saves nothing the walk does not hold)
template <typename T, class PS>
__device__ __noinline__ __attribute__((not_tail_called)) CoopLeafRet<T> coop_solid_leaf(const ShapeCoopCtx<T>* c, const PS ps, uint32_t prim, uint32_t push) {
  const QParams<T> q = c->q;
  const T* const mv = c->mesh_verts;
  const uint32_t* const tri = c->tris + 3 * size_t(prim);
  auto vtx = [&](uint32_t i) { return mk<T>(mv[3 * size_t(i)], mv[3 * size_t(i) + 1], mv[3 * size_t(i) + 2]); };
  const V3<T> ta = vtx(tri[0]), tb = vtx(tri[1]), tc = vtx(tri[2]);
  LaneSolid<T> solid;
  solid.s = c->shapes[c->solid_id];
  solid.v = c->lib_verts + 3 * size_t(solid.s.vertex_offset);
  const uint32_t pair = c->pair;
  auto tfm_of = [&]() { return load_pose(c->pose_m, pair); };
  auto tfs_of = [&]() { return load_pose(c->pose_s, pair); };
  const MDiff<T> sMt = make_mdiff(tfs_of(), tfm_of());
  V3<T> guess = c->guess0;
  ShapeDeferItem<T> item;
  CoopLeafRet<T> r;
  r.distance = Lim<T>::max();
  r.slot = 0u;
  V3<T> p1 = mk<T>(T(0), T(0), T(0)), p2 = p1, n = p1;
  const bool to_epa = mesh_shape_leaf_lane(ta, tb, tc, sMt, tfm_of, tfs_of, solid.s, solid, swept_radius(solid.s), q, guess, ps, r.distance, p1, p2, n, item);
  r.to_epa = to_epa ? 1u : 0u;
  if (to_epa) {
    r.distance = Lim<T>::max();
    if (push) {
      item.seed.pair = pair;
      item.prim = prim;
      item.parent = c->parent;
      item.order = c->order;
      item.bound = T(0);
      item.prev_prim = -1;
      const uint32_t slot = atomicAdd(c->defer_count, 1u);
      if (slot < c->defer_cap) c->defer[slot] = item;  // (the host sizes the queue for one item per unit of the batch; never past its end)
      r.slot = slot;
    }
  } else {
    T* const w = c->wit + (threadIdx.x & 63);
    w[0 * 64] = p1.x; w[1 * 64] = p1.y; w[2 * 64] = p1.z;
    w[3 * 64] = p2.x; w[4 * 64] = p2.y; w[5 * 64] = p2.z;
    w[6 * 64] = n.x;  w[7 * 64] = n.y;  w[8 * 64] = n.z;
  }
  return r;
}
template <typename T>
__global__ void __launch_bounds__(64) __attribute__((amdgpu_waves_per_eu(HFCL_WPE_SHAPE_COOP, 8)))
k_bvh_shape_coop(Work wk, LibView<T> lib, BvhView<T> bv, IO<T> io, QParams<T> q, BvhParams bp, T break_distance2, BvhSplit split) {
  constexpr int W = COOP_W, G = 64 / W;
  __shared__ uint32_t stacks[G][COOP_CAP + COOP_SLACK];
  __shared__ T values[G][COOP_CAP + COOP_SLACK];
  __shared__ T w0_slab[W0Lds<T, 64>::WORDS];
  __shared__ ShapeCoopCtx<T> ctxs[G];
  __shared__ T wit_slab[9 * 64];
  const W0Lds<T, 64> leaf_ps{w0_slab + threadIdx.x};
  const int lane = threadIdx.x, grp = lane / W, lig = lane & (W - 1);
  uint32_t* const stack = stacks[grp];
  T* const value = values[grp];
  ShapeCoopCtx<T>& ctx = ctxs[grp];
  const uint64_t gbits = W == 64 ? ~uint64_t(0) : ((uint64_t(1) << (W & 63)) - 1);
  auto gballot = [&](bool x) -> uint64_t { return (__ballot(x) >> (grp * W)) & gbits; };  // the group's lanes, bit 0 = its first lane
  if (lig == 0) {  // what does not change from walk to walk
    ctx.shapes = lib.shapes;
    ctx.lib_verts = lib.verts;
    ctx.defer = reinterpret_cast<ShapeDeferItem<T>*>(wk.shape_defer);
    ctx.defer_count = &wk.counts[CTR_SHAPE_DEFER];
    ctx.defer_cap = wk.shape_defer_cap;
    ctx.wit = wit_slab;
    ctx.q = q;
  }
  // the units of this launch: the suspended queries (level 0), or the chunks the launch before cut its long walks into
  const uint32_t level = split.level;
  const uint32_t unit0 = level ? min(split.ctr[BVH_CTR_LEVEL0 + level - 1], split.cap) : 0u;
  const uint32_t n_units = level ? min(split.ctr[BVH_CTR_LEVEL0 + level], split.cap) - unit0 : min(split.ctr[BVH_CTR_SUSPENDED], split.n_queries);
  if (n_units == 0u) return;  // (nothing suspended / no chunk cut: no wave draws a ticket)
  const unsigned long long cut_ticks = split.can_suspend ? split.cut_ticks : 0u;
  uint32_t* const ticket = &wk.counts[CTR_SHAPE_TICKET];  // (k_bvh_collide<SOLID> is through with it; k_bvh_level_mark has reset it)
  T* const cut_vals = reinterpret_cast<T*>(split.cut_vals);
  const T big = Lim<T>::max();
  // per-group state of the query being walked (uniform over the group's lanes)
  bool have = false, exhausted = false;
  uint32_t pair = 0, solid_id = 0, ncontacts = 0;
  uint32_t unit = 0, my_parent = 0xFFFFFFFFu, my_order = 0;  // where the unit hangs (a chunk: the cut walk's summary slot, its place among the chunks)
  unsigned long long t_begin = 0, t_wave = __builtin_readcyclecounter();
  unsigned n_walks = 0, trip_no = 0;
  (void)t_wave; (void)n_walks;
  bool swapped = false, overflow = false;
  COOP_PROF_DECL;  // [0] trips [1] box-test ticks [2] boxes tested [3] leaf batches [4] their ticks [5] their lanes [6] ticks from the scans to the end of a trip that ends in a contact [7] ticks drawing and loading units
  uint32_t node_off = 0;
  int sp = 0, fb = -1;
  T dlb = big, rec_dist = big, cand_val = big;
  // the witness of lane `l`'s last leaf, from the slab (record orientation)
  auto slab_witness = [&](int l, V3<T>& a1, V3<T>& a2, V3<T>& an) {
    const T* const w = wit_slab + l;
    const int o1 = swapped ? 3 * 64 : 0, o2 = swapped ? 0 : 3 * 64;  // (rows, not a select between two structs: that one goes through an indexed stack slot)
    const T sg = swapped ? T(-1) : T(1);
    a1 = mk<T>(w[o1], w[o1 + 64], w[o1 + 128]);
    a2 = mk<T>(w[o2], w[o2 + 64], w[o2 + 128]);
    an = mk<T>(sg * w[6 * 64], sg * w[7 * 64], sg * w[8 * 64]);
  };
  for (;;) {
    if (!have && !exhausted) {
      COOP_PROF_T0;
      const CoopUnit u = coop_draw<T, W>(split, ticket, level, unit0, n_units, lig, grp);
      exhausted = u.exhausted;
      if (u.take) {
      BvhSum<T> s;
      if (level) {  // a chunk starts from an empty state
        s.dlb = s.rec_dist = s.cand_val = big;
        s.np1 = s.np2 = s.nn = mk<T>(Lim<T>::nan(), Lim<T>::nan(), Lim<T>::nan());
        s.flags = 0u;
        s.first_child = u.first;
        s.n_child = u.count;
      } else {
        s = *bvh_sum<T>(split, u.index);
      }
      pair = wave_uniform<W>(u.pair);
      unit = wave_uniform<W>(u.index);
      my_parent = wave_uniform<W>(u.parent);
      my_order = wave_uniform<W>(u.order);
      const uint32_t id1 = wave_uniform<W>(wk.shape1[pair]), id2 = wave_uniform<W>(wk.shape2[pair]);
      swapped = wave_uniform<W>(int(lib.kinds[id1] != uint8_t(K_BVH))) != 0;
      solid_id = swapped ? id1 : id2;
      {
        const DMesh mm = bv.meshes[wave_uniform<W>(lib.shapes[swapped ? id2 : id1].bvh_index)];
        node_off = wave_uniform<W>(mm.node_off);
        if (lig == 0) {
          ctx.mesh_verts = bv.verts + 3 * size_t(mm.vert_off);
          ctx.tris = bv.tris + 3 * size_t(mm.tri_off);
          ctx.pose_m = swapped ? io.tf2 : io.tf1;
          ctx.pose_s = swapped ? io.tf1 : io.tf2;
          ctx.pair = pair;
          ctx.solid_id = solid_id;
          ctx.parent = u.parent;
          ctx.order = u.order;
          ctx.oq = reinterpret_cast<const ObbQuery<T>*>(wk.shape_oq)[pair];
          ctx.np1 = s.np1;
          ctx.np2 = s.np2;
          ctx.nn = s.nn;
          ctx.guess0 = initial_guess<T>(io, q, pair);  // (walks whose leaves hand a cached guess on are not split)
        }
      }
      if (level) {  // the chunk's entries, with what is known about them
        for (uint32_t j = uint32_t(lig); j < s.n_child; j += uint32_t(W)) {
          stack[s.n_child - 1u - j] = split.cut_words[s.first_child + j];
          value[s.n_child - 1u - j] = cut_vals[s.first_child + j];
        }
      } else {
        for (uint32_t j = uint32_t(lig); j < s.n_child; j += uint32_t(W)) stack[s.n_child - 1u - j] = split.tasks[s.first_child + j].entry;  // child 0 on top
      }
      sp = wave_uniform<W>(int(s.n_child));
      dlb = wave_uniform<W>(s.dlb);
      rec_dist = wave_uniform<W>(s.rec_dist);
      cand_val = wave_uniform<W>(s.cand_val);
      t_begin = __builtin_readcyclecounter();
      ++n_walks;
      fb = -1;
      ncontacts = 0;
      overflow = wave_uniform<W>(int((s.flags & BVH_SUM_OVERFLOW) != 0)) != 0 || sp > COOP_CAP;
      if (overflow) sp = 0;
      have = true;
      }
      COOP_PROF_DT(7);
    }
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    __builtin_amdgcn_wave_barrier();
    if (!__any(have)) {
      if (__all(exhausted)) break;
      continue;  // (the units drawn were chunks nobody needs)
    }
    if (have && level && (++trip_no & 15u) == 0u && bvh_moot<T>(split, my_parent, my_order)) {
      // (every 16th trip: the look costs a dependent load per level, every trip 3.3 -> 3.75 ms on cfg4s) a chunk in front of this one
      // has found a contact meanwhile: the sequential walk ended there, nobody reads this summary
      if (lig == 0) bvh_sum<T>(split, split.n_queries + unit0 + unit)->flags = 0u;
      COOP_WALK_END(lig == 0, t_begin);
      have = false;
      continue;
    }
    bool done = have && sp == 0;
    if (have && sp > 0) {
      const int w = min(W, min(sp, max(COOP_CAP - sp, 1)));
      const bool act = lig < w;
      const uint32_t ew = act ? stack[sp - 1 - lig] : 0u;
      T val = act ? value[sp - 1 - lig] : big;  // (meaningful for tagged entries only)
      sp -= w;
      const uint32_t e = ew & COOP_NODE;
      uint32_t tag = ew & ~COOP_NODE;
      // (1) entries nothing is known about yet: a box is tested -- an overlapping one only refines the stack (its children
      // take its place, at any position), a disjoint one keeps its bound; a triangle waits for the leaf batch
      const DNode<T>* const np = bv.nodes + node_off + e;
      int32_t fc = 0;
      bool overlap = false, pending = false;
      COOP_PROF_ADD(0, 1);
      COOP_PROF_ADD(2, __popcll(__ballot(act && tag == 0u)));
      COOP_PROF_ADD(12, __popcll(__ballot(act)));                          // window entries
      COOP_PROF_ADD(13, __popcll(__ballot(act && tag == COOP_DISJOINT)));  // ... boxes found disjoint, waiting for their visit
      COOP_PROF_ADD(14, __popcll(__ballot(act && (tag == COOP_LEAF || tag == COOP_LEAF_EPA))));  // ... evaluated triangles, waiting
      COOP_PROF_ADD(15, sp + w);                                           // stack depth
      COOP_PROF_T0;
      if (act && tag == 0u) {
        const DNode<T> n1 = *np;  // (the whole record at once: a leaf's box is read for nothing, an inner node's not a latency later)
        fc = n1.first_child;
        if (fc >= 0) {
          T sq;
          const ObbQuery<T> oq = ctx.oq;
          if (obb_disjoint_q(oq, n1, q.security_margin, break_distance2, sq)) {
            val = hsqrt(sq);
            tag = COOP_DISJOINT;
          } else {
            overlap = true;
          }
        } else {
          pending = true;
        }
      }
      COOP_PROF_DT(1);
      // (2) the triangles, when enough of them wait or the window has nothing left to split
      bool fresh = false;  // this lane ran its leaf in this trip (the slab holds its witness)
      {
        const uint64_t pmask = gballot(pending);
        if (pmask && (__popcll(pmask) >= HFCL_COOP_LEAF_BATCH || !gballot(overlap))) {
          COOP_PROF_ADD(3, 1);
          COOP_PROF_ADD(5, __popcll(pmask));
          COOP_PROF_T0;
          if (pending) {
            const CoopLeafRet<T> r = coop_solid_leaf<T>(&ctx, leaf_ps, uint32_t(-(fc + 1)), 0u);
            val = r.distance;
            tag = r.to_epa ? COOP_LEAF_EPA : COOP_LEAF;
            pending = false;
            fresh = true;
          }
          COOP_PROF_DT(4);
        }
      }
      // (3) what an entry does to the walk's state depends on everything before it in DFS order -- which includes the subtrees of
      // overlapping boxes and the triangles still waiting ahead of it: the entries in front of the first of those are visited now
      COOP_PROF_T0;
      const uint64_t block = gballot(overlap || pending);
      const int f = block ? __ffsll((unsigned long long)block) - 1 : W;
      COOP_PROF_ADD(16, __popcll(gballot(pending)));  // unevaluated triangles left waiting for a batch
      COOP_PROF_ADD(17, min(f, w));                   // entries visited
      bool visit = act && lig < f;
      const bool is_leaf = tag == COOP_LEAF || tag == COOP_LEAF_EPA;
      T bnd = big, recv = big;  // what this entry sets the bound to, and the recorded distance that goes with it
      bool contact = false;
      if (visit) {
        if (tag == COOP_DISJOINT) {  // updateDistanceLowerBoundFromBV
          bnd = val;
          recv = val + q.security_margin;
        } else if (tag == COOP_LEAF) {  // updateDistanceLowerBoundFromLeaf
          bnd = val - q.security_margin;
          recv = val;
          contact = bnd <= q.collision_distance_threshold;
        } else {
          contact = true;  // (needs EPA: a contact on the host's word)
        }
      }
      const uint64_t cmask = gballot(contact);
      const int c = cmask ? __ffsll((unsigned long long)cmask) - 1 : W;  // the first contact in stack order ends the walk
      visit = visit && lig <= c;
      if (!visit) bnd = big;
      const T before = hmin(dlb, group_min_excl_scan<T, W>(bnd, lig, big));  // the bound as entry `lig` found it
      const bool lowered = visit && bnd < before;
      const T wmin = group_min_all<T, W>(bnd);
      if (gballot(lowered)) {
        // the bound ends at the minimum of the visited entries, set by the first of them that reaches it
        const int src = __ffsll((unsigned long long)gballot(visit && bnd == wmin)) - 1;
        dlb = wave_uniform<W>(wmin);
        rec_dist = wave_uniform<W>(__shfl(recv, src, W));
      }
      // the witness: the last triangle that lowered the bound on its visit (its points, if it was evaluated in an earlier trip,
      // by running its leaf once more); the contact's points likewise
      const uint64_t wmask = gballot(lowered && tag == COOP_LEAF);
      const int L = wmask ? 63 - __clzll((unsigned long long)wmask) : -1;
      if (L >= 0) cand_val = wave_uniform<W>(__shfl(bnd, L, W));
      const bool is_contact_lane = c < W && lig == c;
      COOP_PROF_DT(8);
      COOP_PROF_T0;
      if (act && is_leaf && !fresh && ((lig == L) || (is_contact_lane && tag == COOP_LEAF && bp.contacts))) {
        fc = np->first_child;
        coop_solid_leaf<T>(&ctx, leaf_ps, uint32_t(-(fc + 1)), 0u);  // (for its witness: the slab)
      }
      __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
      __builtin_amdgcn_wave_barrier();
      if (L >= 0 && lig == 0) slab_witness(grp * W + L, ctx.np1, ctx.np2, ctx.nn);
      if (c < W) {  // canStop()
        if (act && is_leaf && fc == 0) fc = np->first_child;
        const int prim_c = wave_uniform<W>(__shfl(int(-(fc + 1)), c, W));
        fb = prim_c;
        ncontacts = 1;
        bool lost = false;
        if (is_contact_lane) {
          if (tag == COOP_LEAF_EPA) {  // its leaf once more, this time with the EPA item (k_bvh_shape_finish patches the record)
            const CoopLeafRet<T> r = coop_solid_leaf<T>(&ctx, leaf_ps, uint32_t(prim_c), 1u);
            lost = r.slot >= wk.shape_defer_cap;
          } else if (bp.contacts) {
            const T* const w = wit_slab + lane;  // (its leaf ran in this trip, above at the latest)
            emit_shape_contact(bp, pair, swapped, prim_c, val, mk<T>(w[0 * 64], w[1 * 64], w[2 * 64]), mk<T>(w[3 * 64], w[4 * 64], w[5 * 64]),
                               mk<T>(w[6 * 64], w[7 * 64], w[8 * 64]));
          }
        }
        // (the host sizes the queue of EPA items for the units it expects -- one and a half per query once walks are cut into chunks --;
        // a contact whose item found it full says so in its record instead of keeping a depth nobody computed)
        if (gballot(lost)) overflow = true;
        sp = 0;
        COOP_PROF_DT(6);
      } else {
        // (4) the stack again, in order (entry 0's successors on top): visited entries are gone, an overlapping box is its two
        // children (left above right), everything else stays with what is known about it
        COOP_PROF_DT(9);  // (the witness part of a trip without a contact)
        COOP_PROF_T0;
        const int cnt = !act || lig < f ? 0 : (overlap ? 2 : 1);
        const uint64_t m2 = gballot(cnt == 2), m1b = gballot(cnt == 1);
        const uint64_t deeper = ~((uint64_t(2) << lig) - 1);  // window entries behind this one (pushed first)
        const int pos = sp + 2 * __popcll(m2 & deeper) + __popcll(m1b & deeper);
        if (cnt == 2) {
          stack[pos] = uint32_t(fc) + 1u;
          stack[pos + 1] = uint32_t(fc);
        } else if (cnt == 1) {
          stack[pos] = e | tag;
          value[pos] = val;
        }
        sp += 2 * __popcll(m2) + __popcll(m1b);
        if (sp > COOP_CAP + COOP_SLACK - 2) {  // a tree deeper than the slack on top of a full stack: flagged, never written past the block
          overflow = true;
          sp = 0;
        }
      }
      done = sp == 0;
      if (done) COOP_PROF_T0; else COOP_PROF_DT(10);
      // (cutting sooner once the launch's ticket has run out -- nothing left to draw, idle slots waiting -- was measured: 3.10 -> 5.9 / 7.2 / 8.0 ms
      // with 150k / 80k / 40k ticks: the three launches are all the levels there are, and what the early cuts leave for the last one is long)
      if (cut_ticks && sp > COOP_CHUNK && __builtin_readcyclecounter() - t_begin > cut_ticks) {
        // ---- the walk has had its time: what is left of it becomes chunks for the next launch (coop_cut)
        if (coop_cut<T, W>(split, level ? split.n_queries + unit0 + unit : unit, pair, my_parent, my_order, stack, value, sp, lig, dlb, rec_dist, cand_val,
                           ctx.np1, ctx.np2, ctx.nn, overflow)) {
          COOP_CUT_COUNT(lig == 0);
          COOP_WALK_END(lig == 0, t_begin);
          have = false;
          sp = 0;
        } else {
          t_begin = __builtin_readcyclecounter();  // (no room: the walk goes on and asks again after another budget)
        }
      }
    }
    if (done && level) {
      // ---- a chunk's summary (k_bvh_combine folds it into the walk it was cut from)
      if (lig == 0) {
        coop_write_sum<T>(split, split.n_queries + unit0 + unit, dlb, rec_dist, cand_val, ctx.np1, ctx.np2, ctx.nn, swapped ? -1 : fb, swapped ? fb : -1, ncontacts, 0u, 0u,
                          overflow ? BVH_SUM_OVERFLOW : 0u, my_parent, my_order);
        if (ncontacts) coop_report_contact<T>(split, my_parent, my_order);
      }
      COOP_WALK_END(lig == 0, t_begin);
      have = false;
      done = false;
    }
    if (done) {
      if (lig == 0) {
        bvh_sum<T>(split, unit)->flags = 0u;  // (not cut: k_bvh_combine has nothing to fold for this query)
        PairOut<T> o;
        o.distance = rec_dist;
        o.normal = ctx.nn;
        o.p1 = ctx.np1;
        o.p2 = ctx.np2;
        o.gjk_status = GJK_DID_NOT_RUN;
        o.epa_status = EPA_DID_NOT_RUN;
        o.gjk_iters = o.epa_iters = 0;
        store_bvh_record(io, pair, o, ncontacts, swapped ? -1 : fb, swapped ? fb : -1, overflow);
      }
      COOP_WALK_END(lig == 0, t_begin);
      have = false;
      COOP_PROF_DT(11);
    }
  }
  COOP_PROF_FLUSH;
  COOP_WAVE_END(t_wave, n_walks);
}

// ---------------------------------------------------------------------------------------
// k_bvh_coop: the same continuation for mesh x mesh walks (BvhSplit::coop on the plain form of k_bvh_collide): a query that
// used up its step budget is taken over by a lane group that walks COOP_W entries of its (ordered) stack per trip.  An entry
// is a pair of nodes; a box pair that overlaps is replaced by its two successors (firstOverSecond decides which node is
// split), a disjoint one by its bound, a pair of leaves by its triangles' distance -- applied in stack order exactly as in
// k_bvh_shape_coop above (same scans), so the record is the sequential walk's.
// ---------------------------------------------------------------------------------------
template <typename T>
__global__ void __launch_bounds__(64) __attribute__((amdgpu_waves_per_eu(2, 8)))
k_bvh_coop(Work wk, LibView<T> lib, BvhView<T> bv, IO<T> io, QParams<T> q, BvhParams bp, T break_distance2, BvhSplit split) {
  constexpr int W = COOP_W, G = 64 / W;
  typedef BvhEntry<false> EN;
  __shared__ uint32_t stacks[G][COOP_CAP + COOP_SLACK];
  __shared__ T w0_slab[W0Lds<T, 64>::WORDS];
  const W0Lds<T, 64> leaf_ps{w0_slab + threadIdx.x};
  const int lane = threadIdx.x, grp = lane / W, lig = lane & (W - 1);
  uint32_t* const stack = stacks[grp];
  const uint64_t gbits = W == 64 ? ~uint64_t(0) : ((uint64_t(1) << (W & 63)) - 1);
  auto gballot = [&](bool x) -> uint64_t { return (__ballot(x) >> (grp * W)) & gbits; };
  // the units of this launch: the suspended queries (level 0), or the chunks the launch before cut its long walks into (see above)
  const uint32_t level = split.level;
  uint32_t unit0 = level ? min(split.ctr[BVH_CTR_LEVEL0 + level - 1], split.cap) : 0u;
  uint32_t n_units = level ? min(split.ctr[BVH_CTR_LEVEL0 + level], split.cap) - unit0 : 0u;
  uint32_t* ticket = &wk.counts[B_COUNT + 2];  // (k_bvh_collide is through with it; k_bvh_level_mark has reset it)
  if (!level) {
    const uint32_t cr = split.coop_range;
    if (cr && cr < 0x100u) {  // what round cr - 1 of the walk handed over; the later rounds append to the list meanwhile
      const uint32_t r = cr - 1u;
      unit0 = r ? min(split.walk.ctr[8u * (r - 1u) + WALK_CTR_SNAP], split.n_queries) : 0u;
      n_units = min(split.walk.ctr[8u * r + WALK_CTR_SNAP], split.n_queries) - unit0;
      ticket = &split.walk.ctr[8u * r + WALK_CTR_TICKET_EARLY];
    } else {
      const uint32_t tot = min(split.ctr[BVH_CTR_SUSPENDED], split.n_queries);
      if (cr) unit0 = min(split.walk.ctr[8u * (cr - 0x100u) + WALK_CTR_SNAP], tot);
      n_units = tot - unit0;
    }
  }
  if (n_units == 0u) return;  // (nothing suspended: no wave draws a ticket)
  const unsigned long long cut_ticks = split.can_suspend ? split.cut_ticks : 0u;
  const T big = Lim<T>::max();
  bool have = false, overflow = false, exhausted = false;
  uint32_t pair = 0, ncontacts = 0;
  uint32_t unit = 0, my_parent = 0xFFFFFFFFu, my_order = 0;
  unsigned long long t_begin = 0, t_wave = __builtin_readcyclecounter();
  unsigned n_walks = 0, trip_no = 0;
  (void)t_wave; (void)n_walks;
  DMesh m1 = {0, 0, 0, 0}, m2 = {0, 0, 0, 0};
  M3<T> RT_R;
  V3<T> RT_T = mk<T>(T(0), T(0), T(0));
  RT_R.r0 = RT_R.r1 = RT_R.r2 = RT_T;
  int sp = 0, fb1 = -1, fb2 = -1;
  T dlb = big, rec_dist = big, cand_val = big;
  V3<T> np1 = RT_T, np2 = RT_T, nn = RT_T;
  for (;;) {
    if (!have && !exhausted) {
      const CoopUnit u = coop_draw<T, W>(split, ticket, level, unit0, n_units, lig, grp);
      exhausted = u.exhausted;
      if (u.take) {
      BvhSum<T> s;
      if (level) {  // a chunk starts from an empty state
        s.dlb = s.rec_dist = s.cand_val = big;
        s.np1 = s.np2 = s.nn = mk<T>(Lim<T>::nan(), Lim<T>::nan(), Lim<T>::nan());
        s.flags = 0u;
        s.first_child = u.first;
        s.n_child = u.count;
      } else {
        s = *bvh_sum<T>(split, u.index);
      }
      pair = u.pair;
      unit = u.index;
      my_parent = u.parent;
      my_order = u.order;
      m1 = bv.meshes[lib.shapes[wk.shape1[pair]].bvh_index];
      m2 = bv.meshes[lib.shapes[wk.shape2[pair]].bvh_index];
      {
        const Pose<T> tf1 = load_pose(io.tf1, pair), tf2 = load_pose(io.tf2, pair);
        RT_R = tmul(tf1.R, tf2.R);  // traversal_node_setup.h:560-563
        RT_T = tmul(tf1.R, tf2.t - tf1.t);
      }
      if (level) {
        for (uint32_t j = uint32_t(lig); j < s.n_child; j += uint32_t(W)) stack[s.n_child - 1u - j] = split.cut_words[s.first_child + j];  // the chunk's entries
      } else {
        for (uint32_t j = uint32_t(lig); j < s.n_child; j += uint32_t(W)) stack[s.n_child - 1u - j] = split.tasks[s.first_child + j].entry;  // child 0 on top
      }
      sp = int(s.n_child);
      dlb = s.dlb;
      rec_dist = s.rec_dist;
      cand_val = s.cand_val;
      t_begin = __builtin_readcyclecounter();
      ++n_walks;
      np1 = s.np1;
      np2 = s.np2;
      nn = s.nn;
      fb1 = fb2 = -1;
      ncontacts = 0;
      overflow = (s.flags & BVH_SUM_OVERFLOW) != 0 || sp > COOP_CAP;
      if (overflow) sp = 0;
      have = true;
      }
    }
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    __builtin_amdgcn_wave_barrier();
    if (!__any(have)) {
      if (__all(exhausted)) break;
      continue;  // (the units drawn were chunks nobody needs)
    }
    if (have && level && (++trip_no & 15u) == 0u && bvh_moot<T>(split, my_parent, my_order)) {
      // (every 16th trip: the look costs a dependent load per level, every trip 3.3 -> 3.75 ms on cfg4s) a chunk in front of this one
      // has found a contact meanwhile: the sequential walk ended there, nobody reads this summary
      if (lig == 0) bvh_sum<T>(split, split.n_queries + unit0 + unit)->flags = 0u;
      COOP_WALK_END(lig == 0, t_begin);
      have = false;
      continue;
    }
    bool done = have && sp == 0;
    if (have && sp > 0) {
      const int w = min(W, min(sp, max(COOP_CAP - sp, 1)));
      const bool act = lig < w;
      const uint32_t e = act ? stack[sp - 1 - lig] : 0u;
      sp -= w;
      const uint32_t b1 = EN::first(e), b2 = EN::second(e);
      const DNode<T>* const p1n = bv.nodes + m1.node_off + b1;
      const DNode<T>* const p2n = bv.nodes + m2.node_off + b2;
      const int32_t fc1 = act ? p1n->first_child : 0, fc2 = act ? p2n->first_child : 0;
      const bool l1 = fc1 < 0, l2 = fc2 < 0;
      const bool is_leaf = act && l1 && l2, is_int = act && !(l1 && l2);
      T val = big, recv = big;
      bool overlap = false, first = false;
      if (is_int) {
        const DNode<T> n1 = *p1n;
        const DNode<T> n2 = *p2n;
        first = l2 || (!l1 && (sqnorm(n1.extent) > sqnorm(n2.extent)));  // firstOverSecond
        T sq;
        // argument order of the reference: overlap(RT.R, RT.T, model2.bv(b2), model1.bv(b1))
        if (obb_disjoint(RT_R, RT_T, n2, n1, q.security_margin, break_distance2, sq)) {  // updateDistanceLowerBoundFromBV
          const T nd = hsqrt(sq);
          val = nd;
          recv = nd + q.security_margin;
        } else {
          overlap = true;
        }
      }
      const uint64_t omask = gballot(overlap);
      const int f = omask ? __ffsll((unsigned long long)omask) - 1 : W;  // entries [0, f) are visited now
      bool visit = act && lig < f;
      bool contact = false, leaf_ok = false;
      TriLeafOut<T> lo;
      lo.distance = big;
      const uint32_t lb1 = uint32_t(-(fc1 + 1)), lb2 = uint32_t(-(fc2 + 1));
      if (is_leaf && visit) {
        tri_leaf_call<T>(bv.verts + 3 * size_t(m1.vert_off), bv.tris + 3 * size_t(m1.tri_off + lb1), bv.verts + 3 * size_t(m2.vert_off),
                         bv.tris + 3 * size_t(m2.tri_off + lb2), io.tf1, io.tf2, pair, &q, leaf_ps, &lo);
        const T dtc = lo.distance - q.security_margin;  // updateDistanceLowerBoundFromLeaf
        val = dtc;
        recv = lo.distance;
        contact = dtc <= q.collision_distance_threshold;
        leaf_ok = true;
      }
      const uint64_t cmask = gballot(contact);
      const int c = cmask ? __ffsll((unsigned long long)cmask) - 1 : W;  // the first contact in stack order ends the walk
      visit = visit && lig <= c;
      if (!visit) {
        val = big;
        leaf_ok = false;
      }
      const T before = hmin(dlb, group_min_excl_scan<T, W>(val, lig, big));  // the bound as entry `lig` found it
      const bool lowered = visit && val < before;
      const T wmin = group_min_all<T, W>(val);
      if (gballot(lowered)) {
        const int src = __ffsll((unsigned long long)gballot(visit && val == wmin)) - 1;
        dlb = wmin;
        rec_dist = __shfl(recv, src, W);
      }
      const uint64_t wmask = gballot(lowered && leaf_ok);  // the witness: the last leaf that lowered the bound on its visit
      {
        const int L = wmask ? 63 - __clzll((unsigned long long)wmask) : 0;
        const V3<T> c1 = mk<T>(__shfl(lo.p1.x, L, W), __shfl(lo.p1.y, L, W), __shfl(lo.p1.z, L, W));
        const V3<T> c2 = mk<T>(__shfl(lo.p2.x, L, W), __shfl(lo.p2.y, L, W), __shfl(lo.p2.z, L, W));
        const V3<T> cn = mk<T>(__shfl(lo.n.x, L, W), __shfl(lo.n.y, L, W), __shfl(lo.n.z, L, W));
        if (wmask) {
          np1 = c1;
          np2 = c2;
          nn = cn;
          cand_val = __shfl(val, L, W);
        }
      }
      const int cs = c < W ? c : 0;
      const int cb1 = __shfl(int(lb1), cs, W), cb2 = __shfl(int(lb2), cs, W);
      if (c < W) {  // canStop() (num_max_contacts == 1: the split forms only)
        fb1 = cb1;
        fb2 = cb2;
        ncontacts = 1;
        sp = 0;
      } else {
        const int cnt = !act || lig < f ? 0 : (overlap ? 2 : 1);
        const uint64_t mm2 = gballot(cnt == 2), mm1 = gballot(cnt == 1);
        const uint64_t deeper = ~((uint64_t(2) << lig) - 1);  // window entries behind this one (pushed first)
        const int pos = sp + 2 * __popcll(mm2 & deeper) + __popcll(mm1 & deeper);
        if (cnt == 2) {
          uint32_t ea, eb;
          if (first) {
            ea = EN::pack(uint32_t(fc1), b2);
            eb = EN::pack(uint32_t(fc1) + 1u, b2);
          } else {
            ea = EN::pack(b1, uint32_t(fc2));
            eb = EN::pack(b1, uint32_t(fc2) + 1u);
          }
          stack[pos] = eb;      // second child below
          stack[pos + 1] = ea;  // first child on top
        } else if (cnt == 1) {
          stack[pos] = e;
        }
        sp += 2 * __popcll(mm2) + __popcll(mm1);
        if (sp > COOP_CAP + COOP_SLACK - 2) {
          overflow = true;
          sp = 0;
        }
      }
      done = sp == 0;
      if (cut_ticks && sp > COOP_CHUNK && __builtin_readcyclecounter() - t_begin > cut_ticks) {
        // ---- the walk has had its time: what is left of it becomes chunks for the next launch (coop_cut)
        if (coop_cut<T, W>(split, level ? split.n_queries + unit0 + unit : unit, pair, my_parent, my_order, stack, static_cast<const T*>(nullptr), sp, lig, dlb, rec_dist, cand_val,
                           np1, np2, nn, overflow)) {
          COOP_CUT_COUNT(lig == 0);
          COOP_WALK_END(lig == 0, t_begin);
          have = false;
          sp = 0;
        } else {
          t_begin = __builtin_readcyclecounter();  // (no room: the walk goes on and asks again after another budget)
        }
      }
    }
    if (done && level) {
      // ---- a chunk's summary (k_bvh_combine folds it into the walk it was cut from)
      if (lig == 0) {
        coop_write_sum<T>(split, split.n_queries + unit0 + unit, dlb, rec_dist, cand_val, np1, np2, nn, fb1, fb2, ncontacts, 0u, 0u,
                          overflow ? BVH_SUM_OVERFLOW : 0u, my_parent, my_order);
        if (ncontacts) coop_report_contact<T>(split, my_parent, my_order);
      }
      COOP_WALK_END(lig == 0, t_begin);
      have = false;
      done = false;
    }
    if (done) {
      if (lig == 0) {
        bvh_sum<T>(split, unit)->flags = 0u;  // (not cut: k_bvh_combine has nothing to fold for this query)
        PairOut<T> o;
        o.distance = rec_dist;
        o.normal = nn;
        o.p1 = np1;
        o.p2 = np2;
        o.gjk_status = GJK_DID_NOT_RUN;
        o.epa_status = EPA_DID_NOT_RUN;
        o.gjk_iters = o.epa_iters = 0;
        store_bvh_record(io, pair, o, ncontacts, fb1, fb2, overflow);
      }
      COOP_WALK_END(lig == 0, t_begin);
      have = false;
    }
  }
  COOP_WAVE_END(t_wave, n_walks);
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
  __shared__ uint32_t stack_n[G][BS_STACK];
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
// k_bvh_shape_distance_lane: distance() between a mesh and a solid with ONE QUERY PER LANE (the streaming scheme of
// k_bvh_distance; MeshShapeDistanceTraversalNodeOBBRSS + distanceRecurse as in hfcl_bvh_shape.hpp: mesh_shape_distance).
// The solid's OBBRSS comes from k_shape_obbrss's table (re-read at every step: 15 numbers that do not wait for the node
// gather), the leaf is the SOLID collide form's (solid_leaf_call), the witness of the minimum lives in the record.  A leaf
// that needs EPA ends the walk -- its distance is <= 0 and every bound left on the stack >= 0 -- and is finished by
// k_bvh_shape_finish.  Bounds travel as in k_bvh_distance (4 bytes, rounded down, exact re-evaluation in the band).
// ---------------------------------------------------------------------------------------
#ifndef HFCL_BSD_PARK_MIN
#define HFCL_BSD_PARK_MIN 16
#endif
template <typename T>
__global__ void __launch_bounds__(BVHD_BLOCK) __attribute__((amdgpu_waves_per_eu(2, 8)))
k_bvh_shape_distance_lane(Work wk, LibView<T> lib, BvhView<T> bv, IO<T> io, QParams<T> q, BvhSpill spill) {
  constexpr int STACK = BVHD_STACK;
  typedef typename std::conditional<sizeof(T) == 8, uint32_t, float>::type BD;
  __shared__ uint32_t stack_e[STACK][BVHD_BLOCK];
  __shared__ BD stack_d[STACK][BVHD_BLOCK];
  __shared__ T w0_slab[W0Lds<T, BVHD_BLOCK>::WORDS];
  const W0Lds<T, BVHD_BLOCK> leaf_ps{w0_slab + threadIdx.x};
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
  const uint32_t cnt = wk.counts[B_BVHSHAPE];
  uint32_t* const ticket = &wk.counts[CTR_SHAPE_TICKET];
  const RssQuery<T>* const table = reinterpret_cast<const RssQuery<T>*>(wk.shape_oq);
  const int tid = threadIdx.x, lane = tid & 63;
  const T nanv = Lim<T>::nan();
  bool live = false, pending = false, exhausted = false, swapped = false, overflow = false, unsupported = false, deferred = false;
  bool have_leaf = false;
  uint32_t pair = 0, solid_id = 0, leaf_prim = 0, steps = 0;
  DMesh m1 = {0, 0, 0, 0};
  Pose<T> tfm;
  tfm.R.r0 = tfm.R.r1 = tfm.R.r2 = tfm.t = mk<T>(T(0), T(0), T(0));
  T mind = Lim<T>::max();
  int fb1 = -1, sp = 0;
  V3<T> guess = mk<T>(T(1), T(0), T(0));
  // lower bound between the solid's volume and mesh node b (distance(tf1.R, tf1.T, model2_bv, model1.bv(b)))
  auto bound_of = [&](uint32_t b) -> T {
    const RssQuery<T> rq = table[pair];
    DNode<T> n2;
    n2.axes = rq.axes;
    DRss<T> r2;
    r2.Tr = rq.Tr;
    r2.l0 = rq.l0;
    r2.l1 = rq.l1;
    r2.r = rq.r;
    return rss_lower_bound(tfm.R, tfm.t, n2, r2, bv.nodes[m1.node_off + b], bv.rss[m1.node_off + b]);
  };
  auto leaf = [&](uint32_t prim) {
    SolidLeafIn<T> in{bv.verts + 3 * size_t(m1.vert_off), bv.tris + 3 * size_t(m1.tri_off + prim), lib.shapes, lib.verts,
                      swapped ? io.tf2 : io.tf1, swapped ? io.tf1 : io.tf2, reinterpret_cast<ShapeDeferItem<T>*>(wk.shape_defer),
                      &wk.counts[CTR_SHAPE_DEFER], wk.shape_defer_cap, pair, solid_id, prim, 0xFFFFFFFFu, 0u, mind, fb1};
    SolidLeafOut<T> lo;
    if (solid_leaf_call<T>(in, &q, leaf_ps, guess, &lo)) {
      deferred = true;  // k_bvh_shape_finish writes the record
      sp = 0;
      return;
    }
    guess = lo.guess;
    if (mind > lo.distance) {  // DistanceResult::update
      mind = lo.distance;
      fb1 = int(prim);
      store_witness(io, pair, swapped ? lo.p2 : lo.p1, swapped ? lo.p1 : lo.p2, swapped ? -lo.n : lo.n);
    }
  };
  for (;;) {
    if (live && sp == 0 && !have_leaf) {
      live = false;
      pending = !deferred;
    }
    const uint64_t live_mask = __ballot(live);
    const int n_live = __popcll(live_mask);
    if (exhausted ? n_live == 0 : 64 - n_live >= BVH_REFILL_MIN) {
      if (pending) {
        if (unsupported) {
          store_unsupported(io, pair);
        } else {
          // b1 = the triangle, b2 = NONE whatever the operand order (distance.cpp:84-88 swaps o1 / o2 only)
          store_bvh_record_head(io, pair, mind, mind <= T(0) ? 0x80000000u : 0u, fb1, -1, overflow);
          write_guess<T>(io, pair, guess, 0, 0);
        }
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
          pair = wk.lists[size_t(B_BVHSHAPE) * wk.n + it];
          const uint32_t id1 = wk.shape1[pair], id2 = wk.shape2[pair];
          swapped = lib.kinds[id1] != uint8_t(K_BVH);  // distance.cpp:74-88
          solid_id = swapped ? id1 : id2;
          m1 = bv.meshes[lib.shapes[swapped ? id2 : id1].bvh_index];
          tfm = load_pose(swapped ? io.tf2 : io.tf1, pair);
          mind = Lim<T>::max();
          fb1 = -1;
          overflow = deferred = have_leaf = false;
          steps = 0;
          guess = initial_guess<T>(io, q, pair);
          const T probe = table[pair].r;
          unsupported = !(probe == probe);
          sp = 0;
          if (unsupported) {
            pending = true;
          } else {
            const V3<T> nan3 = mk<T>(nanv, nanv, nanv);
            store_witness(io, pair, nan3, nan3, nan3);
            live = true;
            sp = 1;  // (leaf() of a deferred preprocess() sets it back to 0)
            stack_e[0][tid] = 0u;
            stack_d[0][tid] = bound_down(T(-1));
            leaf(0u);  // preprocess(): triangle 0
          }
        }
      }
      if (base + uint32_t(n_need) >= cnt) exhausted = true;
      continue;
    }
    // A triangle against the solid is a GJK run: ten times a BV step.  A lane that pops one waits until HFCL_BSD_PARK_MIN
    // lanes do (or nobody can walk) and the wave runs the leaves together (k_bvh_distance, whose triangle pairs cost what a
    // BV step costs, evaluates them where they are popped).
    for (;;) {
      if (spill.budget && live && sp > 0 && !have_leaf && steps >= spill.budget) {
        // a long walk: its stack and its minimum go to a record, a wave takes it over (k_bvh_shape_distance_coop)
        ShapeDistSusp<T>* r = reinterpret_cast<ShapeDistSusp<T>*>(spill.susp) + atomicAdd(spill.susp_count, 1u);
        r->pair = pair;
        r->sp = uint32_t(sp);
        r->fb1 = fb1;
        r->pad_ = 0;
        r->mind = mind;
        for (int k = 0; k < sp; ++k) {
          r->entry[k] = stack_e[k][tid];
          r->bound[k] = bound_value(stack_d[k][tid]);
        }
        sp = 0;
        deferred = true;  // (no record from this lane)
      }
      const bool run = live && sp > 0 && !have_leaf;
      const int n_run = __popcll(__ballot(run)), n_wait = __popcll(__ballot(have_leaf));
      if (n_run == 0 || n_wait >= HFCL_BSD_PARK_MIN || (!exhausted && 64 - n_run - n_wait >= BVH_REFILL_MIN)) break;
      if (!run) continue;
      ++steps;
      --sp;
      const uint32_t b = stack_e[sp][tid];
      const BD dc = stack_d[sp][tid];
      const T de = bound_value(dc);
      if (de >= T(0) && de >= mind) continue;  // canStop(d)
      if constexpr (sizeof(T) == 8) {
        if (de >= T(0) && __hiloint2double(int(dc) + 1, 0) > mind) {  // the exact bound may reach the minimum: ask it
          if (bound_of(b) >= mind) continue;
        }
      }
      const int32_t fc = bv.nodes[m1.node_off + b].first_child;
      if (fc < 0) {
        have_leaf = true;
        steps += 15;  // (a GJK leaf counts sixteen steps)
        leaf_prim = uint32_t(-(fc + 1));
        continue;
      }
      const uint32_t a1 = uint32_t(fc), c1 = a1 + 1;
      const T d1 = bound_of(a1), d2 = bound_of(c1);
      if (sp + 2 > STACK) {
        overflow = true;
        sp = 0;
        continue;
      }
      const bool c_first = d2 < d1;  // the nearer child is visited first
      stack_e[sp][tid] = c_first ? a1 : c1;
      stack_d[sp][tid] = bound_down(c_first ? d1 : d2);
      ++sp;
      stack_e[sp][tid] = c_first ? c1 : a1;
      stack_d[sp][tid] = bound_down(c_first ? d2 : d1);
      ++sp;
    }
    if (have_leaf) {
      have_leaf = false;
      leaf(leaf_prim);
    }
  }
}

// ---------------------------------------------------------------------------------------
// k_bvh_shape_distance_coop: the long mesh x solid distance() walks, a wave per query (the scheme of k_bvh_distance_coop below;
// the triangles in front of the first node that is split run their leaves -- per-lane GJK -- side by side; a triangle that
// needs EPA ends the walk as in the lane kernel: only the FIRST such triangle in stack order queues its item).
// ---------------------------------------------------------------------------------------
template <typename T>
__global__ void __launch_bounds__(64) __attribute__((amdgpu_waves_per_eu(2, 8)))
k_bvh_shape_distance_coop(Work wk, LibView<T> lib, BvhView<T> bv, IO<T> io, QParams<T> q, BvhSpill spill) {
  constexpr int CAP = 960, SLACK = 64;
  __shared__ uint32_t stack_e[CAP + SLACK];
  __shared__ T stack_d[CAP + SLACK];
  __shared__ T w0_slab[W0Lds<T, 64>::WORDS];
  const W0Lds<T, 64> leaf_ps{w0_slab + threadIdx.x};
  const int lane = threadIdx.x;
  const uint32_t n_susp = *spill.susp_count;
  const RssQuery<T>* const table = reinterpret_cast<const RssQuery<T>*>(wk.shape_oq);
  const T big = Lim<T>::max();
  for (uint32_t qi = blockIdx.x; qi < n_susp; qi += gridDim.x) {
    const ShapeDistSusp<T>* const r = reinterpret_cast<const ShapeDistSusp<T>*>(spill.susp) + qi;
    const uint32_t pair = r->pair;
    const uint32_t id1 = wk.shape1[pair], id2 = wk.shape2[pair];
    const bool swapped = lib.kinds[id1] != uint8_t(K_BVH);
    const uint32_t solid_id = swapped ? id1 : id2;
    const DMesh m1 = bv.meshes[lib.shapes[swapped ? id2 : id1].bvh_index];
    const Pose<T> tfm = load_pose(swapped ? io.tf2 : io.tf1, pair);
    const RssQuery<T> rq = table[pair];
    DNode<T> n2;
    n2.axes = rq.axes;
    DRss<T> r2;
    r2.Tr = rq.Tr;
    r2.l0 = rq.l0;
    r2.l1 = rq.l1;
    r2.r = rq.r;
    auto bound_of = [&](uint32_t b) -> T { return rss_lower_bound(tfm.R, tfm.t, n2, r2, bv.nodes[m1.node_off + b], bv.rss[m1.node_off + b]); };
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
    int fb1 = r->fb1;
    bool overflow = false, deferred = false;
    const V3<T> guess0 = initial_guess<T>(io, q, pair);  // (walks whose leaves hand a cached guess on are not suspended)
    V3<T> guess = guess0;
    while (sp > 0) {
      const int w = min(64, min(sp, max(CAP - sp, 1)));
      const bool act = lane < w;
      const uint32_t b = act ? stack_e[sp - 1 - lane] : 0u;
      const T db = act ? stack_d[sp - 1 - lane] : big;
      sp -= w;
      const bool alive = act && !(db >= T(0) && db >= mind);  // canStop(d), with the minimum of the moment
      const int32_t fc = alive ? bv.nodes[m1.node_off + b].first_child : 0;
      const bool is_leaf = alive && fc < 0, split = alive && fc >= 0;
      T d1 = big, d2 = big;
      if (split) {
        d1 = bound_of(uint32_t(fc));
        d2 = bound_of(uint32_t(fc) + 1u);
      }
      const uint64_t smask = __ballot(split);
      const int f = smask ? __ffsll((unsigned long long)smask) - 1 : 64;  // the triangles in front of it are visited now
      const bool visit = is_leaf && lane < f;
      T val = big;
      bool to_epa = false;
      SolidLeafOut<T> lo;
      lo.distance = big;
      const uint32_t prim = uint32_t(-(fc + 1));
      auto run_leaf = [&](bool push, T bound, int prev) -> bool {
        SolidLeafIn<T> in{bv.verts + 3 * size_t(m1.vert_off), bv.tris + 3 * size_t(m1.tri_off + prim), lib.shapes, lib.verts,
                          swapped ? io.tf2 : io.tf1, swapped ? io.tf1 : io.tf2, push ? reinterpret_cast<ShapeDeferItem<T>*>(wk.shape_defer) : nullptr,
                          push ? &wk.counts[CTR_SHAPE_DEFER] : nullptr, wk.shape_defer_cap, pair, solid_id, prim, 0xFFFFFFFFu, 0u, bound, prev};
        return solid_leaf_call<T>(in, &q, leaf_ps, guess0, &lo);
      };
      if (visit) {
        to_epa = run_leaf(false, T(0), -1);
        if (!to_epa) val = lo.distance;
      }
      // The evaluated triangles, in stack order, exactly as the lane's walk takes them: a triangle counts only if its bound does
      // not let it be skipped at ITS turn (canStop with the minimum as it stands then -- a bound is clamped at 0, so after a
      // penetration every bounded entry is skipped, and the bounds of unbounded solids are NaN and never skip), the minimum is
      // lowered by strictly smaller distances only, and a counted triangle that needs EPA ends the walk.
      const uint64_t emask = __ballot(to_epa);
      int start = 0, src = -1;
      bool epa_end = false;
      T run = mind;
      for (;;) {
        const uint64_t em = emask & ~((uint64_t(1) << start) - 1);
        const int c = em ? __ffsll((unsigned long long)em) - 1 : 64;  // the next triangle that needs EPA
        for (;;) {  // the record-setting triangles in front of it
          const bool cand = visit && !to_epa && lane >= start && lane < c && !(db >= T(0) && db >= run) && val < run;
          const uint64_t m = __ballot(cand);
          if (!m) break;
          src = __ffsll((unsigned long long)m) - 1;
          run = __shfl(val, src);
          start = src + 1;
        }
        if (c >= 64) break;
        const T dbc = __shfl(db, c);
        if (!(dbc >= T(0) && dbc >= run)) {  // it is visited: the walk ends here
          epa_end = true;
          start = c;
          break;
        }
        start = c + 1;  // (skipped like any entry whose bound cannot beat the minimum)
        if (start >= 64) break;
      }
      if (src >= 0) {  // DistanceResult::update
        mind = run;
        fb1 = __shfl(int(prim), src);
        if (lane == src) store_witness(io, pair, swapped ? lo.p2 : lo.p1, swapped ? lo.p1 : lo.p2, swapped ? -lo.n : lo.n);
        guess = mk<T>(__shfl(lo.guess.x, src), __shfl(lo.guess.y, src), __shfl(lo.guess.z, src));
      }
      if (epa_end) {
        if (lane == start) run_leaf(true, mind, fb1);  // its leaf once more, with the EPA item: k_bvh_shape_finish writes the record
        deferred = true;
        sp = 0;
        break;
      }
      // the stack again, in order: visited triangles and dropped entries are gone, a split node is its two children (the
      // nearer one on top), a triangle behind the first split stays
      const int cnt = split ? 2 : ((is_leaf && lane >= f) ? 1 : 0);
      const uint64_t m2b = __ballot(cnt == 2), m1b = __ballot(cnt == 1);
      const uint64_t deeper = ~((uint64_t(2) << lane) - 1);
      const int pos = sp + 2 * __popcll(m2b & deeper) + __popcll(m1b & deeper);
      if (cnt == 2) {
        const uint32_t a1 = uint32_t(fc), c1 = a1 + 1u;
        const bool c_first = d2 < d1;  // the nearer child is visited first
        stack_e[pos] = c_first ? a1 : c1;
        stack_d[pos] = c_first ? d1 : d2;
        stack_e[pos + 1] = c_first ? c1 : a1;
        stack_d[pos + 1] = c_first ? d2 : d1;
      } else if (cnt == 1) {
        stack_e[pos] = b;
        stack_d[pos] = db;
      }
      sp += 2 * __popcll(m2b) + __popcll(m1b);
      if (sp > CAP + SLACK - 2) {
        overflow = true;
        sp = 0;
      }
      __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
      __builtin_amdgcn_wave_barrier();
    }
    if (lane == 0 && !deferred) {
      // b1 = the triangle, b2 = NONE whatever the operand order (distance.cpp:84-88 swaps o1 / o2 only)
      store_bvh_record_head(io, pair, mind, mind <= T(0) ? 0x80000000u : 0u, fb1, -1, overflow);
      write_guess<T>(io, pair, guess, 0, 0);
    }
  }
}

// ---------------------------------------------------------------------------------------
// k_bvh_shape_distance_pool: the long mesh x solid distance() walks with the scheme of k_bvh_distance_pool (hfcl_k_bvhd.hip, round 4):
// SDP_Q walks per wave, each with its stack in LDS in DFS order; the child tests of every mesh node that is split go into one list for
// the wave, worked off 64 at a time (one rss_lower_bound per lane against the walk's solid, on the packed DNodeD records); the
// triangles of the wave's windows are evaluated together, one GJK run per lane, whichever walk they belong to; a marker per walk
// (entries that stand behind the triangle of the minimum) keeps the reference's choice among equal distances whatever the order of
// the evaluations.  What this row adds to the mesh x mesh form: a triangle inside the solid (its leaf asks for EPA) ends the walk at
// its turn, so the FIRST such triangle in DFS order is the walk's result -- it ranks below every distance, ties by DFS order --, the
// entries behind it are dropped, the ones in front are finished (they may hold an earlier one), and when the stack is empty its
// lane runs the leaf once more to queue the EPA item (k_bvh_shape_finish writes the record), as k_bvh_shape_distance_coop does.
// Suspended walks never hand a cached guess on and their final guess is not read (the host does not suspend those).
// ---------------------------------------------------------------------------------------
// (124 entries before the windows narrow -- it was 160 --: with the chain word per entry the fp64 block is 20 288 B, eight waves per CU)
constexpr int SDP_Q = 4, SDP_SEG = 64 / SDP_Q, SDP_CAPW = 124, SDP_CAP = SDP_CAPW + BVHD_STACK + 8;
template <typename T>
__global__ void __launch_bounds__(64) __attribute__((amdgpu_waves_per_eu(2, 8)))
k_bvh_shape_distance_pool(Work wk, LibView<T> lib, BvhView<T> bv, IO<T> io, QParams<T> q, BvhSpill spill) {
  constexpr int Q = SDP_Q, SEG = SDP_SEG;
  __shared__ uint32_t st_x[Q][SDP_CAP];  // bit 31: a triangle (bits 0-30 its id); else a mesh node whose first child is bits 0-30
  __shared__ T st_d[Q][SDP_CAP];
  __shared__ float st_c[Q][SDP_CAP];  // the largest bound among the entry's ancestors where it exceeds the entry's own, rounded up (0: none; k_bvh_distance_pool)
  __shared__ T q_tf[Q][12];    // pose of the mesh (R rows, t): the R0, T0 of the bound
  __shared__ T q_rss[Q][15];   // the solid's RSS in the mesh frame's terms (RssQuery): axes, Tr, l0, l1, r
  __shared__ uint32_t q_ids[Q][6];  // pair, solid id, swapped, vert_off, tri_off, node_off
  __shared__ uint32_t t_node[128], t_x[128];
  __shared__ uint8_t t_q[128];
  __shared__ T t_res[128];
  __shared__ T w0_slab[W0Lds<T, 64>::WORDS];
  const W0Lds<T, 64> leaf_ps{w0_slab + threadIdx.x};
  const int lane = threadIdx.x, qs = lane / SEG, j = lane % SEG;
  const uint64_t lt_mask = (uint64_t(1) << lane) - 1;
  const uint64_t segm = ((uint64_t(1) << SEG) - 1) << (qs * SEG);
  const uint64_t deeper = segm & ~((uint64_t(2) << lane) - 1);
  const uint32_t n_susp = *spill.susp_count;
  const RssQuery<T>* const table = reinterpret_cast<const RssQuery<T>*>(wk.shape_oq);
  const int leaf_min = int(spill.pool_leaf_min), starve = int(spill.pool_starve);
  const T big = Lim<T>::max();
  auto sync = []() {
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    __builtin_amdgcn_wave_barrier();
  };
  // slot state, identical in the SEG lanes of a slot
  // `ended`: a triangle that ENDS the sequential walk at its turn has been found (p marks it): it asks for EPA (ended_epa), or its
  // leaf reported a negative distance without EPA (a closed form) while its bound is a number -- bounds are clamped at 0, so behind
  // such a triangle every bounded entry is skipped: the walk's result is the FIRST such triangle in DFS order, not the smallest value
  // `ordered`: the slot walks a flagged record again, in the reference's order (k_bvh_distance_pool, hfcl_k_bvhd.hip; below); `redo`: it is
  // about to take that record again; `touched`: the ordered walk has set a minimum or ended (else the record's witness, which the pooled
  // pass has overwritten, is evaluated again at the end)
  bool active = false, exhausted = false, ended = false, ended_epa = false, overflow = false, flagged = false, ordered = false, redo = false, touched = false;
  int sp = 0, p = 0, fb1 = -1, end_prim = -1;
  T mind = big, end_val = T(0), margin = T(0), pooled = T(0), cut = big;
  uint32_t pair = 0, susp_it = 0;
  for (;;) {
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
        it = base + uint32_t(__popcll(idle & ((uint64_t(1) << (qs * SEG)) - 1)));
      }
      if (redo) it = susp_it;  // the slot's own record once more
      if ((!active && it < n_susp) || redo) {
        const ShapeDistSusp<T>* const r = reinterpret_cast<const ShapeDistSusp<T>*>(spill.susp) + it;
        pair = r->pair;
        ordered = redo;
        redo = touched = false;
        const uint32_t id1 = wk.shape1[pair], id2 = wk.shape2[pair];
        const bool swapped = lib.kinds[id1] != uint8_t(K_BVH);
        const DMesh m1 = bv.meshes[lib.shapes[swapped ? id2 : id1].bvh_index];
        sp = int(r->sp);
        p = sp;  // everything on the stack comes after what the lane has visited
        mind = r->mind;
        fb1 = r->fb1;
        ended = ended_epa = false;
        end_prim = -1;
        overflow = false;
        susp_it = it;
        flagged = false;  // (the lane's minimum is one the sequential walk holds)
        const RssQuery<T> rq0 = table[pair];
        const Pose<T> ltf = load_pose(swapped ? io.tf2 : io.tf1, pair);
        DNodeD<T> lS;
        lS.axes = rq0.axes;
        lS.Tr = rq0.Tr;
        lS.l0 = rq0.l0;
        lS.l1 = rq0.l1;
        lS.r = rq0.r;
        // (not vectorised: hipcc 7.2's instruction selection dies on the masked gather the loop vectoriser makes of this loop)
#pragma clang loop vectorize(disable) interleave(disable)
        for (int k = j; k < sp; k += SEG) {
          const uint32_t b = r->entry[k];
          const DNodeD<T>* const mn = bv.dnodes + m1.node_off + b;
          const int32_t fc = mn->first_child;
          st_x[qs][k] = fc < 0 ? (0x80000000u | uint32_t(-(fc + 1))) : uint32_t(fc);
          // fp64: the exact bounds again (the lane's stack carries them rounded down to 32 bits); the root's -1 stays
          T bk = r->bound[k];
          if constexpr (sizeof(T) == 8) {
            if (bk >= T(0)) bk = rss_lower_bound(ltf.R, ltf.t, lS, *mn);
          }
          st_d[qs][k] = bk;
          st_c[qs][k] = 0.0f;  // (its ancestors were decided by the lane, in the reference's order)
        }
        {
          // what a bound may exceed a distance beneath it by: a few ulps of the scene's size (the solid's volume in the mesh's frame and
          // the mesh's root volume), as in k_bvh_distance_pool -- an entry IN FRONT of the minimum is kept within that margin
          const DNodeD<T>* const root = bv.dnodes + m1.node_off;
          const T scale = habs(rq0.Tr.x) + habs(rq0.Tr.y) + habs(rq0.Tr.z) + rq0.l0 + rq0.l1 + T(2) * rq0.r + habs(root->Tr.x) + habs(root->Tr.y) +
                          habs(root->Tr.z) + root->l0 + root->l1 + T(2) * root->r;
          margin = T(16) * Lim<T>::eps() * scale;
          // ... and what a leaf's distance may fall short of the true one by: the supports of Box, Cone and Cylinder are inflated by 1e-10
          // (support_functions.cpp:146,229,281: hfcl_shapes.hpp), so GJK measures to a solid that much larger than the one the bound is
          // taken to -- a triangle 0.89 from a box had its leaf bound 2.9e-11 ABOVE its distance
          const DShape<T> sd = lib.shapes[swapped ? id1 : id2];
          if (sd.kind == K_BOX || sd.kind == K_CONE || sd.kind == K_CYLINDER) margin += T(2e-10) * (habs(sd.p0) + habs(sd.p1) + habs(sd.p2));
        }
        if (j == 0) {
          const Pose<T>& tfm = ltf;
          T* o = q_tf[qs];
          o[0] = tfm.R.r0.x; o[1] = tfm.R.r0.y; o[2] = tfm.R.r0.z; o[3] = tfm.R.r1.x; o[4] = tfm.R.r1.y; o[5] = tfm.R.r1.z;
          o[6] = tfm.R.r2.x; o[7] = tfm.R.r2.y; o[8] = tfm.R.r2.z; o[9] = tfm.t.x; o[10] = tfm.t.y; o[11] = tfm.t.z;
          const RssQuery<T>& rq = rq0;
          T* g = q_rss[qs];
          g[0] = rq.axes.r0.x; g[1] = rq.axes.r0.y; g[2] = rq.axes.r0.z; g[3] = rq.axes.r1.x; g[4] = rq.axes.r1.y; g[5] = rq.axes.r1.z;
          g[6] = rq.axes.r2.x; g[7] = rq.axes.r2.y; g[8] = rq.axes.r2.z; g[9] = rq.Tr.x; g[10] = rq.Tr.y; g[11] = rq.Tr.z;
          g[12] = rq.l0; g[13] = rq.l1; g[14] = rq.r;
          uint32_t* c = q_ids[qs];
          c[0] = pair; c[1] = swapped ? id1 : id2; c[2] = swapped ? 1u : 0u; c[3] = m1.vert_off; c[4] = m1.tri_off; c[5] = m1.node_off;
        }
        active = true;
      }
      sync();
    }
    if (__ballot(active) == 0) break;
    // ---- the windows
    const int w = active ? min(SEG, min(sp, max(SDP_CAPW - sp, 1))) : 0;
    const int base_i = sp - w;
    const bool act = j < w;
    const int idx = sp - 1 - j;
    uint32_t x = 0u;
    T db = big;
    float ch = 0.0f;
    if (act) {
      x = st_x[qs][idx];
      db = st_d[qs][idx];
      ch = st_c[qs][idx];
    }
    // behind a triangle that ends the walk nothing is visited; else canStop(bound) with the slot's minimum: behind the triangle of
    // the minimum the sequential walk's test, in front of it a tie is kept (k_bvh_distance_pool; NaN bounds never skip)
    // (ordered mode: what lies above the cut is gone, everything else is decided below)
    const bool alive = act && (ordered ? !(db > cut) : !(ended && idx < p) && !(db >= T(0) && (idx < p && !ended ? db >= mind : db > mind + margin)));
    bool is_leaf = alive && (x >> 31) != 0u, split = alive && (x >> 31) == 0u, held = false;
    // ---- Ordered mode (k_bvh_distance_pool): a node whose bound is below the pooled minimum -- or 0: the sequential walk's minimum is
    // positive until the walk ends -- is split by that walk whatever its minimum, and so here, wherever it stands; a node of the band
    // between that and the cut is decided at the top of the stack only; the triangles in front of the slot's first node are evaluated
    // together and applied in stack order (the scan below does that in either mode); everything else waits for its turn.
    if (__ballot(ordered) != 0) {
      const uint64_t boxm = __ballot(split) & segm;
      const int f = boxm ? __ffsll((unsigned long long)boxm) - 1 - qs * SEG : SEG;
      if (ordered) {
        if (split && !(db < pooled || db <= T(0))) {
          if (j == 0) {
            if (db >= mind) split = false;  // canStop at its turn
          } else {
            split = false;
            held = true;
          }
        } else if (is_leaf) {
          if (j > f) {
            is_leaf = false;  // behind a node: not its turn yet
            held = true;
          } else if (db >= T(0) && db >= mind) {
            is_leaf = false;  // canStop with a minimum that is not above the one of its turn
          }
        }
      }
    }
    // ---- the children's bounds of every split node of the wave in one list
    const uint64_t smask = __ballot(split);
    const int ns = __popcll(smask), k2 = 2 * __popcll(smask & lt_mask);
    if (split) {
      t_node[k2] = x;
      t_node[k2 + 1] = x + 1u;
      t_q[k2] = uint8_t(qs);
      t_q[k2 + 1] = uint8_t(qs);
    }
    sync();
    for (int tb = 0; tb < 2 * ns; tb += 64) {
      const int t = tb + lane;
      if (t < 2 * ns) {
        const uint32_t ts = t_q[t];
        const T* const f = q_tf[ts];
        const T* const g = q_rss[ts];
        M3<T> R0;
        R0.r0 = mk<T>(f[0], f[1], f[2]);
        R0.r1 = mk<T>(f[3], f[4], f[5]);
        R0.r2 = mk<T>(f[6], f[7], f[8]);
        const V3<T> T0 = mk<T>(f[9], f[10], f[11]);
        DNodeD<T> S;
        S.axes.r0 = mk<T>(g[0], g[1], g[2]);
        S.axes.r1 = mk<T>(g[3], g[4], g[5]);
        S.axes.r2 = mk<T>(g[6], g[7], g[8]);
        S.Tr = mk<T>(g[9], g[10], g[11]);
        S.l0 = g[12];
        S.l1 = g[13];
        S.r = g[14];
        const DNodeD<T> M = bv.dnodes[q_ids[ts][5] + t_node[t]];
        t_res[t] = rss_lower_bound(R0, T0, S, M);
        t_x[t] = M.first_child < 0 ? (0x80000000u | uint32_t(-(M.first_child + 1))) : uint32_t(M.first_child);
      }
    }
    sync();
    T d1 = big, d2 = big;
    uint32_t xa = 0u, xc = 0u;
    if (split) {
      d1 = t_res[k2];
      d2 = t_res[k2 + 1];
      xa = t_x[k2];
      xc = t_x[k2 + 1];
    }
    // ---- the triangles, once they are worth a pass (one GJK run per lane)
    const int nl = __popcll(__ballot(is_leaf));
    const bool do_leaves = nl > 0 && (nl >= leaf_min || 2 * ns < starve || ns == 0);  // (nothing but triangles in the windows: they run, whatever the knobs say)
    int jw = -1;
    T run_at = mind;  // the slot's minimum as of this lane's own turn in the pass, and whether the pass had set one / ended the walk by then
    bool upd_at = false, stop_at = false;
    if (do_leaves) {
      T val = big;
      bool to_epa = false;
      SolidLeafOut<T> lo;
      lo.distance = big;
      const uint32_t prim = x & 0x7FFFFFFFu;
      if (is_leaf) {
        const uint32_t* c = q_ids[qs];
        const bool swapped = c[2] != 0u;
        SolidLeafIn<T> in{bv.verts + 3 * size_t(c[3]), bv.tris + 3 * size_t(c[4] + prim), lib.shapes, lib.verts,
                          swapped ? io.tf2 : io.tf1, swapped ? io.tf1 : io.tf2, nullptr, nullptr, 0u, pair, c[1], prim, 0xFFFFFFFFu, 0u, T(0), -1};
        to_epa = solid_leaf_call<T>(in, &q, leaf_ps, initial_guess<T>(io, q, pair), &lo);
        if (!to_epa) val = lo.distance;
      }
      // DistanceResult::update over the slot's evaluated triangles IN THE ORDER OF THEIR VISITS (window index 0 = the top of the stack
      // = visited first).  The sequential walk tests an entry against the minimum of ITS turn: a triangle behind a minimum that an
      // earlier triangle of this very pass has set is visited only if its bound is below that minimum -- evaluated together, a
      // triangle whose bound exceeds its own distance by an ulp would otherwise lower a minimum the reference never lowers (two
      // records in 20 000 box queries, profiles/r05_c).  A triangle that ends the walk (asks for EPA, or a negative closed-form
      // distance behind a bound) is the result if the walk reaches it; against the standing result a tie wins only in front of it.
      const bool ends = is_leaf && (to_epa || (val < T(0) && db >= T(0)));
      // a bound of the triangle's chain above its distance (an ending triangle: above 0 -- the sequential walk's minimum is positive until
      // the walk ends, so bounds of 0 never turn it away): the sequential walk may not have come here, the walk is re-run in order
      const T chain = ch != 0.0f ? T(ch) : db;
      const bool viol = is_leaf && chain > (ends ? T(0) : val);
      T run = mind;
      bool upd = false, stop = false;
      int win_min = -1, win_end = -1;
#pragma unroll 4
      for (int s = 0; s < SEG; ++s) {
        const int src = qs * SEG + s;
        if (j == s) {
          run_at = run;
          upd_at = upd;
          stop_at = stop;
        }
        const bool leaf_s = __shfl(int(is_leaf), src) != 0, ends_s = __shfl(int(ends), src) != 0, viol_s = __shfl(int(viol), src) != 0;
        const T db_s = __shfl(db, src), val_s = __shfl(val, src);
        const int idx_s = sp - 1 - s;
        const bool visited = leaf_s && !stop && !(upd && db_s >= T(0) && db_s >= run);
        const bool take_end = visited && ends_s && (!ended || idx_s >= p);
        const bool take_min = visited && !ends_s && !ended && (val_s < run || (val_s == run && !upd && idx_s >= p && !ordered));
        if (take_min) {
          run = val_s;
          win_min = s;
          upd = true;
        }
        if (take_end) {
          win_end = s;
          stop = true;
        }
        flagged = flagged || ((take_min || take_end) && viol_s);
      }
      const bool swapped = q_ids[qs][2] != 0u;
      bool w_epa = false;
      if (win_min >= 0) {
        mind = run;
        fb1 = __shfl(int(prim), qs * SEG + win_min);
        jw = win_min;
        touched = true;
      }
      if (win_end >= 0) {
        const int src = qs * SEG + win_end;
        w_epa = __shfl(int(to_epa), src) != 0;
        ended = true;
        touched = true;
        ended_epa = w_epa;
        end_prim = __shfl(int(prim), src);
        if (!w_epa) end_val = __shfl(val, src);
        jw = win_end;
      }
      // the witness of the record: the ending triangle's when it carries a distance, else the minimum's (an EPA item falls back to it)
      const int w_lane = (win_end >= 0 && !w_epa) ? win_end : win_min;
      if (w_lane >= 0 && j == w_lane && (w_lane == win_end || win_min >= 0))
        store_witness(io, pair, swapped ? lo.p2 : lo.p1, swapped ? lo.p1 : lo.p2, swapped ? -lo.n : lo.n);
    }
    // ---- the windows written back, in order (a node behind a minimum of this pass that its bound does not beat, or behind a triangle
    // that ended the walk in this pass, is not split: the sequential walk skips it at its turn)
    const bool unsplit = split && (stop_at || (upd_at && db >= T(0) && db >= run_at));
    const int cnt = (split && !unsplit) ? 2 : (((is_leaf && !do_leaves) || held) ? 1 : 0);
    const uint64_t m2b = __ballot(cnt == 2), m1b = __ballot(cnt == 1);
    const int pos = base_i + 2 * __popcll(m2b & deeper) + __popcll(m1b & deeper);
    if (cnt == 2) {
      const bool c_first = d2 < d1;  // the nearer child is visited first
      // the children's chains: the parent's largest bound (its own, or what it inherited) where the child's own is below it
      const T pc = ch != 0.0f ? T(ch) : db;
      const float pcw = ch != 0.0f ? ch : chain_up(db);
      const float ca = d1 >= pc ? 0.0f : pcw, cc = d2 >= pc ? 0.0f : pcw;
      st_x[qs][pos] = c_first ? xa : xc;
      st_d[qs][pos] = c_first ? d1 : d2;
      st_c[qs][pos] = c_first ? ca : cc;
      st_x[qs][pos + 1] = c_first ? xc : xa;
      st_d[qs][pos + 1] = c_first ? d2 : d1;
      st_c[qs][pos + 1] = c_first ? cc : ca;
    } else if (cnt == 1) {
      st_x[qs][pos] = x;
      st_d[qs][pos] = db;
      st_c[qs][pos] = ch;
    }
    const uint64_t later_m = __ballot(act && (jw >= 0 ? j > jw : idx < p)) & segm;
    p = (jw >= 0 ? base_i : min(p, base_i)) + 2 * __popcll(m2b & later_m) + __popcll(m1b & later_m);
    sp = base_i + 2 * __popcll(m2b & segm) + __popcll(m1b & segm);
    if (sp > SDP_CAP - 2) {
      overflow = true;
      sp = 0;
    }
    // ordered mode: a triangle that ends the walk has everything that is left behind it; and a minimum that has come down to the pooled one
    // is the first the reference meets at the final distance -- nothing behind it replaces it
    if (ordered && (ended || mind <= pooled)) sp = 0;
    sync();
    if (active && sp == 0) {  // this walk is over
      if ((flagged || spill.rerun_all) && !ordered && !overflow && spill.rerun_count) {
        // not written: the slot takes the record again and walks it in the reference's order.  What the pooled pass found bounds that walk:
        // no entry whose bound exceeds it by a multiple of the arithmetic's slack can hold the reference's triangle
        pooled = ended ? T(0) : mind;
        cut = pooled + T(64) * margin;
        redo = true;
        if (j == 0) atomicAdd(spill.rerun_count, 1u);
      } else if (j == 0) {
        const bool epa_item = ended && ended_epa && !overflow;
        // the ordered walk has kept the minimum its record came with: that triangle's witness once more (the pooled pass wrote its own)
        const bool witness_again = ordered && !touched && !overflow && fb1 >= 0;
        if (epa_item || witness_again) {
          // the triangle that ended the walk: its leaf once more, with the EPA item (k_bvh_shape_finish writes the record)
          const uint32_t* c = q_ids[qs];
          const bool swapped = c[2] != 0u;
          const uint32_t again = uint32_t(epa_item ? end_prim : fb1);
          SolidLeafIn<T> in{bv.verts + 3 * size_t(c[3]), bv.tris + 3 * size_t(c[4] + again), lib.shapes, lib.verts,
                            swapped ? io.tf2 : io.tf1, swapped ? io.tf1 : io.tf2, epa_item ? reinterpret_cast<ShapeDeferItem<T>*>(wk.shape_defer) : nullptr,
                            epa_item ? &wk.counts[CTR_SHAPE_DEFER] : nullptr, epa_item ? wk.shape_defer_cap : 0u, pair, c[1], again, 0xFFFFFFFFu, 0u, epa_item ? mind : T(0), epa_item ? fb1 : -1};
          SolidLeafOut<T> lo;
          const bool deferred = solid_leaf_call<T>(in, &q, leaf_ps, initial_guess<T>(io, q, pair), &lo);
          if (!epa_item && !deferred) store_witness(io, pair, swapped ? lo.p2 : lo.p1, swapped ? lo.p1 : lo.p2, swapped ? -lo.n : lo.n);
        }
        if (!epa_item) {
          // b1 = the triangle, b2 = NONE whatever the operand order (distance.cpp:84-88 swaps o1 / o2 only)
          const T dist = ended && !overflow ? end_val : mind;
          store_bvh_record_head(io, pair, dist, dist <= T(0) ? 0x80000000u : 0u, ended && !overflow ? end_prim : fb1, -1, overflow);
          write_guess<T>(io, pair, initial_guess<T>(io, q, pair), 0, 0);
        }
      }
      active = false;
      ordered = false;
    }
  }
}


#if HFCL_BVH_MESH_PART || HFCL_BVH_SOLID_PART
// ---------------------------------------------------------------------------------------
// Mesh x mesh collide() in rounds of three kernels (hfcl_dev.hpp: WalkRec): the walk without its leaves, the leaves without a walk, the
// replay of each query's events in the reference's order.
// ---------------------------------------------------------------------------------------
// the lanes of `active` each need `count` consecutive slots behind *counter: one atomic per wave, the lane's first slot back
__device__ __forceinline__ uint32_t wave_reserve(uint32_t* counter, uint32_t count, bool active) {
  const int lane = threadIdx.x & 63;
  uint32_t incl = active ? count : 0u;
#pragma unroll
  for (int d = 1; d < 64; d <<= 1) {
    const uint32_t o = uint32_t(__shfl_up(int(incl), d));
    if (lane >= d) incl += o;
  }
  const uint32_t total = uint32_t(__shfl(int(incl), 63));
  uint32_t base = 0;
  if (lane == 0 && total) base = atomicAdd(counter, total);
  base = uint32_t(__builtin_amdgcn_readfirstlane(int(base)));
  return base + incl - (active ? count : 0u);
}
// the distance k_shape_leaves files for a leaf that needs EPA (no leaf is that far inside anything)
template <typename T> __device__ __forceinline__ T walk_leaf_epa() { return -Lim<T>::max(); }
#ifndef HFCL_WPE_BVH_WALK
#define HFCL_WPE_BVH_WALK 2
#endif
#ifndef HFCL_TRI_LEAVES_INLINE
#define HFCL_TRI_LEAVES_INLINE 1
#endif
#ifndef HFCL_WALK_NODE_CACHE
#define HFCL_WALK_NODE_CACHE 1
#endif
// k_bvh_walk: collisionRecurse (src/traversal/traversal_recurse.cpp:44-85) with the triangle pairs LISTED instead of tested, one query
// per lane, lanes refilled from a ticket as in k_bvh_collide.  A disjoint box pair only ever enters the walk's state through a
// minimum (updateDistanceLowerBoundFromBV), so the boxes between two leaves leave one number: the smallest bound among them.
// SOLID (mesh x solid, its own instantiation in its own part of this unit): one tree -- the entry is the mesh node, the other box is the
// solid's (k_shape_obb's ObbQuery, kept where the mesh x mesh walk keeps the relative pose), a leaf is a triangle against the solid.
template <typename T, bool SOLID = false>
__global__ void __launch_bounds__(BVH_BLOCK) __attribute__((amdgpu_waves_per_eu(HFCL_WPE_BVH_WALK, 8)))
k_bvh_walk(Work wk, LibView<T> lib, BvhView<T> bv, IO<T> io, QParams<T> q, T break_distance2, WalkArgs wa) {
  typedef BvhEntry<false> EN;
  constexpr int STACK = BVH_STACK_WALK;
  __shared__ uint32_t stack[STACK][BVH_BLOCK];
  WalkRec<T>* const recs = reinterpret_cast<WalkRec<T>*>(wa.recs);
  uint32_t* const c = wa.ctr + 8 * wa.round;
  const uint32_t cnt = wa.round ? wa.ctr[8 * (wa.round - 1) + 2] : wk.counts[SOLID ? B_BVHSHAPE : B_BVH];
  if (cnt == 0u) return;  // (a batch without mesh x mesh pairs: not one ticket drawn -- 1 500 same-address atomics are 30 us)
  const int tid = threadIdx.x, lane = tid & 63;
  V3<T> q_ext = mk<T>(T(0), T(0), T(0));  // SOLID: RT_R / RT_T hold the ObbQuery's M / V, q_ext its extent
  const T big = Lim<T>::max(), nanv = Lim<T>::nan();
  bool live = false, pending = false, exhausted = false;  // exhausted is wave-uniform
  uint32_t ri = 0, steps = 0, n_leaf = 0, flags = 0, noff1 = 0, noff2 = 0;
  int sp = 0;
  M3<T> RT_R;
  V3<T> RT_T = mk<T>(T(0), T(0), T(0));
  RT_R.r0 = RT_R.r1 = RT_R.r2 = RT_T;
  T pre_cur = big;
#if HFCL_WALK_NODE_CACHE
  DNode<T> n1, n2;
  n1.first_child = n2.first_child = 0;
  n1.axes = n2.axes = RT_R;
  n1.To = n2.To = n1.extent = n2.extent = RT_T;
  uint32_t id1 = 0xFFFFFFFFu, id2 = 0xFFFFFFFFu;
#endif
  for (;;) {
    if (live && (sp == 0 || n_leaf >= wa.k || flags != 0u)) {  // this round is over for the lane's query
      if (sp == 0) flags |= WALK_OVER;
      live = false;
      pending = true;
    }
    const uint64_t live_mask = __ballot(live);
    const int n_live = __popcll(live_mask);
    if (exhausted ? n_live == 0 : 64 - n_live >= BVH_REFILL_MIN) {
      // ---- the parked queries' records and their items (one reservation per wave), then new queries from the ticket
      if (__ballot(pending)) {
        WalkRec<T>* const r = recs + ri;
        const uint32_t first = wave_reserve(&c[1], n_leaf, pending);
        if (pending) {
          const bool fits = first + n_leaf <= wa.item_cap;  // (the host sizes the list for WALK_K items per query: always)
          r->sp = uint32_t(sp);
          r->n_leaf = fits ? n_leaf : 0u;
          r->flags = fits ? flags : (flags | WALK_LOST);
          r->first_item = first;
          r->pre[n_leaf] = pre_cur;
          for (int k = 0; k < sp; ++k) r->stack[k] = stack[k][tid];
          if (fits)
            for (uint32_t s2 = 0; s2 < n_leaf; ++s2) wa.items[first + s2] = ri | (s2 << 28);
          pending = false;
        }
      }
      if (exhausted) break;
      const int n_need = 64 - n_live;
      uint32_t base = 0;
      if (lane == 0) base = atomicAdd(&c[0], uint32_t(n_need));
      base = uint32_t(__builtin_amdgcn_readfirstlane(int(base)));
      if (!live) {
        const uint32_t it = base + uint32_t(__popcll(~live_mask & ((uint64_t(1) << lane) - 1)));
        if (it < cnt) {
          uint32_t pair;
          if (wa.round) {
            ri = wa.list_in[it];
            const WalkRec<T>* const r = recs + ri;
            pair = r->pair;
            sp = int(r->sp);
            for (int k = 0; k < sp; ++k) stack[k][tid] = r->stack[k];
          } else {
            ri = it;
            pair = wk.lists[size_t(SOLID ? B_BVHSHAPE : B_BVH) * wk.n + it];
            WalkRec<T>* const r = recs + ri;
            r->pair = pair;
            r->dlb = r->rec_dist = r->cand_val = big;
            store_witness(io, pair, mk<T>(nanv, nanv, nanv), mk<T>(nanv, nanv, nanv), mk<T>(nanv, nanv, nanv));
            stack[0][tid] = 0u;  // (b1 = 0, b2 = 0)
            sp = 1;
          }
          bool valid = true;
          if constexpr (SOLID) {
            const uint32_t sid1 = wk.shape1[pair], sid2 = wk.shape2[pair];
            const bool swapped = lib.kinds[sid1] != uint8_t(K_BVH);  // (shape, BVH): collide(o2, o1) then swapObjects (collision.cpp:93-108)
            noff1 = bv.meshes[lib.shapes[swapped ? sid2 : sid1].bvh_index].node_off;
            const ObbQuery<T> oq = reinterpret_cast<const ObbQuery<T>*>(wk.shape_oq)[pair];
            if (!(oq.ext.x == oq.ext.x)) {  // k_shape_obb: no OBB for this solid (k_bvh_collide's SOLID form flags the pair the same way)
              valid = false;
              store_unsupported(io, pair);
              WalkRec<T>* const r = recs + ri;
              r->sp = 0u;
              r->n_leaf = 0u;
              r->flags = WALK_VOID;
              r->first_item = 0u;
              sp = 0;
            }
            RT_R = oq.M;
            RT_T = oq.V;
            q_ext = oq.ext;
          } else {
          noff1 = bv.meshes[lib.shapes[wk.shape1[pair]].bvh_index].node_off;
          noff2 = bv.meshes[lib.shapes[wk.shape2[pair]].bvh_index].node_off;
          const Pose<T> tf1 = load_pose(io.tf1, pair), tf2 = load_pose(io.tf2, pair);
          RT_R = tmul(tf1.R, tf2.R);  // traversal_node_setup.h:560-563
          RT_T = tmul(tf1.R, tf2.t - tf1.t);
          }
          steps = 0;
          n_leaf = 0;
          flags = 0;
          pre_cur = big;
#if HFCL_WALK_NODE_CACHE
          id1 = id2 = 0xFFFFFFFFu;
#endif
          live = valid;
        }
      }
      if (base + uint32_t(n_need) >= cnt) exhausted = true;
      continue;
    }
    for (;;) {
      const bool can = live && sp > 0 && n_leaf < wa.k && flags == 0u;
      const uint64_t can_mask = __ballot(can);
      if (!can_mask) break;
      if (!exhausted && 64 - __popcll(can_mask) >= BVH_REFILL_MIN) break;
      if (!can) continue;
      if (wa.budget && steps >= wa.budget) {
        flags = WALK_BUDGET;
        continue;
      }
      ++steps;
      const uint32_t e = stack[--sp][tid];
      if constexpr (SOLID) {
        const DNode<T>* const np = bv.nodes + noff1 + e;
        const int32_t fc = np->first_child;
        if (fc < 0) {
          WalkRec<T>* const r = recs + ri;
          r->leaf1[n_leaf] = uint32_t(-(fc + 1));
          r->pre[n_leaf] = pre_cur;
          pre_cur = big;
          ++n_leaf;
          continue;
        }
        const DNode<T> nd1 = *np;
#if HFCL_BVH_PREFETCH
        const int32_t touched = np[fc - int32_t(e)].first_child;  // the node popped next if the boxes overlap
#endif
        ObbQuery<T> oq;
        oq.M = RT_R;
        oq.V = RT_T;
        oq.ext = q_ext;
        T sq;
        const bool disjoint = obb_disjoint_q(oq, nd1, q.security_margin, break_distance2, sq);
#if HFCL_BVH_PREFETCH
        asm volatile("" ::"v"(touched));
#endif
        if (disjoint) {
          const T nd = hsqrt(sq);
          if (nd < pre_cur) pre_cur = nd;
        } else if (sp + 2 > STACK) {
          stack[sp++][tid] = e;  // (a tree deeper than the stack: the rest of the walk is k_bvh_shape_coop's)
          flags = WALK_BUDGET;
        } else {
          stack[sp++][tid] = uint32_t(fc) + 1u;  // second child below
          stack[sp++][tid] = uint32_t(fc);       // first child on top
        }
        continue;
      }
      const uint32_t b1 = EN::first(e), b2 = EN::second(e);
#if HFCL_WALK_NODE_CACHE
      // the pair popped behind a box test shares a node with the pair tested (its child pair or its sibling): that record stays in registers
      if (b1 != id1) {
        n1 = bv.nodes[noff1 + b1];
        id1 = b1;
      }
      if (b2 != id2) {
        n2 = bv.nodes[noff2 + b2];
        id2 = b2;
      }
#else
      const DNode<T> n1 = bv.nodes[noff1 + b1];
      const DNode<T> n2 = bv.nodes[noff2 + b2];
#endif
      const bool l1 = n1.first_child < 0, l2 = n2.first_child < 0;
      if (l1 && l2) {
        WalkRec<T>* const r = recs + ri;
        r->leaf1[n_leaf] = uint32_t(-(n1.first_child + 1));
        r->leaf2[n_leaf] = uint32_t(-(n2.first_child + 1));
        r->pre[n_leaf] = pre_cur;
        pre_cur = big;
        ++n_leaf;
        continue;
      }
      const T sz1 = sqnorm(n1.extent), sz2 = sqnorm(n2.extent);
      const bool first = l2 || (!l1 && (sz1 > sz2));  // firstOverSecond
#if HFCL_BVH_PREFETCH
      const DNode<T>* const next = bv.nodes + (first ? noff1 + uint32_t(n1.first_child) : noff2 + uint32_t(n2.first_child));
      const int32_t touched = next[0].first_child;  // (k_bvh_collide: the record of the pair popped next, on its way up the caches)
#endif
      T sq;
      // argument order of the reference: overlap(RT.R, RT.T, model2.bv(b2), model1.bv(b1))
      const bool disjoint = obb_disjoint(RT_R, RT_T, n2, n1, q.security_margin, break_distance2, sq);
#if HFCL_BVH_PREFETCH
      asm volatile("" ::"v"(touched));
#endif
      if (disjoint) {
        const T nd = hsqrt(sq);
        if (nd < pre_cur) pre_cur = nd;
      } else if (sp + 2 > STACK) {
        stack[sp++][tid] = e;  // no room for its children: the pair goes back, the rest of the walk is k_bvh_coop's
        flags = WALK_BUDGET;
      } else {
        uint32_t ea, eb;
        if (first) {
          const uint32_t c1 = uint32_t(n1.first_child);
          ea = EN::pack(c1, b2);
          eb = EN::pack(c1 + 1, b2);
        } else {
          const uint32_t c1 = uint32_t(n2.first_child);
          ea = EN::pack(b1, c1);
          eb = EN::pack(b1, c1 + 1);
        }
        stack[sp++][tid] = eb;  // second child below
        stack[sp++][tid] = ea;  // first child on top
      }
    }
  }
}

// k_tri_leaves: every triangle pair the round's walks listed, one per lane (leafCollides' distance, traversal_node_bvhs.h:184-233,
// through the same tri_leaf_call as k_bvh_coop: one machine code for the leaf, whoever asks).
#if HFCL_BVH_SOLID_PART
// k_shape_leaves: the same for mesh x solid -- every triangle the round's walks listed against its solid, one per lane, through the
// solid_leaf_call of k_bvh_collide's SOLID form (internal/traversal_node_bvh_shape.h:139-188).  A leaf whose GJK ends inside the solid
// (it needs EPA: a contact whatever EPA finds, mesh_shape_lane_request) is marked by its distance (WALK_LEAF_EPA) -- nothing else of it is
// read --; redo = 1: the leaves k_bvh_resolve found to END their walks that way (wa.redo, their number at ctr[5]) once more, this time with
// the EPA item queued for k_bvh_shape_finish (as k_bvh_shape_coop does with the leaf that ends a walk of its own).
// The listed leaves ordered by the kind of their solid (WalkArgs::perm: position -> item), so that the 64 leaves of a wave of k_shape_leaves run ONE
// support function: in the order the walks listed them a wave held all six kinds of cfg4s and ran their GJK loops one after the other (953 waves:
// 440 us; profiles/r06_g).  Two launches: the kinds and their histogram (hist[16], zeroed by the host), then the scatter (cursor = hist + 16).
template <typename T>
__device__ __forceinline__ uint32_t item_kind(const Work& wk, const LibView<T>& lib, const WalkRec<T>* recs, uint32_t code) {
  const uint32_t pair = recs[code & 0x0FFFFFFFu].pair;
  const uint32_t id1 = wk.shape1[pair], id2 = wk.shape2[pair];
  const uint32_t k1 = lib.kinds[id1];
  return (k1 != uint32_t(K_BVH) ? k1 : uint32_t(lib.kinds[id2])) & 15u;
}
template <typename T>
__global__ void __launch_bounds__(256) k_item_kinds(Work wk, LibView<T> lib, WalkArgs wa, int redo) {
  __shared__ uint32_t h[16];
  const WalkRec<T>* const recs = reinterpret_cast<const WalkRec<T>*>(wa.recs);
  const uint32_t n_items = redo ? min(wa.ctr[8 * wa.round + 5], wa.list_stride) : min(wa.ctr[8 * wa.round + 1], wa.item_cap);
  uint32_t* const hist = wa.hist + (redo ? 32 : 0);
  if (threadIdx.x < 16) h[threadIdx.x] = 0u;
  __syncthreads();
  for (uint32_t it = blockIdx.x * 256u + threadIdx.x; it < n_items; it += gridDim.x * 256u) atomicAdd(&h[item_kind(wk, lib, recs, redo ? wa.redo[it] : wa.items[it])], 1u);
  __syncthreads();
  if (threadIdx.x < 16 && h[threadIdx.x]) atomicAdd(&hist[threadIdx.x], h[threadIdx.x]);
}
template <typename T>
__global__ void __launch_bounds__(256) k_item_scatter(Work wk, LibView<T> lib, WalkArgs wa, int redo) {
  __shared__ uint32_t h[16], base[16];
  const WalkRec<T>* const recs = reinterpret_cast<const WalkRec<T>*>(wa.recs);
  const uint32_t n_items = redo ? min(wa.ctr[8 * wa.round + 5], wa.list_stride) : min(wa.ctr[8 * wa.round + 1], wa.item_cap);
  uint32_t* const hist = wa.hist + (redo ? 32 : 0);
  uint32_t* const cursor = hist + 16;
  uint32_t* const perm = redo ? wa.perm + wa.item_cap : wa.perm;
  for (uint32_t it0 = blockIdx.x * 256u; it0 < n_items; it0 += gridDim.x * 256u) {
    if (threadIdx.x < 16) h[threadIdx.x] = 0u;
    __syncthreads();
    const uint32_t it = it0 + threadIdx.x;
    uint32_t kind = 0, rank = 0;
    if (it < n_items) {
      kind = item_kind(wk, lib, recs, redo ? wa.redo[it] : wa.items[it]);
      rank = atomicAdd(&h[kind], 1u);
    }
    __syncthreads();
    if (threadIdx.x < 16) {
      uint32_t start = 0;
      for (uint32_t k = 0; k < threadIdx.x; ++k) start += hist[k];
      base[threadIdx.x] = start + (h[threadIdx.x] ? atomicAdd(&cursor[threadIdx.x], h[threadIdx.x]) : 0u);
    }
    __syncthreads();
    if (it < n_items) perm[base[kind] + rank] = it;
    __syncthreads();
  }
}
template <typename T, bool REDO>
__global__ void __launch_bounds__(64) __attribute__((amdgpu_waves_per_eu(2, 8)))
k_shape_leaves(Work wk, LibView<T> lib, BvhView<T> bv, IO<T> io, QParams<T> q, WalkArgs wa) {
  __shared__ T w0_slab[W0Lds<T, 64>::WORDS];
  const W0Lds<T, 64> leaf_ps{w0_slab + threadIdx.x};
  const WalkRec<T>* const recs = reinterpret_cast<const WalkRec<T>*>(wa.recs);
  TriLeafOut<T>* const res = reinterpret_cast<TriLeafOut<T>*>(wa.res);
  const uint32_t n_items = REDO ? min(wa.ctr[8 * wa.round + 5], wa.list_stride) : min(wa.ctr[8 * wa.round + 1], wa.item_cap);
  for (uint32_t base = blockIdx.x * 64u; base < n_items; base += gridDim.x * 64u) {
    if (base + threadIdx.x < n_items) {
      const uint32_t it = wa.perm ? (REDO ? wa.perm + wa.item_cap : wa.perm)[base + threadIdx.x] : base + threadIdx.x;
      const uint32_t item_code = REDO ? wa.redo[it] : wa.items[it];
      const WalkRec<T>* const r = recs + (item_code & 0x0FFFFFFFu);
      const uint32_t s2 = item_code >> 28, pair = r->pair, prim = r->leaf1[s2];
      const uint32_t id1 = wk.shape1[pair], id2 = wk.shape2[pair];
      const bool swapped = lib.kinds[id1] != uint8_t(K_BVH);
      const uint32_t solid_id = swapped ? id1 : id2;
      const DMesh m1 = bv.meshes[lib.shapes[swapped ? id2 : id1].bvh_index];
      // the leaf INLINE (the same mesh_shape_leaf_lane as solid_leaf_call's: here no walk shares the lane's registers with it, and a call's
      // argument block and saved registers were 320 B of scratch per lane -- 230 MB written per 100k cfg4s queries)
      const uint32_t* const t3 = bv.tris + 3 * size_t(m1.tri_off + prim);
      const T* const mv = bv.verts + 3 * size_t(m1.vert_off);
      auto vtx = [&](uint32_t i) { return mk<T>(mv[3 * size_t(i)], mv[3 * size_t(i) + 1], mv[3 * size_t(i) + 2]); };
      const V3<T> ta = vtx(t3[0]), tb = vtx(t3[1]), tc = vtx(t3[2]);
      LaneSolid<T> solid;
      solid.s = lib.shapes[solid_id];
      solid.v = lib.verts + 3 * size_t(solid.s.vertex_offset);
      auto tfm_of = [&]() { return load_pose(swapped ? io.tf2 : io.tf1, pair); };
      auto tfs_of = [&]() { return load_pose(swapped ? io.tf1 : io.tf2, pair); };
      const MDiff<T> sMt = make_mdiff(tfs_of(), tfm_of());  // Transform3f::inverseTimes: the mesh frame in the solid's frame
      V3<T> guess = initial_guess<T>(io, q, pair);  // (split walks start every leaf from the request's guess)
      ShapeDeferItem<T> item;
      TriLeafOut<T> tlo;
      const bool to_epa = mesh_shape_leaf_lane(ta, tb, tc, sMt, tfm_of, tfs_of, solid.s, solid, swept_radius(solid.s), q, guess, leaf_ps, tlo.distance, tlo.p1, tlo.p2, tlo.n, item);
      if constexpr (REDO) {
        if (to_epa) {  // (always: the resolve kernel listed the leaf because this run's twin said so)
          item.seed.pair = pair;
          item.prim = prim;
          item.parent = 0xFFFFFFFFu;
          item.order = 0u;
          item.bound = T(0);
          item.prev_prim = -1;
          const uint32_t slot = atomicAdd(&wk.counts[CTR_SHAPE_DEFER], 1u);
          if (slot < wk.shape_defer_cap) reinterpret_cast<ShapeDeferItem<T>*>(wk.shape_defer)[slot] = item;
        }
      } else {
        if (to_epa) tlo.distance = walk_leaf_epa<T>();
        res[it] = tlo;
      }
    }
  }
}
#endif
#if HFCL_BVH_MESH_PART
template <typename T>
__global__ void __launch_bounds__(64) __attribute__((amdgpu_waves_per_eu(2, 8)))
k_tri_leaves(Work wk, LibView<T> lib, BvhView<T> bv, IO<T> io, QParams<T> q, WalkArgs wa) {
  __shared__ T w0_slab[W0Lds<T, 64>::WORDS];
  const W0Lds<T, 64> leaf_ps{w0_slab + threadIdx.x};
  const WalkRec<T>* const recs = reinterpret_cast<const WalkRec<T>*>(wa.recs);
  TriLeafOut<T>* const res = reinterpret_cast<TriLeafOut<T>*>(wa.res);
  const uint32_t n_items = min(wa.ctr[8 * wa.round + 1], wa.item_cap);
  for (uint32_t base = blockIdx.x * 64u; base < n_items; base += gridDim.x * 64u) {
    const uint32_t it = base + threadIdx.x;
    if (it < n_items) {
      const uint32_t item = wa.items[it];
      const WalkRec<T>* const r = recs + (item & 0x0FFFFFFFu);
      const uint32_t s2 = item >> 28, pair = r->pair;
      const DMesh m1 = bv.meshes[lib.shapes[wk.shape1[pair]].bvh_index], m2 = bv.meshes[lib.shapes[wk.shape2[pair]].bvh_index];
      TriLeafOut<T> tlo;
#if HFCL_TRI_LEAVES_INLINE
      {  // the leaf inline (k_bvh_collide's form of it): no walk shares the lane's registers here, and the call cost 192 B of scratch per lane
        const T* const v1 = bv.verts + 3 * size_t(m1.vert_off);
        const T* const v2 = bv.verts + 3 * size_t(m2.vert_off);
        const uint32_t* const t1 = bv.tris + 3 * size_t(m1.tri_off + r->leaf1[s2]);
        const uint32_t* const t2 = bv.tris + 3 * size_t(m2.tri_off + r->leaf2[s2]);
        auto vtx = [](const T* v, uint32_t i) { return mk<T>(v[3 * size_t(i)], v[3 * size_t(i) + 1], v[3 * size_t(i) + 2]); };
        TriSupport<T> tri;
        {
          const Pose<T> tf1 = load_pose(io.tf1, pair);
          tri.p1 = xform(tf1, vtx(v1, t1[0]));
          tri.p2 = xform(tf1, vtx(v1, t1[1]));
          tri.p3 = xform(tf1, vtx(v1, t1[2]));
        }
        {
          const Pose<T> tf2 = load_pose(io.tf2, pair);
          tri.q1 = xform(tf2, vtx(v2, t2[0]));
          tri.q2 = xform(tf2, vtx(v2, t2[1]));
          tri.q3 = xform(tf2, vtx(v2, t2[2]));
        }
        int gst, git;
        tlo.distance = tri_tri_distance(tri, q.gjk, q.guess_mode == HFCL_GUESS_CACHED, mk<T>(q.guess[0], q.guess[1], q.guess[2]), tlo.p1, tlo.p2, tlo.n, gst, git,
                                        (V3<T>*)nullptr, leaf_ps);
      }
#else
      tri_leaf_call<T>(bv.verts + 3 * size_t(m1.vert_off), bv.tris + 3 * size_t(m1.tri_off + r->leaf1[s2]), bv.verts + 3 * size_t(m2.vert_off),
                       bv.tris + 3 * size_t(m2.tri_off + r->leaf2[s2]), io.tf1, io.tf2, pair, &q, leaf_ps, &tlo);
#endif
      res[it] = tlo;
    }
  }
}
#endif

// The suspended queries as round `round` left them (the launch of k_bvh_coop that runs beside the later rounds continues those this round
// added), and the order its waves draw them in: by the stack entries a query still holds, most first (BvhSplit::order; counting sort by
// one block).  10 000 walks of 80 us on 2 048 wave slots are 0.4 ms of work; drawn as they were suspended the launch took 0.85.
template <typename T>
__global__ void __launch_bounds__(1024) k_walk_order(BvhSplit split, uint32_t round) {
  __shared__ uint32_t hist[64], base[64];
  const uint32_t hi = min(split.ctr[BVH_CTR_SUSPENDED], split.n_queries);
  const uint32_t lo = round ? min(split.walk.ctr[8u * (round - 1u) + WALK_CTR_SNAP], hi) : 0u;
  if (!split.order) {  // (option bvh_walk_order = 0: the snapshot alone)
    if (threadIdx.x == 0) split.walk.ctr[8u * round + WALK_CTR_SNAP] = hi;
    return;
  }
  if (threadIdx.x < 64) hist[threadIdx.x] = 0u;
  __syncthreads();
  auto key = [&](uint32_t slot) -> uint32_t { return 63u - min(bvh_sum<T>(split, slot)->n_child, 63u); };  // (most entries: bucket 0)
  for (uint32_t i = lo + threadIdx.x; i < hi; i += blockDim.x) atomicAdd(&hist[key(i)], 1u);
  __syncthreads();
  if (threadIdx.x == 0) {
    uint32_t run = lo;
    for (int b = 0; b < 64; ++b) {
      base[b] = run;
      run += hist[b];
    }
    split.walk.ctr[8u * round + WALK_CTR_SNAP] = hi;
  }
  __syncthreads();
  for (uint32_t i = lo + threadIdx.x; i < hi; i += blockDim.x) split.order[atomicAdd(&base[key(i)], 1u)] = i;
}

// k_bvh_resolve: a query's events of this round in the reference's order -- the boxes in front of a leaf (their smallest bound:
// updateDistanceLowerBoundFromBV), the leaf (updateDistanceLowerBoundFromLeaf, the witness of the last one that lowered the bound, the
// contact that ends the walk: whatever the walk listed behind it is void) -- and where the query goes from here: its record, the next
// round, or k_bvh_coop (a suspended query with its stack as tasks, exactly what k_bvh_collide's suspension leaves).
// SOLID: the leaf is a triangle against the solid (fb1 / fb2 and the witness in the caller's operand order: collision.cpp:93-108), a leaf that
// needs EPA ends the walk as a contact whose numbers k_bvh_shape_finish writes: it goes on the redo list (k_shape_leaves, redo = 1).
template <typename T, bool SOLID = false>
__global__ void __launch_bounds__(256) k_bvh_resolve(Work wk, LibView<T> lib, IO<T> io, QParams<T> q, BvhSplit split, WalkArgs wa) {
  WalkRec<T>* const recs = reinterpret_cast<WalkRec<T>*>(wa.recs);
  const TriLeafOut<T>* const res = reinterpret_cast<const TriLeafOut<T>*>(wa.res);
  uint32_t* const c = wa.ctr + 8 * wa.round;
  const uint32_t cnt = wa.round ? wa.ctr[8 * (wa.round - 1) + 2] : wk.counts[SOLID ? B_BVHSHAPE : B_BVH];
  for (uint32_t base = blockIdx.x * blockDim.x; base < cnt; base += gridDim.x * blockDim.x) {
    const uint32_t it = base + threadIdx.x;
    const bool valid = it < cnt;
    uint32_t ri = 0, pair = 0, sp = 0, redo_item = 0;
    bool next = false, coop = false, lost = false, redo = false;
    WalkRec<T>* r = recs;
    T dlb = T(0), rec_dist = T(0), cand_val = T(0);
    if (valid && !(SOLID && (recs[wa.round ? wa.list_in[it] : it].flags & WALK_VOID))) {
      ri = wa.round ? wa.list_in[it] : it;
      r = recs + ri;
      pair = r->pair;
      sp = r->sp;
      const uint32_t n_leaf = r->n_leaf, flags = r->flags;
      bool swapped = false;
      if constexpr (SOLID) swapped = lib.kinds[wk.shape1[pair]] != uint8_t(K_BVH);
      dlb = r->dlb;
      rec_dist = r->rec_dist;
      cand_val = r->cand_val;
      int fb1 = -1, fb2 = -1, wit = -1;
      bool contact = false;
      auto boxes = [&](T pre) {  // updateDistanceLowerBoundFromBV over a run of disjoint boxes
        if (!(dlb <= T(0)) && pre < dlb) {
          dlb = pre;
          rec_dist = pre + q.security_margin;
        }
      };
      for (uint32_t s2 = 0; s2 < n_leaf; ++s2) {
        boxes(r->pre[s2]);
        const T distance = res[r->first_item + s2].distance;
        if (SOLID && distance == walk_leaf_epa<T>()) {  // canStop(): the bound and its witness stay as they are (k_bvh_collide's SOLID form)
          contact = true;
          redo = true;
          redo_item = ri | (s2 << 28);
          fb1 = swapped ? -1 : int(r->leaf1[s2]);
          fb2 = swapped ? int(r->leaf1[s2]) : -1;
          break;
        }
        const T dtc = distance - q.security_margin;
        if (dtc < dlb) {  // updateDistanceLowerBoundFromLeaf
          dlb = dtc;
          cand_val = dtc;
          rec_dist = distance;
          wit = int(s2);
        }
        if (dtc <= q.collision_distance_threshold) {  // the first contact: canStop()
          contact = true;
          fb1 = SOLID ? (swapped ? -1 : int(r->leaf1[s2])) : int(r->leaf1[s2]);
          fb2 = SOLID ? (swapped ? int(r->leaf1[s2]) : -1) : int(r->leaf2[s2]);
          break;
        }
      }
      if (!contact) boxes(r->pre[n_leaf]);
      if (wit >= 0) {
        const TriLeafOut<T> w = res[r->first_item + uint32_t(wit)];
        if (SOLID && swapped)
          store_witness(io, pair, w.p2, w.p1, -w.n);
        else
          store_witness(io, pair, w.p1, w.p2, w.n);
      }
      lost = (flags & WALK_LOST) != 0u;  // (its items did not fit the list)
      if (contact || (flags & WALK_OVER) || lost) {
        store_bvh_record_head(io, pair, rec_dist, contact ? 1u : 0u, fb1, fb2, lost);
      } else if (!wa.last && !(flags & WALK_BUDGET)) {
        next = true;
      } else {
        coop = true;
      }
      if (next || coop) {
        r->dlb = dlb;
        r->rec_dist = rec_dist;
        r->cand_val = cand_val;
      }
    }
    // ---- the queries that walk on: next round's list
    {
      const uint32_t slot = wave_reserve(&c[2], 1u, next);
      if (next) wa.list_out[slot] = ri;
    }
    if constexpr (SOLID) {  // ---- the leaves that ended their walks needing EPA
      const uint32_t slot = wave_reserve(&c[5], 1u, redo);
      if (redo && slot < wa.list_stride) wa.redo[slot] = redo_item;
    }
    // ---- the queries k_bvh_coop continues: a suspended query (its state a summary) whose stack entries are its tasks, top first
    if (__ballot(coop)) {
      const uint32_t my_slot = wave_reserve(&split.ctr[BVH_CTR_SUSPENDED], 1u, coop);
      const uint32_t first = wave_reserve(&split.ctr[BVH_CTR_TASKS], sp, coop);
      if (coop) {
        const bool fits = first + sp <= split.cap && my_slot < split.n_queries;
        if (my_slot < split.n_queries) split.suspended[my_slot] = pair;
        if (fits)
          for (uint32_t j = 0; j < sp; ++j) split.tasks[first + j] = BvhTask{pair, my_slot, r->stack[sp - 1u - j], j};
        else
          for (uint32_t j = first; j < min(first + sp, split.cap); ++j) split.tasks[j] = BvhTask{0u, 0u, 0xFFFFFFFFu, 0u};
        if (my_slot < split.n_queries) {
          BvhSum<T>* const sm = bvh_sum<T>(split, my_slot);
          sm->contact_order = 0xFFFFFFFFu; sm->parent = 0xFFFFFFFFu; sm->order = 0u; sm->pad_ = 0u;
          sm->dlb = dlb; sm->rec_dist = rec_dist; sm->cand_val = cand_val;
          sm->fb1 = sm->fb2 = -1;
          sm->ncontacts = 0u; sm->first_child = first; sm->n_child = fits ? sp : 0u;
          sm->flags = BVH_SUM_SUSPENDED | (fits ? 0u : BVH_SUM_OVERFLOW);  // (a full task table: flagged, as k_bvh_collide flags a stack it cannot hand on)
          load_witness(io, pair, sm->np1, sm->np2, sm->nn);
        }
      }
    }
  }
}
#endif

// =======================================================================================
// launchers (hfcl_launch.hpp)
// =======================================================================================
// One level per launch (levels > 0 walk the tasks the level before made), the fold-back launches in reverse order.
// split.tasks == nullptr (or split.n_levels <= 1): the plain single-pass traversal.
// one launch of the collide kernel in the form the batch asks for: WIDE (32-bit node ids) or not, with the fp32 filter in
// front of the fp64 test (fp64 batches of a library whose filter records are uploaded, bv.fnodes) or without
#if HFCL_BVH_COLLIDE_PART
template <typename T>
static void launch_collide_kernel(bool wide, int grid, hipStream_t st, const Work& wk, const LibView<T>& lv, const BvhView<T>& bv, const IO<T>& io, const QParams<T>& q, const BvhParams& bp, T break_distance2, const BvhSplit& split, const BvhSpill& spill, bool solid = false) {
#if HFCL_BVH_SOLID_PART
  if (solid) {  // mesh x solid: the narrow form, 32-bit node ids in the entry
    hipLaunchKernelGGL((k_bvh_collide<T, false, false, true>), dim3(grid), dim3(BVH_BLOCK), 0, st, wk, lv, bv, io, q, bp, break_distance2, split, spill);
    return;
  }
#endif
#if HFCL_BVH_MESH_PART
#if HFCL_KEEP_AB_FORMS
  if constexpr (sizeof(T) == 8) {
    if (bv.fnodes) {
      if (wide)
        hipLaunchKernelGGL((k_bvh_collide<T, true, true>), dim3(grid), dim3(BVH_BLOCK), 0, st, wk, lv, bv, io, q, bp, break_distance2, split, spill);
      else
        hipLaunchKernelGGL((k_bvh_collide<T, false, true>), dim3(grid), dim3(BVH_BLOCK), 0, st, wk, lv, bv, io, q, bp, break_distance2, split, spill);
      return;
    }
  }
#endif
  if (wide)
    hipLaunchKernelGGL((k_bvh_collide<T, true, false>), dim3(grid), dim3(BVH_BLOCK), 0, st, wk, lv, bv, io, q, bp, break_distance2, split, spill);
  else
    hipLaunchKernelGGL((k_bvh_collide<T, false, false>), dim3(grid), dim3(BVH_BLOCK), 0, st, wk, lv, bv, io, q, bp, break_distance2, split, spill);
#else
  (void)wide;
#endif
}
// The continuation of the suspended queries by k_bvh_coop / k_bvh_shape_coop: one launch, or -- BvhSplit::cut_ticks -- three, the second and
// third walking the chunks the launch before cut its long walks into (levels 2 and 3 of the task table: k_bvh_level_mark files the
// tasks made so far under ctr[LEVEL0 + level + 1] and resets the ticket), and the fold-back of the chunks' summaries.
// after_first (optional): called once the mark behind the first launch is in the stream -- returns false when there is no such mark (no cutting)
template <typename T, class Launch, class AfterFirst>
static bool launch_coop_levels(int grid, hipStream_t st, const Work& wk, const IO<T>& io, BvhSplit s, int ticket, Launch&& launch, AfterFirst&& after_first) {
  const bool cutting = s.cut_ticks != 0 && s.cut_words != nullptr;
  s.level = 0;
  for (uint32_t l = 0; l < (cutting ? 3u : 1u); ++l) {
    hipLaunchKernelGGL((k_bvh_level_mark<HFCL_BVH_PART>), dim3(1), dim3(64), 0, st, wk, s, ticket);
    if (l == 1) after_first();
    s.level = l ? l + 1 : 0u;  // launch 0: the suspended queries; launches 1, 2: the chunks of the launch before
    s.can_suspend = cutting && l < 2 ? 1u : 0u;
    launch(s);
    s.level = l + 1;  // (what the next mark files the tasks under)
  }
  if (cutting) {
    s.level = 2;
    hipLaunchKernelGGL((k_bvh_combine<T, HFCL_BVH_PART>), dim3(std::max(1, std::min(grid, 1024))), dim3(256), 0, st, wk, io, s);
    s.level = 0;
    hipLaunchKernelGGL((k_bvh_combine<T, HFCL_BVH_PART>), dim3(std::max(1, std::min(grid, 1024))), dim3(256), 0, st, wk, io, s);
  }
  return cutting;
}
template <typename T, class Launch>
static void launch_coop_levels(int grid, hipStream_t st, const Work& wk, const IO<T>& io, BvhSplit s, int ticket, Launch&& launch) {
  launch_coop_levels<T>(grid, st, wk, io, s, ticket, launch, [] {});
}
// (static: each part of this unit has its own -- the mesh x solid part walks its task levels with it, solid = true)
template <typename T>
static void bvh_collide_levels(int grid, hipStream_t st, const Work& wk, const LibView<T>& lv, const BvhView<T>& bv, const IO<T>& io, const QParams<T>& q, const BvhParams& bp, T break_distance2, BvhSplit split, BvhSpill spill, bool solid, const AsideStream* aside = nullptr) {
  if (spill.wide && !solid) {  // models with 32-bit node ids: single pass, global spill instead of tasks
    split.tasks = nullptr;
    split.budget = 0;
    split.level = 0;
    split.can_suspend = 0;
    if (spill.slab) grid = std::min(grid, int(spill.max_blocks));
    launch_collide_kernel<T>(true, grid, st, wk, lv, bv, io, q, bp, break_distance2, split, spill);
    return;
  }
  const bool splitting = split.tasks && split.n_levels > 1 && bp.num_max_contacts == 1 && !bp.contacts;
  if (!splitting) {
    split.tasks = nullptr;
    split.budget = 0;
    split.level = 0;
    split.can_suspend = 0;
    launch_collide_kernel<T>(false, grid, st, wk, lv, bv, io, q, bp, break_distance2, split, spill, solid);
    return;
  }
#if HFCL_BVH_MESH_PART
  if (split.coop && !solid) {
    // the queries for their step budget, one per lane; then the suspended ones, a lane group each (k_bvh_coop)
    BvhSplit s0 = split;
    s0.level = 0;
    s0.budget = split.budget0;
    s0.can_suspend = 1;
    uint32_t n_early = 0;  // helper streams with a continuation launch on them
    if (split.walk.recs && split.walk_rounds) {
      // the queries' own phase in rounds: walk (its leaves listed), leaves (one per lane), resolve (hfcl_dev.hpp: WalkRec)
      WalkArgs wa = split.walk;
      uint32_t* const lists = wa.list_in;
      const int lgrid = std::max(1, std::min(grid * 2, int(split.coop_grid ? split.coop_grid : 2048u)));
      // The queries a round hands over (step budget) are continued on a helper stream BESIDE the later rounds: at 100k queries neither
      // fills the chip (1 500 waves of lanes, then a few thousand waves of one query each), both are chains of dependent steps.
      const bool early = aside && split.walk_rounds > 1 && !(split.cut_ticks && split.cut_words);
      const int coop_grid_w = std::max(1, std::min(grid * BVH_BLOCK, split.coop_grid ? int(split.coop_grid) : grid * 2));
      for (uint32_t r = 0; r < split.walk_rounds; ++r) {
        if (early && r >= 1 && aside[r - 1].stream) {
          const AsideStream& as = aside[r - 1];
          hipLaunchKernelGGL((k_walk_order<T>), dim3(1), dim3(1024), 0, st, s0, r - 1u);
          hipEventRecord(as.fork, st);
          hipStreamWaitEvent(as.stream, as.fork, 0);
          BvhSplit s1 = s0;
          s1.coop_range = r;  // (1 + the round that just ended)
          s1.can_suspend = 0u;
          hipLaunchKernelGGL((k_bvh_coop<T>), dim3(coop_grid_w), dim3(64), 0, as.stream, wk, lv, bv, io, q, bp, break_distance2, s1);
          hipEventRecord(as.join, as.stream);
          s0.coop_range = 0x100u + (r - 1u);
          n_early = r;
        }
        wa.round = r;
        wa.k = std::min<uint32_t>(split.walk_k[r], uint32_t(WALK_K));
        wa.budget = split.walk_budget[r];
        wa.last = r + 1 == split.walk_rounds ? 1u : 0u;
        wa.list_out = lists + size_t(r & 1u) * wa.list_stride;
        wa.list_in = lists + size_t((r + 1u) & 1u) * wa.list_stride;  // (= the list_out of round r - 1; not read in round 0)
        hipLaunchKernelGGL((k_bvh_walk<T>), dim3(grid), dim3(BVH_BLOCK), 0, st, wk, lv, bv, io, q, break_distance2, wa);
        hipLaunchKernelGGL((k_tri_leaves<T>), dim3(lgrid), dim3(64), 0, st, wk, lv, bv, io, q, wa);
        hipLaunchKernelGGL((k_bvh_resolve<T>), dim3(std::max(1, std::min(grid, 1024))), dim3(256), 0, st, wk, lv, io, q, s0, wa);
      }
    } else {
      launch_collide_kernel<T>(false, grid, st, wk, lv, bv, io, q, bp, break_distance2, s0, spill, false);
    }
    // a wave per suspended query, up to what the chip holds (`grid` blocks of BVH_BLOCK queries: `grid * 2` waves left a quarter of the
    // wave slots empty at 100k queries, profiles/r04_h)
    const int coop_grid = std::max(1, std::min(grid * BVH_BLOCK, split.coop_grid ? int(split.coop_grid) : grid * 2));
    if (s0.order && split.walk.recs && split.walk_rounds && !(split.cut_ticks && split.cut_words)) {
      // (what the last round added -- or, without launches beside the rounds, every suspended query -- longest stack first)
      hipLaunchKernelGGL((k_walk_order<T>), dim3(1), dim3(1024), 0, st, s0, n_early ? split.walk_rounds - 1u : 0u);
    } else {
      s0.order = nullptr;
    }
    launch_coop_levels<T>(grid, st, wk, io, s0, int(B_COUNT + 2), [&](const BvhSplit& s) {
      hipLaunchKernelGGL((k_bvh_coop<T>), dim3(coop_grid), dim3(64), 0, st, wk, lv, bv, io, q, bp, break_distance2, s);
    });
    for (uint32_t r = 0; r < n_early; ++r) hipStreamWaitEvent(st, aside[r].join, 0);
    return;
  }
#endif
  const uint32_t budget = split.budget;
  for (uint32_t l = 0; l < split.n_levels; ++l) {
    split.level = l;
    split.budget = l + 1 < split.n_levels ? (l == 0 ? split.budget0 : budget) : 0u;  // the last level runs to the end
    BvhSplit s = split;
    s.can_suspend = l + 1 < split.n_levels;  // ... and cannot suspend (its stack overflows are flagged)
    launch_collide_kernel<T>(false, grid, st, wk, lv, bv, io, q, bp, break_distance2, s, spill, solid);
    hipLaunchKernelGGL((k_bvh_level_mark<HFCL_BVH_PART>), dim3(1), dim3(64), 0, st, wk, s, solid ? int(CTR_SHAPE_TICKET) : int(B_COUNT + 2));
  }
  for (int l = int(split.n_levels) - 2; l >= 0; --l) {
    split.level = uint32_t(l);
    hipLaunchKernelGGL((k_bvh_combine<T, HFCL_BVH_PART>), dim3(std::max(1, std::min(grid, 1024))), dim3(256), 0, st, wk, io, split);
  }
}
#endif
#if HFCL_BVH_MESH_PART
template <typename T>
void launch_bvh_collide(int grid, hipStream_t st, const Work& wk, const LibView<T>& lv, const BvhView<T>& bv, const IO<T>& io, const QParams<T>& q, const BvhParams& bp, T break_distance2, BvhSplit split, BvhSpill spill, const AsideStream* aside) {
  bvh_collide_levels<T>(grid, st, wk, lv, bv, io, q, bp, break_distance2, split, spill, false, aside);
}
#endif
#if HFCL_BVH_SOLID_PART
template <typename T>
void launch_bvh_shape(int grid, hipStream_t st, const Work& wk, const LibView<T>& lv, const BvhView<T>& bv, const IO<T>& io, const QParams<T>& q, const BvhParams& bp, T break_distance2) {
  hipLaunchKernelGGL((k_bvh_shape<T>), dim3(grid), dim3(64), 0, st, wk, lv, bv, io, q, bp, break_distance2);
}
#endif
#if HFCL_BVH_DISTANCE_PART
// mesh x solid distance(), one query per lane
template <typename T>
void launch_bvh_shape_distance_fast(int grid, int grid_finish, hipStream_t st, const Work& wk, const LibView<T>& lv, const BvhView<T>& bv, const IO<T>& io, const QParams<T>& q, BvhSpill spill) {
  hipLaunchKernelGGL((k_shape_obbrss<T>), dim3(std::max(1, grid / 4)), dim3(256), 0, st, wk, lv, io);
  hipLaunchKernelGGL((k_bvh_shape_distance_lane<T>), dim3(grid), dim3(BVHD_BLOCK), 0, st, wk, lv, bv, io, q, spill);
  const int n_est = int(std::min<uint32_t>(wk.n, 0x7FFFFFFFu));
  if (spill.budget && spill.pool)
    hipLaunchKernelGGL((k_bvh_shape_distance_pool<T>), dim3(std::max(1, std::min((n_est + SDP_Q - 1) / SDP_Q, int(spill.max_blocks)))), dim3(64), 0, st, wk, lv, bv, io, q, spill);
  else if (spill.budget)
    hipLaunchKernelGGL((k_bvh_shape_distance_coop<T>), dim3(std::max(1, std::min(n_est, int(spill.max_blocks)))), dim3(64), 0, st, wk, lv, bv, io, q, spill);
  BvhSplit none;
  memset(&none, 0, sizeof(none));
  BvhParams bp;
  memset(&bp, 0, sizeof(bp));
  launch_shape_finish<T>(grid_finish, st, wk, lv, io, q, bp, none, 1);
}
#endif
#if HFCL_BVH_SOLID_PART
// mesh x solid collide(), one query per lane: the solids' OBBs, the walk (split as `split` says), the EPA leaves
template <typename T>
void launch_bvh_shape_fast(int grid, int grid_finish, int coop_grid, hipStream_t st, const Work& wk, const LibView<T>& lv, const BvhView<T>& bv, const IO<T>& io, const QParams<T>& q, const BvhParams& bp, T break_distance2, BvhSplit split, BvhSpill spill, const AsideStream* aside) {
  hipLaunchKernelGGL((k_shape_obb<T>), dim3(std::max(1, grid / 2)), dim3(256), 0, st, wk, lv, io);
  spill.wide = 0;
  spill.slab = nullptr;
  const bool splitting = split.tasks && split.n_levels > 1 && bp.num_max_contacts == 1 && !bp.contacts;
  if (!splitting) split.tasks = nullptr;
  if (splitting && split.coop) {
    // the queries for their step budget, one per lane; then the suspended ones, one per wave and 64 entries at a time
    BvhSplit s0 = split;
    s0.level = 0;
    s0.budget = split.budget0;
    s0.can_suspend = 1;
    if (split.walk.recs && split.walk_rounds) {
      // the queries' own phase as walk (its leaves listed) / leaves (one per lane, dense) / resolve (hfcl_dev.hpp: WalkRec), as mesh x mesh;
      // then the leaves that ended a walk needing EPA once more, for their items
      WalkArgs wa = split.walk;
      wa.round = 0;
      wa.k = std::min<uint32_t>(split.walk_k[0], uint32_t(WALK_K));
      wa.budget = split.walk_budget[0];
      wa.last = 1u;
      wa.list_out = wa.list_in + wa.list_stride;
      const int lgrid = std::max(1, std::min(grid * 2, 2048));
      const int sgrid = std::max(1, std::min(grid, 512));
      hipLaunchKernelGGL((k_bvh_walk<T, true>), dim3(grid), dim3(BVH_BLOCK), 0, st, wk, lv, bv, io, q, break_distance2, wa);
      if (wa.perm) {
        hipLaunchKernelGGL((k_item_kinds<T>), dim3(sgrid), dim3(256), 0, st, wk, lv, wa, 0);
        hipLaunchKernelGGL((k_item_scatter<T>), dim3(sgrid), dim3(256), 0, st, wk, lv, wa, 0);
      }
      hipLaunchKernelGGL((k_shape_leaves<T, false>), dim3(lgrid), dim3(64), 0, st, wk, lv, bv, io, q, wa);
      hipLaunchKernelGGL((k_bvh_resolve<T, true>), dim3(std::max(1, std::min(grid, 1024))), dim3(256), 0, st, wk, lv, io, q, s0, wa);
      if (wa.perm) {
        hipLaunchKernelGGL((k_item_kinds<T>), dim3(std::max(1, sgrid / 8)), dim3(256), 0, st, wk, lv, wa, 1);
        hipLaunchKernelGGL((k_item_scatter<T>), dim3(std::max(1, sgrid / 8)), dim3(256), 0, st, wk, lv, wa, 1);
      }
      hipLaunchKernelGGL((k_shape_leaves<T, true>), dim3(lgrid), dim3(64), 0, st, wk, lv, bv, io, q, wa);
    } else {
      launch_collide_kernel<T>(false, grid, st, wk, lv, bv, io, q, bp, break_distance2, s0, spill, true);
    }
    // (the EPA item of a chunk that stands behind another chunk's contact is skipped: bvh_moot over the summaries; items of whole walks
    // hang under no summary)
    BvhSplit fin = s0;
    fin.level = 0;
    // The items of whole walks -- all that k_bvh_collide<SOLID> and the first launch of k_bvh_shape_coop queue -- are final when that launch
    // ends: their EPA runs on the helper stream beside the two launches that walk the chunks (few waves, as long as their longest chains:
    // 0.54 ms of cfg4s's 3.45, and k_bvh_shape_finish another 0.52 behind them before; profiles/r05_f).
    bool forked = false;
    const bool cut = launch_coop_levels<T>(grid, st, wk, io, s0, int(CTR_SHAPE_TICKET), [&](const BvhSplit& s) {
      hipLaunchKernelGGL((k_bvh_shape_coop<T>), dim3(coop_grid), dim3(64), 0, st, wk, lv, bv, io, q, bp, break_distance2, s);
    }, [&] {
      if (!aside || !wk.shape_finish_over) return;
      if (hipEventRecord(aside->fork, st) != hipSuccess || hipStreamWaitEvent(aside->stream, aside->fork, 0) != hipSuccess) return;
      launch_shape_finish<T>(grid_finish, aside->stream, wk, lv, io, q, bp, fin, 0, 1);
      forked = hipEventRecord(aside->join, aside->stream) == hipSuccess;
    });
    (void)cut;
    launch_shape_finish<T>(grid_finish, st, wk, lv, io, q, bp, fin, 0, forked ? 2 : 0);
    if (forked) hipStreamWaitEvent(st, aside->join, 0);
    return;
  }
  bvh_collide_levels<T>(grid, st, wk, lv, bv, io, q, bp, break_distance2, split, spill, true);
  split.level = 0;
  launch_shape_finish<T>(grid_finish, st, wk, lv, io, q, bp, split, 0);
}
#endif
#if HFCL_BVH_DISTANCE_PART
template <typename T>
void launch_bvh_shape_distance(int grid, hipStream_t st, const Work& wk, const LibView<T>& lv, const BvhView<T>& bv, const IO<T>& io, const QParams<T>& q) {
  hipLaunchKernelGGL((k_bvh_shape_distance<T>), dim3(grid), dim3(64), 0, st, wk, lv, bv, io, q);
}
#endif
#if HFCL_BVH_MESH_PART
template <typename T>
void launch_triangle(int grid, hipStream_t st, const Work& wk, const LibView<T>& lv, const IO<T>& io, const QParams<T>& q) {
  hipLaunchKernelGGL((k_triangle<T>), dim3(grid), dim3(64), 0, st, wk, lv, io, q);
}
#endif
#if HFCL_BVH_MESH_PART
#define HFCL_INST(T)                                                                                                             \
  template void launch_bvh_collide<T>(int, hipStream_t, const Work&, const LibView<T>&, const BvhView<T>&, const IO<T>&, const QParams<T>&, const BvhParams&, T, BvhSplit, BvhSpill, const AsideStream*); \
  template void launch_triangle<T>(int, hipStream_t, const Work&, const LibView<T>&, const IO<T>&, const QParams<T>&);
HFCL_INST(float)
HFCL_INST(double)
#undef HFCL_INST
#endif
#if HFCL_BVH_SOLID_PART
#define HFCL_INST(T)                                                                                                             \
  template void launch_bvh_shape<T>(int, hipStream_t, const Work&, const LibView<T>&, const BvhView<T>&, const IO<T>&, const QParams<T>&, const BvhParams&, T);   \
  template void launch_bvh_shape_fast<T>(int, int, int, hipStream_t, const Work&, const LibView<T>&, const BvhView<T>&, const IO<T>&, const QParams<T>&, const BvhParams&, T, BvhSplit, BvhSpill, const AsideStream*);
HFCL_INST(float)
HFCL_INST(double)
#undef HFCL_INST
#endif
#if HFCL_BVH_DISTANCE_PART
#define HFCL_INST(T)                                                                                                             \
  template void launch_bvh_shape_distance<T>(int, hipStream_t, const Work&, const LibView<T>&, const BvhView<T>&, const IO<T>&, const QParams<T>&);          \
  template void launch_bvh_shape_distance_fast<T>(int, int, hipStream_t, const Work&, const LibView<T>&, const BvhView<T>&, const IO<T>&, const QParams<T>&, BvhSpill);
HFCL_INST(float)
HFCL_INST(double)
#undef HFCL_INST
#endif
