// hfcl_epa.hpp -- Expanding Polytope Algorithm core for the batched MI355X narrow phase.
//
// Behavioural contract: hpp-fcl's details::EPA (/root/reference/src/narrowphase/gjk.cpp:
// reset :1014-1037, newFace :1068-1138, findClosestFace :1141-1154, evaluate :1156-1316,
// expand :1361-1449, getWitnessPointsAndNormal :1451-1466) and GJK::encloseOrigin (:437-492).
//
// Re-designed for the GPU:
//   * the polytope (<= 68 vertices, <= 132 faces) lives in a caller-provided scratch block
//     (LDS on the device), structure-of-arrays, 8-bit indices instead of pointers;
//   * the hull/stock doubly linked lists of the reference are replaced by an in-hull flag
//     plus an append stamp: the reference's list order (most recently appended first) only
//     matters as the tie-break of findClosestFace, which becomes a lane-parallel
//     arg-min over (d^2 ascending, stamp descending);
//   * the recursive expand() is split in two: a serial silhouette walk (explicit frame stack) that
//     only *lists* the horizon edges, then the new faces -- one per horizon edge, independent of
//     each other -- are built by the lanes of the group in parallel.  The reference creates them
//     inside the recursion; which of its failure exits fires first (stock exhausted, degenerate /
//     non-convex face, invalid hull) is reproduced from the walk order (see expand_iteration);
//   * encloseOrigin() is an explicit state machine.
// One polytope is driven by one lane group; all lanes of the group execute the scalar
// control flow redundantly (uniform LDS addresses broadcast); face scans, face creation and
// face removal are split over the lanes.
#pragma once
#include <type_traits>

#include "hfcl_gjk.hpp"

namespace hfcl {

enum {
  EPA_FAILED = 0, EPA_VALID = 1, EPA_ACCURACY_REACHED = 3, EPA_DEGENERATED = 2, EPA_NON_CONVEX = 4,
  EPA_INVALID_HULL = 6, EPA_OUT_OF_FACES = 8, EPA_OUT_OF_VERTICES = 10, EPA_FALLBACK = 12, EPA_DID_NOT_RUN = 15
};

constexpr int EPA_MAX_ITER = 64;                 // device limit (reference default, narrowphase_defaults.h:60)
constexpr int EPA_MAX_VERTS = EPA_MAX_ITER + 4;  // gjk.cpp:1020
constexpr int EPA_MAX_FACES = 2 * EPA_MAX_ITER + 4;  // gjk.cpp:1021
// The append stamp of a face is packed into 14 bits of its topology record (FaceTopo).  A polytope appends 4 faces at the
// start and, per iteration, at most as many as it has faces: the bound below holds for the reference's capacity; raising
// EPA_MAX_ITER needs a wider stamp (the hull-order tie-break of find_closest_face would silently truncate otherwise).
static_assert((2 * EPA_MAX_ITER + 4) * EPA_MAX_ITER + 4 < (1 << 14), "EPA face stamps no longer fit 14 bits");
constexpr int EPA_NULL = 255;

template <typename T>
struct PW0 {  // simplex-vertex payload: support point on shape 0 (w1 = w0 - w)
  V3<T> w0;
};
template <typename T>
HFCL_HD PW0<T> psel(bool c, const PW0<T>& a, const PW0<T>& b) {
  PW0<T> r;
  r.w0 = sel(c, a.w0, b.w0);
  return r;
}

// Scratch layout (SoA), sized for CAP iterations: CAP+4 vertices, 2*CAP+4 faces (gjk.cpp:1020-1021).
// CAP = 64 is the reference capacity; the fast kernel uses a smaller CAP (several polytopes per
// wave fit in LDS) and hands polytopes that outgrow it to the full-capacity kernel.
// Per-face connectivity: three 32-bit words (12 bytes per face; it was 16 with a 32-bit stamp of its own: the block
// size bounds the number of resident waves of the EPA kernels).  Single fields are updated with byte stores (two lanes
// may bind different edges of the same kept face concurrently, so no read-modify-write).  The append stamp (the
// reference's hull list position; 14 bits: a polytope creates far fewer than 16384 faces in 64 iterations) is split
// over the two spare bytes: both are written when the face is created and only read while it is in the hull.
struct alignas(4) FaceTopo {
  uint32_t vf;  // vertex ids of the 3 corners (bytes 0-2); byte 3: bit0 in hull, bit1 ignore, bits 2-7 = stamp >> 8
  uint32_t ap;  // neighbour face across edge 0..2 (bytes 0-2), pass mark (byte 3)
  uint32_t ae;  // edge index on the neighbour's side for edge 0..2 (bytes 0-2), stamp & 255 (byte 3)
  HFCL_HD int vid(int e) const { return int((vf >> (8 * e)) & 255u); }
  HFCL_HD int flag() const { return int((vf >> 24) & 3u); }
  HFCL_HD int adj(int e) const { return int((ap >> (8 * e)) & 255u); }
  HFCL_HD int pass() const { return int(ap >> 24); }
  HFCL_HD int adje(int e) const { return int((ae >> (8 * e)) & 255u); }
  HFCL_HD int stamp() const { return int((ae >> 24) | ((vf >> 26) << 8)); }
};
HFCL_HD uint8_t* topo_bytes(FaceTopo& t, int word) { return reinterpret_cast<uint8_t*>(&t) + 4 * word; }

template <typename T>
struct alignas(4 * sizeof(T) > 16 ? 16 : 4 * sizeof(T)) Quad {  // one LDS vector load/store per record
  T x, y, z, w;
};
// What has to travel with a polytope that outgrew its small scratch block so that the full-capacity
// tier can continue it instead of starting over (the first CAP iterations are the reference's in
// either tier, so continuing is bit-identical to restarting).
struct EpaHeader {
  int32_t closest, iterations, pass, status, num_vertices, hull_count, stock_top, stamp, hw;
};
// Where the shape-0 support point of every polytope vertex lives (needed once, for the witness points):
//   V0_BLOCK   in the scratch block itself;
//   V0_EXTERN  in a caller-provided array outside the block (global memory in the full-capacity kernel, whose LDS
//              block bounds its occupancy);
//   V0_TAG     nowhere: the fourth component of the vertex record carries a tag the caller can turn back into the point
//              (tag >= 0: vertex index of shape 0's hull; tag < 0: vertex -1-tag of GJK's final simplex), see
//              Epa::v0r.  The convex x convex fast tier works this way.
// (The first two are the former `bool V0IN` values false / true.)
enum { V0_EXTERN = 0, V0_BLOCK = 1, V0_TAG = 2 };
template <typename T, int N, bool ON>
struct V0Store {
  Quad<T> v0[N];  // vertex w0 (xyz)
};
template <typename T, int N>
struct V0Store<T, N, false> {};
template <typename T, int CAP, int V0M = V0_BLOCK>
struct EpaScratch : V0Store<T, CAP + 4, V0M == V0_BLOCK> {
  static constexpr int NV = CAP + 4;
  static constexpr int NF = 2 * CAP + 4;
  static constexpr int V0MODE = V0M;
  // horizon entries: kept face | its edge << HZ_SHIFT; one byte where the face ids leave two bits free
  typedef typename std::conditional<(NF <= 64), uint8_t, uint16_t>::type HzT;
  static constexpr int HZ_SHIFT = NF <= 64 ? 6 : 8;
  Quad<T> vw[NV];   // vertex w (xyz); w: tag (V0_TAG)
  Quad<T> fn[NF];   // face normal (xyz) and distance (w)
  FaceTopo ft[NF];  // connectivity + flags + stamp of a face, one 12-byte record
  union {
    uint16_t stack[NF];  // silhouette-walk frames: face | edge<<8 | stage<<10
    EpaHeader hdr;       // loop state of a polytope that is handed over to the full-capacity tier (the walk is over then)
  };
  uint32_t top;        // stock top while faces are being released by several lanes
  HzT hz[NF];          // horizon edges in walk order
  uint8_t stock[NF];   // free-face stack
  static_assert(sizeof(uint16_t) * NF >= sizeof(EpaHeader), "the header overlays the walk stack");
};
// A polytope saved for the tier hand-over, one format whatever way the saving tier kept the shape-0 support points:
// the block without them (= the V0_TAG layout; the members after the V0Store base are laid out the same in every
// mode) plus the points as coordinates.
template <typename T, int CAP>
struct EpaSaved {
  EpaScratch<T, CAP, V0_TAG> blk;
  Quad<T> v0[CAP + 4];
};

// All-reduce over a lane group: stage M = 1, 2, 4, .. < W hands every lane the value of a partner lane
// (Grp::exchange<M>) such that after the last stage all lanes have seen all values.  The partner of a
// stage is a lane of the "other half" at that level, not necessarily lane ^ M: only reductions with a
// total order (min/max with a unique tie-break) may be built on it.
template <int W, int M = 1, class F>
HFCL_HD void butterfly_stages(F&& f) {
  if constexpr (M < W) {
    f(std::integral_constant<int, M>());
    butterfly_stages<W, 2 * M>(f);
  }
}
// Lane-group operations.  W = 1 on the host validation build.
template <int W_>
struct SerialGroup {
  static constexpr int W = 1;
  static HFCL_HD int lane() { return 0; }
  template <int M, class X> static HFCL_HD X exchange(X v) { return v; }
  static HFCL_HD void sync() {}
  static HFCL_HD uint32_t atomic_inc(uint32_t* p) { return (*p)++; }
  // the lanes of the group for which `pred` holds: as a mask (bit = lane in group), and counted (below the caller / in all)
  static HFCL_HD uint64_t ballot(bool pred) { return pred ? 1u : 0u; }
  static HFCL_HD void count(bool pred, int& below, int& total) {
    below = 0;
    total = pred ? 1 : 0;
  }
};

template <typename T>
struct EpaLoop {  // state of the expansion loop between two trips
  int closest, iterations, pass;
  V3<T> outer_n;
  T outer_d;
  int o0, o1, o2;
};

template <typename T>
struct EpaResult {
  int status;
  int iterations;
  V3<T> normal;
  T depth;
  // result face (reference order) or single vertex for FallBack: w and w0 of its 3 vertices
  V3<T> rw0_, rw1_, rw2_, r00, r01, r02;
};

// Resolver of V0_TAG tags for the modes that do not use tags (never called).
struct NoTags {
  template <typename I> HFCL_HD int operator()(I) const { return 0; }
};
// sup(dir, w, w0[, tag]): supports that can name the shape-0 vertex they return take a fourth argument
template <bool TAGGED, typename T, class Sup>
HFCL_HD void epa_support(Sup& sup, const V3<T>& dir, V3<T>& w, V3<T>& w0, int& tag) {
  if constexpr (TAGGED) {
    sup(dir, w, w0, tag);
  } else {
    sup(dir, w, w0);
    tag = 0;
  }
}

// The plane of a polytope face (a, b, c): the arithmetic of newFace (:1081-1137) on its own, so that every place that
// builds a face -- Epa::face_geometry in a scratch block, epa_prepare_tetrahedron in registers -- rounds the same way.
// Returns 0 when the face is kept, else the status newFace sets (NonConvex / Degenerated); flag: 1 in hull, 3 in hull + ignore.
template <typename T>
HFCL_HD int epa_face_plane(const V3<T>& a, const V3<T>& b, const V3<T>& c, T tolerance, bool force, V3<T>& n, T& dist, int& flag) {
  n = cross(b - a, c - a);
  int fail = 0;
  flag = 1;
  dist = T(0);
  if (norm(n) > Lim<T>::eps()) {
    n = normalized(n);
    const T a_dot_nab = dot(a, cross(b - a, n));
    const T b_dot_nbc = dot(b, cross(c - b, n));
    const T c_dot_nca = dot(c, cross(a - c, n));
    T d;
    if (a_dot_nab >= -tolerance && b_dot_nbc >= -tolerance && c_dot_nca >= -tolerance) {
      d = dot(a, n);
    } else {
      d = Lim<T>::max();
      flag = 3;  // in hull + ignore
    }
    dist = d;
    if (!(d >= -tolerance || force)) fail = EPA_NON_CONVEX;
  } else {
    fail = EPA_DEGENERATED;
  }
  return fail;
}

// ---------------------------------------------------------------------------------------
// The convex x convex fast tier in three stages (hfcl_k_epa.hip: k_epa_prepare / k_epa_loop / k_epa_records).
// What EPA::evaluate does before its loop for a seed of rank 4 -- orientation, the first tetrahedron, its closest face
// (:1188-1230) -- and after it -- witness points, the record (narrowphase.h:658-711) -- is serial work per polytope; inside
// the streaming kernel an 8-lane group executes it redundantly while the live groups of the wave wait.  Here one LANE per
// polytope does it, in kernels of their own around the loop kernel, and the two hand each other a block per polytope:
//   prepare -> loop : EpaReady   the oriented tetrahedron (vertex records with their tags, face planes, ignore flags, the first
//                                closest face) and what a group needs to evaluate supports (hull offsets, relative pose);
//   loop -> records : the same block, its pose / vertex area overwritten with the loop's result (EpaLoopOut) and `state` set.
// ---------------------------------------------------------------------------------------
enum { EPA_READY_PENDING = 0u, EPA_READY_DONE = 1u, EPA_READY_HANDED_OVER = 2u, EPA_READY_NONE = 3u };
template <typename T>
struct alignas(16) EpaReady {
  uint32_t seed;            // slot of the polytope's seed in the convex x convex queue (item `seed` of that queue)
  uint32_t voff_a, voff_b;  // first vertex of the two hulls in the library's vertex table
  // bits 0-5 / 6-11: vertices of hull a / b; 12-13: first closest face; 14-17: ignore flag of face 0..3; 18: relative pose is the identity
  uint32_t packed;
  T md[12];                 // MDiff: oR1 rows, ot1
  Quad<T> vw[4];            // vertex records of the oriented tetrahedron (w: tag)
  Quad<T> fn[4];            // face planes
  uint32_t state;           // EPA_READY_*
  uint32_t pair;            // the query ...
  uint32_t gjk_iters;       // ... and what its record says about GJK
  uint32_t pad_;
};
template <typename T>
struct __attribute__((may_alias)) EpaLoopOut {  // overlays EpaReady::md .. (the loop is over: the block's pose and vertices are no longer needed)
  int32_t status, iterations;
  T nx, ny, nz, depth;       // the last valid `outer` face: normal and distance (EpaLoop::outer_n / outer_d)
  T rw[9];                   // w of its three vertices
  int32_t tag[3];            // ... and their tags
};
static_assert(sizeof(EpaLoopOut<float>) <= sizeof(float) * 12 + sizeof(Quad<float>) * 4, "the result overlays the pose and vertex area");
// connectivity of the first tetrahedron as Epa::begin binds it (faces (0,1,2) (1,0,3) (2,1,3) (0,2,3), stamps 0..3)
struct EpaTetraTopo {
  uint32_t vf[4], ap[4], ae[4];
};
constexpr EpaTetraTopo epa_tetra_topo() {
  EpaTetraTopo t{{0, 0, 0, 0}, {0, 0, 0, 0}, {0, 0, 0, 0}};
  const int corner[4][3] = {{0, 1, 2}, {1, 0, 3}, {2, 1, 3}, {0, 2, 3}};
  int adj[4][3] = {{0, 0, 0}, {0, 0, 0}, {0, 0, 0}, {0, 0, 0}}, adje[4][3] = {{0, 0, 0}, {0, 0, 0}, {0, 0, 0}, {0, 0, 0}};
  const int binds[6][4] = {{0, 0, 1, 0}, {0, 1, 2, 0}, {0, 2, 3, 0}, {1, 1, 3, 2}, {1, 2, 2, 1}, {2, 2, 3, 1}};  // (fa, ea, fb, eb), :1222-1227
  for (int k = 0; k < 6; ++k) {
    adj[binds[k][0]][binds[k][1]] = binds[k][2];
    adje[binds[k][0]][binds[k][1]] = binds[k][3];
    adj[binds[k][2]][binds[k][3]] = binds[k][0];
    adje[binds[k][2]][binds[k][3]] = binds[k][1];
  }
  for (int f = 0; f < 4; ++f) {
    t.vf[f] = uint32_t(corner[f][0]) | (uint32_t(corner[f][1]) << 8) | (uint32_t(corner[f][2]) << 16);
    t.ap[f] = uint32_t(adj[f][0]) | (uint32_t(adj[f][1]) << 8) | (uint32_t(adj[f][2]) << 16);
    t.ae[f] = uint32_t(adje[f][0]) | (uint32_t(adje[f][1]) << 8) | (uint32_t(adje[f][2]) << 16) | (uint32_t(f) << 24);
  }
  return t;
}
// Epa::begin for a seed of rank 4 (w[i] with tag -1-i), by one lane, nothing but registers.  Returns false when the
// reference falls back (origin not enclosed, degenerate face: :1299-1315); otherwise vw / fn / flags / closest describe the
// block Epa::begin would have left.
template <typename T>
HFCL_HD bool epa_prepare_tetrahedron(const V3<T>* w, T tolerance, Quad<T>* vw, Quad<T>* fn, int* flags, int& closest) {
  // GJK::encloseOrigin, rank 4 (:483-488)
  if (!(habs(triple(w[0] - w[3], w[1] - w[3], w[2] - w[3])) > T(0))) return false;
  const bool swap01 = dot(w[0] - w[3], cross(w[1] - w[3], w[2] - w[3])) < T(0);  // :1196-1201
  const V3<T> v0 = swap01 ? w[1] : w[0], v1 = swap01 ? w[0] : w[1];
  vw[0] = Quad<T>{v0.x, v0.y, v0.z, T(swap01 ? -2 : -1)};
  vw[1] = Quad<T>{v1.x, v1.y, v1.z, T(swap01 ? -1 : -2)};
  vw[2] = Quad<T>{w[2].x, w[2].y, w[2].z, T(-3)};
  vw[3] = Quad<T>{w[3].x, w[3].y, w[3].z, T(-4)};
  const V3<T> p[4] = {v0, v1, w[2], w[3]};
  const int corner[4][3] = {{0, 1, 2}, {1, 0, 3}, {2, 1, 3}, {0, 2, 3}};
  bool ok = true;
#if defined(__HIP_DEVICE_COMPILE__)
#pragma unroll
#endif
  for (int f = 0; f < 4; ++f) {
    V3<T> n;
    T d;
    const int fail = epa_face_plane(p[corner[f][0]], p[corner[f][1]], p[corner[f][2]], tolerance, true, n, d, flags[f]);
    fn[f] = Quad<T>{n.x, n.y, n.z, d};
    ok = ok && fail == 0;
  }
  if (!ok) return false;  // hull_count != 4
  // findClosestFace over the four faces (stamps 0..3: the later face wins a tie; all ignored: the list head, face 3)
  T best = Lim<T>::max();
  int best_f = EPA_NULL;
#if defined(__HIP_DEVICE_COMPILE__)
#pragma unroll
#endif
  for (int f = 0; f < 4; ++f) {
    if (flags[f] & 2) continue;
    const T sq = fn[f].w * fn[f].w;
    if (sq < best || (sq == best && best_f != EPA_NULL)) {
      best = sq;
      best_f = f;
    }
  }
  closest = best_f != EPA_NULL ? best_f : 3;
  return true;
}

// HFCL_EPA_PAR_HZ: 1 = the fp32 convex x convex fast tier (the V0_TAG blocks) finds the horizon of an expansion with all
// lanes at once (Epa::silhouette_parallel) instead of walking it; 2 = every fp32 polytope (validation builds); 0 = never.
// fp64 always walks: its statuses and iteration counts are the reference's to the letter.  (The full-capacity fp32 tier walks
// too: with 132 face slots, 9 per lane, classifying all of them costs more than walking a dozen: 0.215 against 0.18 ms.)
#ifndef HFCL_EPA_PAR_HZ
#define HFCL_EPA_PAR_HZ 1
#endif
#ifndef HFCL_EPA_RESUME_PAR_HZ
#define HFCL_EPA_RESUME_PAR_HZ 0  // 1: k_epa_resume_cc finds horizons with all lanes as well (A/B)
#endif
#ifndef HFCL_EPA_FLAT
#define HFCL_EPA_FLAT 1  // find_closest_face_flat in the fp32 convex x convex fast tier (k_epa_loop 0.975 -> 0.961 ms; the horizon search built the same way -- ballots instead of pass marks and LDS atomics -- is 3 % slower: profiles/r05_b)
#endif
// GJK::encloseOrigin :437-492 on the vertices 0..rank-1 of a store (reference order): st.vw(i) reads vertex i, st.eo_support(sup, dir, ...)
// evaluates a support point, st.eo_put(i, w, w0, tag) stores vertex i.  One function for the polytope block in LDS (Epa) and for the
// registers of a lane that prepares a polytope on its own (EpaRegStore): the same operations in the same order.
template <typename T, class Store, class Sup>
HFCL_HD bool epa_enclose_origin(Store& st, int& rank, Sup& sup) {
  const int base = rank;
  int c1 = 0, c2 = 0, c3 = 0;  // candidate counters of the rank-1/2/3 levels (no arrays: registers)
  bool entering = true;
  for (;;) {
    if (entering) {
      if (rank == 4) {
        if (habs(triple(st.vw(0) - st.vw(3), st.vw(1) - st.vw(3), st.vw(2) - st.vw(3))) > T(0)) return true;
        if (base == 4) return false;
        --rank;  // parent removes the vertex
        entering = false;
        continue;
      }
      if (rank == 1) c1 = 0;
      if (rank == 2) c2 = 0;
      if (rank == 3) c3 = 0;
    }
    // try the next candidate direction at this level
    V3<T> dir = mk<T>(T(0), T(0), T(0));
    bool have = false;
    while (!have) {
      const int c = (rank == 1) ? c1 : ((rank == 2) ? c2 : c3);
      if (rank == 1) {
        if (c >= 6) break;
        const int i = c >> 1;  // both the "+" and the "-" attempt use +e_i (reference quirk :443-448)
        dir = mk<T>(i == 0 ? T(1) : T(0), i == 1 ? T(1) : T(0), i == 2 ? T(1) : T(0));
        have = true;
      } else if (rank == 2) {
        if (c >= 6) break;
        const int i = c >> 1;
        const V3<T> d = st.vw(1) - st.vw(0);
        const V3<T> axis = mk<T>(i == 0 ? T(1) : T(0), i == 1 ? T(1) : T(0), i == 2 ? T(1) : T(0));
        const V3<T> p = cross(d, axis);
        if (is_zero(p)) {
          c2 = (i + 1) * 2;
          continue;
        }
        dir = (c & 1) ? -p : p;
        have = true;
      } else {  // rank 3
        if (c >= 2) break;
        const V3<T> axis = cross(st.vw(1) - st.vw(0), st.vw(2) - st.vw(0));
        if (is_zero(axis)) {
          c3 = 2;
          continue;
        }
        dir = (c & 1) ? -axis : axis;
        have = true;
      }
    }
    if (!have) {
      if (rank == base) return false;
      --rank;
      entering = false;
      continue;
    }
    if (rank == 1) ++c1;
    if (rank == 2) ++c2;
    if (rank == 3) ++c3;
    V3<T> w, w0;
    int tag;
    st.eo_support(sup, dir, w, w0, tag);
    st.eo_put(rank, w, w0, tag);
    ++rank;
    entering = true;
  }
}
// the four vertices of a polytope-to-be in the registers of one lane (k_epa_prepare_general)
template <typename T>
struct EpaRegStore {
  V3<T> w[4], w0[4];
  HFCL_HD const V3<T>& vw(int i) const { return w[i]; }
  template <class Sup>
  HFCL_HD void eo_support(Sup& sup, const V3<T>& dir, V3<T>& ws, V3<T>& w0s, int& tag) const {
    sup(dir, ws, w0s);
    tag = 0;
  }
  HFCL_HD void eo_put(int i, const V3<T>& ws, const V3<T>& w0s, int) {  // (i = 1, 2 or 3; constant indices keep the arrays in registers)
    if (i == 1) { w[1] = ws; w0[1] = w0s; }
    if (i == 2) { w[2] = ws; w0[2] = w0s; }
    if (i == 3) { w[3] = ws; w0[3] = w0s; }
  }
};

// ---------------------------------------------------------------------------------------
// The same three stages for pairs of any convex kinds (V0_BLOCK blocks: supports are coordinates, not vertex tags; seeds of any rank:
// encloseOrigin runs in the preparing lane with the pair's support functions; fp64 as well: k_epa_prepare_general / k_epa_loop_general /
// k_epa_records_general in hfcl_k_epa.hip).
// ---------------------------------------------------------------------------------------
template <typename T>
struct alignas(16) EpaReadyG {
  uint32_t seed;            // slot of the seed in its queue (bit 31: the queue at the top end of epa_queue)
  uint32_t pair;
  uint32_t sid1, sid2;      // the two shapes
  uint32_t packed;          // bits 0-1: first closest face; 2-5: ignore flag of face 0..3; 6: relative pose is the identity
  uint32_t gjk_iters;
  uint32_t state;           // EPA_READY_*
  uint32_t pad_;
  T md[12];                 // MDiff: oR1 rows, ot1
  Quad<T> vw[4];            // vertex records of the oriented tetrahedron
  Quad<T> v0[4];            // ... their support points on shape 0
  Quad<T> fn[4];            // face planes
};
template <typename T>
struct __attribute__((may_alias)) EpaLoopOutG {  // overlays EpaReadyG::md .. once the loop is over
  int32_t status, iterations;
  T nx, ny, nz, depth;       // the last valid `outer` face: normal and distance
  T rw[9];                   // w of its three vertices
  T r0[9];                   // ... and their support points on shape 0
};
static_assert(sizeof(EpaLoopOutG<float>) <= sizeof(float) * 12 + 2 * sizeof(Quad<float>) * 4 && sizeof(EpaLoopOutG<double>) <= sizeof(double) * 12 + 2 * sizeof(Quad<double>) * 4,
              "the result overlays the pose and vertex area");
// Epa::begin for a seed of any rank, by one lane: st holds the seed's vertices (reference order), `sup` evaluates the pair's supports.
// Returns false when the reference falls back (:1299-1315); otherwise vw / v0 / fn / flags / closest describe the block begin() leaves.
template <typename T, class Sup>
HFCL_HD bool epa_prepare_general(EpaRegStore<T>& st, int rank, T tolerance, Sup& sup, Quad<T>* vw, Quad<T>* v0, Quad<T>* fn, int* flags, int& closest) {
  const bool enclosed = epa_enclose_origin<T>(st, rank, sup);
  if (!(rank > 1 && enclosed)) return false;
  const bool swap01 = dot(st.w[0] - st.w[3], cross(st.w[1] - st.w[3], st.w[2] - st.w[3])) < T(0);  // :1196-1201
  const V3<T> p[4] = {swap01 ? st.w[1] : st.w[0], swap01 ? st.w[0] : st.w[1], st.w[2], st.w[3]};
  const V3<T> p0[4] = {swap01 ? st.w0[1] : st.w0[0], swap01 ? st.w0[0] : st.w0[1], st.w0[2], st.w0[3]};
#if defined(__HIP_DEVICE_COMPILE__)
#pragma unroll
#endif
  for (int i = 0; i < 4; ++i) {
    vw[i] = Quad<T>{p[i].x, p[i].y, p[i].z, T(0)};
    v0[i] = Quad<T>{p0[i].x, p0[i].y, p0[i].z, T(0)};
  }
  const int corner[4][3] = {{0, 1, 2}, {1, 0, 3}, {2, 1, 3}, {0, 2, 3}};
  bool ok = true;
#if defined(__HIP_DEVICE_COMPILE__)
#pragma unroll
#endif
  for (int f = 0; f < 4; ++f) {
    V3<T> n;
    T d;
    const int fail = epa_face_plane(p[corner[f][0]], p[corner[f][1]], p[corner[f][2]], tolerance, true, n, d, flags[f]);
    fn[f] = Quad<T>{n.x, n.y, n.z, d};
    ok = ok && fail == 0;
  }
  if (!ok) return false;  // hull_count != 4
  T best = Lim<T>::max();
  int best_f = EPA_NULL;
#if defined(__HIP_DEVICE_COMPILE__)
#pragma unroll
#endif
  for (int f = 0; f < 4; ++f) {
    if (flags[f] & 2) continue;
    const T sq = fn[f].w * fn[f].w;
    if (sq < best || (sq == best && best_f != EPA_NULL)) {
      best = sq;
      best_f = f;
    }
  }
  closest = best_f != EPA_NULL ? best_f : 3;
  return true;
}

template <typename T, class Grp, int CAP = EPA_MAX_ITER, int V0M = V0_BLOCK>
struct Epa {
  static constexpr bool TAGGED = V0M == V0_TAG;
  // (blocks of the reference's capacity walk: the tier that continues a handed-over convex x convex polytope does what the general
  // full-capacity tier does, so that a record does not depend on which of the two continued it -- i.e. on the size of its batch)
  static constexpr bool PARALLEL_HORIZON = sizeof(T) == 4 && (HFCL_EPA_PAR_HZ >= 2 || (HFCL_EPA_PAR_HZ == 1 && V0M == V0_TAG && (CAP < EPA_MAX_ITER || HFCL_EPA_RESUME_PAR_HZ)));
  typedef EpaScratch<T, CAP, V0M> Block;
  // The branch-free form of the closest-face scan (find_closest_face_flat): the small blocks of the fp32 convex x convex fast tier.
  static constexpr bool FLAT_CF = PARALLEL_HORIZON && HFCL_EPA_FLAT && EpaScratch<T, CAP, V0M>::NF <= 64;
  Block* m;
  Quad<T>* v0p;  // m->v0, or the caller's array (V0_EXTERN); unused with tags
  T tolerance;
  int max_iterations;  // the request's (reference) limit
  int cap_iterations;  // min(max_iterations, CAP): what this scratch block can hold
  bool overflow;       // the polytope outgrew CAP although the reference's capacity would not be exhausted
  bool resumable;      // ... at an iteration boundary: the scratch block (incl. its hdr) describes it completely
  int status;
  int num_vertices;
  int hull_count;
  int stock_top;
  int stamp;
  int pending_release;  // pass mark of faces still to be released by the next find_closest_face(), or -1
  // High-water mark of the face store: slots >= hw were never used.  The stock hands out slot 0, 1, 2, ... and reuses
  // released slots first, so a polytope at iteration i occupies the first ~2i + 4 slots of its block, and every scan over
  // "all faces" (closest face, releases, horizon) stops at hw instead of the block's capacity -- half the slots of a
  // fast-tier block for the average polytope, a third of the full tier's 132.
  int hw;

  HFCL_HD V3<T> vw(int i) const {
    const Quad<T> q = m->vw[i];
    return mk<T>(q.x, q.y, q.z);
  }
  HFCL_HD V3<T> v0(int i) const {  // V0_BLOCK / V0_EXTERN
#if defined(__HIP_DEVICE_COMPILE__)
    if constexpr (V0M == V0_EXTERN) {
      // written by lane 0 of this group, possibly long ago and for an earlier polytope of the same slot:
      // read past the per-CU cache
      const T* q = &v0p[i].x;
      return mk<T>(__hip_atomic_load(q, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT),
                   __hip_atomic_load(q + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT),
                   __hip_atomic_load(q + 2, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT));
    }
#endif
    const Quad<T> q = v0p[i];
    return mk<T>(q.x, q.y, q.z);
  }
  // the shape-0 support point of vertex i in any mode; `tags`: int -> V3 (V0_TAG only)
  template <class Tags>
  HFCL_HD V3<T> v0r(int i, const Tags& tags) const {
    if constexpr (TAGGED)
      return tags(int(m->vw[i].w));
    else
      return v0(i);
  }
  HFCL_HD void set_vert(int i, const V3<T>& w, const V3<T>& w0, int tag = 0) {
    m->vw[i] = Quad<T>{w.x, w.y, w.z, T(tag)};
    if constexpr (V0M == V0_BLOCK) v0p[i] = Quad<T>{w0.x, w0.y, w0.z, T(0)};
    if constexpr (V0M == V0_EXTERN)
      if (Grp::lane() == 0) v0p[i] = Quad<T>{w0.x, w0.y, w0.z, T(0)};
  }
  HFCL_HD static unsigned hz_pack(int f, int e) { return unsigned(f) | (unsigned(e) << Block::HZ_SHIFT); }
  HFCL_HD static int hz_face(unsigned h) { return int(h & ((1u << Block::HZ_SHIFT) - 1u)); }
  HFCL_HD static int hz_edge(unsigned h) { return int((h >> Block::HZ_SHIFT) & 3u); }
  HFCL_HD V3<T> fn(int f) const {
    const Quad<T> q = m->fn[f];
    return mk<T>(q.x, q.y, q.z);
  }
  HFCL_HD T fd(int f) const { return m->fn[f].w; }
  HFCL_HD void set_adj(int f, int e, int nb, int nb_edge) {
    topo_bytes(m->ft[f], 1)[e] = uint8_t(nb);
    topo_bytes(m->ft[f], 2)[e] = uint8_t(nb_edge);
  }
  HFCL_HD void set_flag(int f, int v) { topo_bytes(m->ft[f], 0)[3] = uint8_t(v); }
  HFCL_HD void set_pass(int f, int v) { topo_bytes(m->ft[f], 1)[3] = uint8_t(v); }
  HFCL_HD void bind(int fa, int ea, int fb, int eb) {  // gjk.h:312-320
    set_adj(fa, ea, fb, eb);
    set_adj(fb, eb, fa, ea);
  }
  HFCL_HD void hull_remove(int f) {
    set_flag(f, m->ft[f].flag() & ~1);
    --hull_count;
    m->stock[stock_top++] = uint8_t(f);
  }

  HFCL_HD void reset(Block* mem, int max_it, T tol, Quad<T>* v0_ext = nullptr) {  // :1014-1037
    m = mem;
    if constexpr (V0M == V0_BLOCK)
      v0p = mem->v0;
    else
      v0p = v0_ext;
    tolerance = tol;
    max_iterations = max_it;
    cap_iterations = max_it < CAP ? max_it : CAP;
    overflow = false;
    resumable = false;
    status = EPA_DID_NOT_RUN;
    num_vertices = 0;
    hull_count = 0;
    stamp = 0;
    pending_release = -1;
    hw = 0;
    const int nf = 2 * cap_iterations + 4;
    // face 0 on top of the stock, as in the reference (stock filled in reverse order)
    for (int i = Grp::lane(); i < nf; i += Grp::W) {
      m->stock[i] = uint8_t(nf - 1 - i);
      set_flag(i, 0);
    }
    stock_top = nf;
    Grp::sync();
  }

  // Geometry part of newFace (:1081-1137) for the triangle (ia, ib, ic) stored in slot f.
  // Returns 0 when the face is kept, else the status newFace sets (NonConvex / Degenerated).
  HFCL_HD int face_geometry(int f, int ia, int ib, int ic, bool force, int face_stamp) {
    V3<T> n;
    T dist;
    int flag;
    const int fail = epa_face_plane(vw(ia), vw(ib), vw(ic), tolerance, force, n, dist, flag);
    m->fn[f] = Quad<T>{n.x, n.y, n.z, dist};
    // corners + flags + high stamp bits in one word; the pass mark is cleared; adjacency bytes are written by the binds
    m->ft[f].vf = uint32_t(ia) | (uint32_t(ib) << 8) | (uint32_t(ic) << 16) | (uint32_t(flag) << 24) |
                  ((uint32_t(face_stamp) >> 8) << 26);
    topo_bytes(m->ft[f], 2)[3] = uint8_t(face_stamp);
    set_pass(f, 0);
    return fail;
  }

  // newFace :1068-1138 (serial form, used for the initial tetrahedron).  Face index or EPA_NULL.
  HFCL_HD int new_face(int ia, int ib, int ic, bool force) {
    if (stock_top == 0) {
      if (cap_iterations < max_iterations) overflow = true;  // the reference still has faces in stock
      status = EPA_OUT_OF_FACES;
      return EPA_NULL;
    }
    const int f = m->stock[--stock_top];
    ++hull_count;
    if (f + 1 > hw) hw = f + 1;
    const int fail = face_geometry(f, ia, ib, ic, force, stamp++);
    if (!fail) return f;
    status = fail;
    hull_remove(f);
    return EPA_NULL;
  }

  // findClosestFace :1141-1154: min d^2 over non-ignored hull faces, first in list order
  // (= largest append stamp) on ties; if every face is ignored: the list head.
  // The same scan also releases the faces the last expansion made obsolete (pending_release = their pass
  // mark): they go back to the stock through a group-shared counter, in no particular order (slot
  // numbers never influence a result, ties are decided by the stamps).
  // The same scan without a branch or an LDS atomic: every lane loads its share of the block's slots at once (the trip count
  // of the loop above differs from group to group of a wave and each trip is two dependent LDS round trips), decides with
  // selects, and the released faces find their stock slots by counting (Grp::count) -- in the order the atomic hands them
  // out, slot j * W + lane ascending.  A lane without a face to release stores into a byte of the walk stack nobody reads.
  HFCL_HD int find_closest_face_flat() {
    const int rel = pending_release;
    pending_release = -1;
    Grp::sync();
    const int nf = hw;
    constexpr int PER_LANE = (Block::NF + Grp::W - 1) / Grp::W;
    uint8_t* const dummy = reinterpret_cast<uint8_t*>(m->stack) + Grp::lane();
    FaceTopo t[PER_LANE];
    T dist[PER_LANE];
#if defined(__HIP_DEVICE_COMPILE__)
#pragma unroll
#endif
    for (int j = 0; j < PER_LANE; ++j) {
      const int f = Grp::lane() + j * Grp::W;
      const int fc = f < nf ? f : 0;
      t[j] = m->ft[fc];
      dist[j] = m->fn[fc].w;
    }
    T best = Lim<T>::max();
    int best_stamp = -1, best_f = EPA_NULL;
    int head_stamp = -1, head_f = EPA_NULL;
    int released = 0;
#if defined(__HIP_DEVICE_COMPILE__)
#pragma unroll
#endif
    for (int j = 0; j < PER_LANE; ++j) {
      const int f = Grp::lane() + j * Grp::W;
      const int fl = t[j].flag();
      const bool in_hull = (f < nf) & ((fl & 1) != 0);
      const bool release = in_hull & (rel >= 0) & (t[j].pass() == rel);
      int below, total;
      Grp::count(release, below, total);
      *(release ? topo_bytes(m->ft[f], 0) + 3 : dummy) = uint8_t(0);  // set_flag(f, 0)
      *(release ? &m->stock[stock_top + released + below] : dummy) = uint8_t(f);
      released += total;
      const bool cand = in_hull & !release;
      const int st = t[j].stamp();
      const bool th = cand & (st > head_stamp);
      head_stamp = th ? st : head_stamp;
      head_f = th ? f : head_f;
      const T sq = dist[j] * dist[j];
      const bool take = cand & !(fl & 2) & ((sq < best) | ((sq == best) & (st > best_stamp) & (best_f != EPA_NULL)));  // (no short circuits: selects, not branches)
      best = take ? sq : best;
      best_stamp = take ? st : best_stamp;
      best_f = take ? f : best_f;
    }
    butterfly_stages<Grp::W>([&](auto stage) {
      constexpr int M = decltype(stage)::value;
      const T ob = Grp::template exchange<M>(best);
      const int os = Grp::template exchange<M>(best_stamp), of = Grp::template exchange<M>(best_f);
      const int ohs = Grp::template exchange<M>(head_stamp), ohf = Grp::template exchange<M>(head_f);
      const bool take = (of != EPA_NULL) & ((best_f == EPA_NULL) | (ob < best) | ((ob == best) & (os > best_stamp)));
      best = take ? ob : best;
      best_stamp = take ? os : best_stamp;
      best_f = take ? of : best_f;
      const bool th = ohs > head_stamp;
      head_stamp = th ? ohs : head_stamp;
      head_f = th ? ohf : head_f;
    });
    stock_top += released;
    hull_count -= released;
    return best_f != EPA_NULL ? best_f : head_f;
  }

  HFCL_HD int find_closest_face() {
    if constexpr (FLAT_CF) return find_closest_face_flat();
    const int rel = pending_release;
    pending_release = -1;
    if (rel >= 0 && Grp::lane() == 0) m->top = uint32_t(stock_top);
    Grp::sync();
    const int nf = hw;
    T best = Lim<T>::max();
    int best_stamp = -1, best_f = EPA_NULL;
    int head_stamp = -1, head_f = EPA_NULL;
    for (int f = Grp::lane(); f < nf; f += Grp::W) {
      const FaceTopo t = m->ft[f];
      const int fl = t.flag();
      if (!(fl & 1)) continue;
      if (rel >= 0 && t.pass() == rel) {
        set_flag(f, 0);
        m->stock[Grp::atomic_inc(&m->top)] = uint8_t(f);
        continue;
      }
      const int st = t.stamp();
      if (st > head_stamp) {
        head_stamp = st;
        head_f = f;
      }
      if (fl & 2) continue;
      const T d = fd(f);
      const T sq = d * d;
      if (sq < best || (sq == best && st > best_stamp && best_f != EPA_NULL)) {
        best = sq;
        best_stamp = st;
        best_f = f;
      }
    }
    butterfly_stages<Grp::W>([&](auto stage) {
      constexpr int M = decltype(stage)::value;
      const T ob = Grp::template exchange<M>(best);
      const int os = Grp::template exchange<M>(best_stamp), of = Grp::template exchange<M>(best_f);
      const int ohs = Grp::template exchange<M>(head_stamp), ohf = Grp::template exchange<M>(head_f);
      const bool take = (of != EPA_NULL) & ((best_f == EPA_NULL) | (ob < best) | ((ob == best) & (os > best_stamp)));  // (selects: see HullRegs::support)
      best = take ? ob : best;
      best_stamp = take ? os : best_stamp;
      best_f = take ? of : best_f;
      const bool th = ohs > head_stamp;
      head_stamp = th ? ohs : head_stamp;
      head_f = th ? ohf : head_f;
    });
    if (rel >= 0) {
      Grp::sync();
      const int new_top = int(m->top);
      hull_count -= new_top - stock_top;
      stock_top = new_top;
    }
    return best_f != EPA_NULL ? best_f : head_f;
  }

  // Releases the faces marked `pass` right away (needed before the new faces are built only when the
  // stock holds fewer free faces than the horizon has edges).
  HFCL_HD void release_visible(int pass) {
    if (Grp::lane() == 0) m->top = uint32_t(stock_top);
    Grp::sync();
    const int nf = hw;
    for (int f = Grp::lane(); f < nf; f += Grp::W)
      if ((m->ft[f].flag() & 1) && m->ft[f].pass() == pass) {
        set_flag(f, 0);
        m->stock[Grp::atomic_inc(&m->top)] = uint8_t(f);
      }
    Grp::sync();
    const int new_top = int(m->top);
    hull_count -= new_top - stock_top;
    stock_top = new_top;
  }

  // Silhouette walk = the control flow of expand() (:1361-1449) from (f0, e0) without creating the
  // faces: horizon edges are appended to m->hz in the order the reference would create their
  // faces.  `level` emulates the reference's stock size along the walk (one pop per horizon edge,
  // one push when a visible face is left, :1437-1446) so that its OutOfFaces exit fires at the
  // same edge; `stop_kind/stop_at` record the first walk-level failure (OutOfFaces before the
  // stop_at-th face, or InvalidHull after stop_at faces).
  HFCL_HD void silhouette_walk(int pass, int f0, int e0, T dummy_precision, const V3<T>& ww, int& hz_count, int& level,
                               int& stop_kind, int& stop_at) {
    int sp = 0;
    m->stack[sp++] = uint16_t(f0 | (e0 << 8));
    while (sp > 0) {
      const unsigned fr = m->stack[sp - 1];
      const int f = fr & 255, e = (fr >> 8) & 3, stage = (fr >> 10) & 3;
      const int e1 = (e + 1) % 3, e2 = (e + 2) % 3;
      const FaceTopo t = m->ft[f];
      if (stage == 0) {
        if (t.pass() == pass) {
          stop_kind = EPA_INVALID_HULL;
          stop_at = hz_count;
          return;
        }
        if (dot(fn(f), ww - vw(t.vid(e))) < dummy_precision) {
          // case 1: the support point is "below" f: horizon edge, new face (f[e1], f[e], w)
          if (level == 0) {
            stop_kind = EPA_OUT_OF_FACES;
            stop_at = hz_count;
            return;
          }
          --level;
          m->hz[hz_count++] = typename Block::HzT(hz_pack(f, e));
          --sp;
          continue;
        }
        // case 2: above f
        set_pass(f, pass);
        m->stack[sp - 1] = uint16_t(f | (e << 8) | (1 << 10));
        m->stack[sp++] = uint16_t(t.adj(e1) | (t.adje(e1) << 8));
      } else if (stage == 1) {
        m->stack[sp - 1] = uint16_t(f | (e << 8) | (2 << 10));
        m->stack[sp++] = uint16_t(t.adj(e2) | (t.adje(e2) << 8));
      } else {
        ++level;  // the reference returns f to the stock here
        --sp;
      }
    }
  }

  // The same walk without the stock-level bookkeeping (valid whenever the horizon has no more edges
  // than there are free faces, checked by the caller): plain pre-order traversal, one frame per
  // visited face.  Children are pushed second-edge first so that the first-edge subtree is walked
  // first, as in the recursion; the three expand() calls of :1266-1268 are the three root frames.
  HFCL_HD void silhouette_walk_fast(int pass, int closest, T dummy_precision, const V3<T>& ww, int& hz_count,
                                    int& stop_kind, int& stop_at) {
    int sp = 0;
    const FaceTopo tc = m->ft[closest];
    m->stack[sp++] = uint16_t(tc.adj(2) | (tc.adje(2) << 8));
    m->stack[sp++] = uint16_t(tc.adj(1) | (tc.adje(1) << 8));
    m->stack[sp++] = uint16_t(tc.adj(0) | (tc.adje(0) << 8));
    while (sp > 0) {
      const unsigned fr = m->stack[--sp];
      const int f = fr & 255, e = (fr >> 8) & 3;
      const FaceTopo t = m->ft[f];
      if (t.pass() == pass) {
        stop_kind = EPA_INVALID_HULL;
        stop_at = hz_count;
        return;
      }
      if (dot(fn(f), ww - vw(t.vid(e))) < dummy_precision) {
        m->hz[hz_count++] = typename Block::HzT(hz_pack(f, e));
        continue;
      }
      set_pass(f, pass);
      const int e1 = (e + 1) % 3, e2 = (e + 2) % 3;
      m->stack[sp++] = uint16_t(t.adj(e2) | (t.adje(e2) << 8));
      m->stack[sp++] = uint16_t(t.adj(e1) | (t.adje(e1) << 8));
    }
  }

  // The horizon without a walk (fp32 fast tier).  expand() (:1361-1449) walks from the closest face over the faces the new
  // vertex is above and lists the edges where it meets a face the vertex is below: a chain of dependent LDS round trips
  // (frame, face record, plane, vertex: ~300 clocks per visited face, a dozen faces per expansion), executed by every lane
  // of the group, with the groups of a wave diverging on it.  Here every lane classifies its share of the hull's faces --
  // independent loads -- marks the visible ones with the pass mark the rest of the expansion works with, and appends
  // (kept face, edge) for every edge of a kept face whose neighbour is marked.  Two tables (vertex -> the new face that starts
  // / ends there along the horizon; they overlay the walk's stack) give every new face its two neighbours without an order
  // on the horizon.  What differs from the walk, and why this form is fp32 only: a face is tested against its first
  // vertex, not the vertex of the edge it was entered through (last bits); a face the vertex is above is removed wherever
  // it lies, not only when it is connected to the closest face through such faces; the new faces get their stamps in list
  // order, not walk order (stamps only break exact ties of the closest-face search).  The expansion remains a valid
  // hull update, so EPA converges to the same depth (tests/test_epa_ground_truth.py holds it to the qhull ground truth).
  // Returns false -- marks undone by the caller, the walk decides -- when the marked faces do not leave a horizon of simple
  // closed loops (a vertex with two outgoing or two incoming horizon edges, or an open end).
  HFCL_HD bool silhouette_parallel(int pass, int closest, T dummy_precision, const V3<T>& ww, int& hz_count) {
    const int nf = hw;
    uint8_t* const start_at = reinterpret_cast<uint8_t*>(m->stack);  // [vertex] -> horizon entry whose new face starts there
    uint8_t* const end_at = start_at + Block::NV;                     // [vertex] -> ... ends there
    static_assert(2 * Block::NV <= int(sizeof(uint16_t)) * Block::NF, "the vertex tables overlay the walk stack");
    // (fixed trip counts, loads of all of a lane's faces issued before any is used: the point is independent loads)
    constexpr int PER_LANE = (Block::NF + Grp::W - 1) / Grp::W;
    {
      FaceTopo t[PER_LANE];
      Quad<T> pl[PER_LANE];
#if defined(__HIP_DEVICE_COMPILE__)
#pragma unroll
#endif
      for (int j = 0; j < PER_LANE; ++j) {
        const int f = Grp::lane() + j * Grp::W;
        const int fc = f < nf ? f : 0;
        t[j] = m->ft[fc];
        pl[j] = m->fn[fc];
      }
#if defined(__HIP_DEVICE_COMPILE__)
#pragma unroll
#endif
      for (int j = 0; j < PER_LANE; ++j) {
        const int f = Grp::lane() + j * Grp::W;
        const V3<T> v = vw(t[j].vid(0));
        const bool in_play = (f < nf) & ((t[j].flag() & 1) != 0) & (f != closest);
        const bool mark = in_play & !(dot(mk<T>(pl[j].x, pl[j].y, pl[j].z), ww - v) < dummy_precision);
        // set_pass(f, pass) where `mark`; elsewhere a store into the vertex tables, which are initialised just below (no branch per face)
        // (the same lane initialises that byte of the tables right below)
        *(mark ? topo_bytes(m->ft[f < nf ? f : 0], 1) + 3 : start_at + Grp::lane()) = uint8_t(pass);
      }
    }
    for (int v = Grp::lane(); v < 2 * Block::NV; v += Grp::W) start_at[v] = 255;
    if (Grp::lane() == 0) m->top = 0u;
    Grp::sync();
    {
      FaceTopo t[PER_LANE];
#if defined(__HIP_DEVICE_COMPILE__)
#pragma unroll
#endif
      for (int j = 0; j < PER_LANE; ++j) {
        const int f = Grp::lane() + j * Grp::W;
        t[j] = m->ft[f < nf ? f : 0];
      }
#if defined(__HIP_DEVICE_COMPILE__)
#pragma unroll
#endif
      for (int j = 0; j < PER_LANE; ++j) {
        const int f = Grp::lane() + j * Grp::W;
        const bool kept = f < nf && (t[j].flag() & 1) && t[j].pass() != pass;
        const int p0 = m->ft[t[j].adj(0)].pass(), p1 = m->ft[t[j].adj(1)].pass(), p2 = m->ft[t[j].adj(2)].pass();
        if (kept) {
          if (p0 == pass) {
            const uint32_t k = Grp::atomic_inc(&m->top);
            if (k < uint32_t(Block::NF)) m->hz[k] = typename Block::HzT(hz_pack(f, 0));
          }
          if (p1 == pass) {
            const uint32_t k = Grp::atomic_inc(&m->top);
            if (k < uint32_t(Block::NF)) m->hz[k] = typename Block::HzT(hz_pack(f, 1));
          }
          if (p2 == pass) {
            const uint32_t k = Grp::atomic_inc(&m->top);
            if (k < uint32_t(Block::NF)) m->hz[k] = typename Block::HzT(hz_pack(f, 2));
          }
        }
      }
    }
    Grp::sync();
    hz_count = int(m->top);
    if (hz_count > Block::NF) return false;
    // new face of entry k: (a = f[e+1], b = f[e], w): it starts at a and ends at b along the horizon
    for (int k = Grp::lane(); k < hz_count; k += Grp::W) {
      const unsigned fr = m->hz[k];
      const FaceTopo t = m->ft[hz_face(fr)];
      const int e = hz_edge(fr);
      start_at[t.vid((e + 1) % 3)] = uint8_t(k);
      end_at[t.vid(e)] = uint8_t(k);
    }
    Grp::sync();
    int bad = 0;
    for (int k = Grp::lane(); k < hz_count; k += Grp::W) {
      const unsigned fr = m->hz[k];
      const FaceTopo t = m->ft[hz_face(fr)];
      const int e = hz_edge(fr), a = t.vid((e + 1) % 3), b = t.vid(e);
      // the only entry that starts at a / ends at b, and the loop goes on at both ends
      if (start_at[a] != k || end_at[b] != k || start_at[b] == 255 || end_at[a] == 255) bad = 1;
    }
    butterfly_stages<Grp::W>([&](auto stage) {
      constexpr int M = decltype(stage)::value;
      bad |= Grp::template exchange<M>(bad);
    });
    return bad == 0;
  }

  // One polytope expansion by vertex id_w seen from face `closest` (the body of the loop
  // :1261-1280).  Returns true when the hull was updated (valid && horizon >= 3); on false the
  // caller leaves the loop and `status` is what the reference's first failing step sets.
  HFCL_HD bool expand_iteration(int pass, int closest, int id_w) {
    // gjk.cpp:1410-1411: 3*sqrt(DBL_EPSILON) = 4.47e-8 in fp64.  In fp32 the literal formula
    // would give 1e-3 (three orders above the solver tolerance: expand() then keeps faces the new
    // vertex is clearly above), while 4.47e-8 is below fp32 round-off for coplanar supports of
    // flat-faced shapes (degenerate faces).  2e-7 (~1.7 ulp) minimises the mismatch against the
    // fp64 oracle on the cfg2/cfg3/cfg5 sets.
    const T dummy_precision = sizeof(T) == 4 ? T(2e-7) : T(4.470348358154297e-08);
    const V3<T> ww = vw(id_w);
    int hz_count = 0, stop_kind = 0, stop_at = 0;
    bool by_tables = false;  // the new faces find their neighbours through the vertex tables (no order on the horizon)
    if constexpr (PARALLEL_HORIZON) {
#if defined(__HIP_DEVICE_COMPILE__) && defined(HFCL_EPA_PHASE_TWICE) && HFCL_EPA_PHASE_TWICE == 2  // ... the horizon search once more (idempotent)
      {
        int hz2 = 0;
        const bool bt2 = silhouette_parallel(pass, closest, dummy_precision, ww, hz2);
        asm volatile("" ::"v"(hz2), "v"(int(bt2)));
        Grp::sync();
      }
#endif
      by_tables = silhouette_parallel(pass, closest, dummy_precision, ww, hz_count);
      if (!by_tables) {  // not a horizon of simple loops: marks undone, the walk decides
        const int nfu = hw;
        Grp::sync();
        for (int f = Grp::lane(); f < nfu; f += Grp::W)
          if ((m->ft[f].flag() & 1) && m->ft[f].pass() == pass && f != closest) set_pass(f, 0);
        Grp::sync();
        hz_count = 0;
      }
    }
    if (!by_tables) silhouette_walk_fast(pass, closest, dummy_precision, ww, hz_count, stop_kind, stop_at);
    if (hz_count > stock_top) {
      by_tables = false;
      // More horizon edges than free faces: whether (and where) the reference runs out of faces
      // depends on how its pops and pushes interleave -> undo the marks and redo the walk with
      // the stock level tracked (rare: only near the capacity of the face store).
      const int nf = hw;
      Grp::sync();
#if defined(__HIP_DEVICE_COMPILE__)
#pragma unroll 1
#endif
      for (int f = Grp::lane(); f < nf; f += Grp::W)
        if ((m->ft[f].flag() & 1) && m->ft[f].pass() == pass && f != closest) set_pass(f, 0);
      Grp::sync();
      hz_count = 0;
      stop_kind = 0;
      stop_at = 0;
      int level = stock_top;
      for (int j = 0; j < 3 && !stop_kind; ++j)
        silhouette_walk(pass, m->ft[closest].adj(j), m->ft[closest].adje(j), dummy_precision, ww, hz_count, level, stop_kind, stop_at);
      if (stop_kind == EPA_OUT_OF_FACES && cap_iterations < max_iterations) {
        // A block smaller than the reference's face store ran out where the reference still has faces.
        // Nothing but pass marks was touched so far; with them undone the polytope is exactly as at the
        // start of this iteration and the full-capacity tier can redo the iteration from there.
        Grp::sync();
#if defined(__HIP_DEVICE_COMPILE__)
#pragma unroll 1
#endif
        for (int f = Grp::lane(); f < nf; f += Grp::W)
          if ((m->ft[f].flag() & 1) && m->ft[f].pass() == pass) set_pass(f, 0);
        Grp::sync();
        overflow = true;
        resumable = true;
        return false;
      }
    }
    const int n_new = stop_kind ? stop_at : hz_count;
    Grp::sync();
    // 1. the visible faces (pass mark, includes `closest`) leave the hull: normally during the next
    //    closest-face scan; right away only if the stock is too short for the new faces
    if (stock_top < n_new)
      release_visible(pass);
    else
      pending_release = pass;
    // 2. the new faces, one per horizon edge, lanes in parallel; k-th face takes the k-th slot
    //    from the top of the stock and the k-th stamp
    int first_fail = n_new, fail_code = 0, top_slot = hw;
    for (int k = Grp::lane(); k < n_new; k += Grp::W) {
      const unsigned fr = m->hz[k];
      const int f = hz_face(fr), e = hz_edge(fr), e1 = (e + 1) % 3;
      const int nfc = m->stock[stock_top - 1 - k];
      if (nfc + 1 > top_slot) top_slot = nfc + 1;
      const uint32_t fv = m->ft[f].vf;
      const int va = int((fv >> (8 * e1)) & 255u), vb = int((fv >> (8 * e)) & 255u);
      // previous face on the horizon loop: the one before it in walk order, or the one that ends where this one starts
      const int kp = by_tables ? int(reinterpret_cast<const uint8_t*>(m->stack)[Block::NV + va]) : ((k == 0) ? n_new - 1 : k - 1);
      const int pf = m->stock[stock_top - 1 - kp];
      const int fail = face_geometry(nfc, va, vb, id_w, false, stamp + k);
      // bind(nf, 0, f, e); bind(nf, 2, previous, 1)  (:1421-1425, closing bind :1273)
      set_adj(nfc, 0, f, e);
      set_adj(f, e, nfc, 0);
      set_adj(nfc, 2, pf, 1);
      set_adj(pf, 1, nfc, 2);
      if (fail && k < first_fail) {
        first_fail = k;
        fail_code = fail;
      }
    }
    butterfly_stages<Grp::W>([&](auto stage) {
      constexpr int M = decltype(stage)::value;
      const int ok = Grp::template exchange<M>(first_fail), oc = Grp::template exchange<M>(fail_code);
      const int ot = Grp::template exchange<M>(top_slot);
      const bool earlier = ok < first_fail;
      first_fail = earlier ? ok : first_fail;
      fail_code = earlier ? oc : fail_code;
      top_slot = ot > top_slot ? ot : top_slot;
    });
    hw = top_slot;
    stock_top -= n_new;
    hull_count += n_new;
    stamp += n_new;
    Grp::sync();
    if (first_fail < n_new) {  // a face before the walk-level stop failed first
      status = fail_code;
      return false;
    }
    if (stop_kind) {
      status = stop_kind;
      return false;
    }
    return hz_count >= 3;
  }

  // Take over the tetrahedron epa_prepare_tetrahedron described: the state begin() leaves behind for a seed of rank 4.
  // Call after reset(); ends with a sync.  Returns the first closest face.
  HFCL_HD int install(const EpaReady<T>* rb, uint32_t packed) {
    constexpr EpaTetraTopo topo = epa_tetra_topo();
    for (int i = Grp::lane(); i < 8; i += Grp::W) {
      if (i < 4) {
        m->vw[i] = rb->vw[i];
      } else {
        const int f = i - 4;
        m->fn[f] = rb->fn[f];
        const uint32_t flag = 1u | (((packed >> (14 + f)) & 1u) << 1);
        m->ft[f].vf = (f == 0 ? topo.vf[0] : (f == 1 ? topo.vf[1] : (f == 2 ? topo.vf[2] : topo.vf[3]))) | (flag << 24);
        m->ft[f].ap = f == 0 ? topo.ap[0] : (f == 1 ? topo.ap[1] : (f == 2 ? topo.ap[2] : topo.ap[3]));
        m->ft[f].ae = f == 0 ? topo.ae[0] : (f == 1 ? topo.ae[1] : (f == 2 ? topo.ae[2] : topo.ae[3]));
      }
    }
    status = EPA_VALID;
    num_vertices = 4;
    hull_count = 4;
    stock_top -= 4;
    stamp = 4;
    hw = 4;
    Grp::sync();
    return int((packed >> 12) & 3u);
  }
  // ... of a general pair (V0_BLOCK): twelve records -- vertices, their shape-0 support points, face planes
  HFCL_HD int install(const EpaReadyG<T>* rb, uint32_t packed) {
    static_assert(V0M == V0_BLOCK, "general pairs keep their support points in the block");
    constexpr EpaTetraTopo topo = epa_tetra_topo();
    for (int i = Grp::lane(); i < 12; i += Grp::W) {
      if (i < 4) {
        m->vw[i] = rb->vw[i];
      } else if (i < 8) {
        v0p[i - 4] = rb->v0[i - 4];
      } else {
        const int f = i - 8;
        m->fn[f] = rb->fn[f];
        const uint32_t flag = 1u | (((packed >> (2 + f)) & 1u) << 1);
        m->ft[f].vf = (f == 0 ? topo.vf[0] : (f == 1 ? topo.vf[1] : (f == 2 ? topo.vf[2] : topo.vf[3]))) | (flag << 24);
        m->ft[f].ap = f == 0 ? topo.ap[0] : (f == 1 ? topo.ap[1] : (f == 2 ? topo.ap[2] : topo.ap[3]));
        m->ft[f].ae = f == 0 ? topo.ae[0] : (f == 1 ? topo.ae[1] : (f == 2 ? topo.ae[2] : topo.ae[3]));
      }
    }
    status = EPA_VALID;
    num_vertices = 4;
    hull_count = 4;
    stock_top -= 4;
    stamp = 4;
    hw = 4;
    Grp::sync();
    return int(packed & 3u);
  }
  HFCL_HD void loop_out(const EpaLoop<T>& L, EpaLoopOutG<T>& o) const {
    o.status = status;
    o.iterations = L.iterations;
    o.nx = L.outer_n.x; o.ny = L.outer_n.y; o.nz = L.outer_n.z;
    o.depth = L.outer_d;
    const V3<T> a = vw(L.o0), b = vw(L.o1), c = vw(L.o2), a0 = v0(L.o0), b0 = v0(L.o1), c0 = v0(L.o2);
    o.rw[0] = a.x; o.rw[1] = a.y; o.rw[2] = a.z; o.rw[3] = b.x; o.rw[4] = b.y; o.rw[5] = b.z; o.rw[6] = c.x; o.rw[7] = c.y; o.rw[8] = c.z;
    o.r0[0] = a0.x; o.r0[1] = a0.y; o.r0[2] = a0.z; o.r0[3] = b0.x; o.r0[4] = b0.y; o.r0[5] = b0.z; o.r0[6] = c0.x; o.r0[7] = c0.y; o.r0[8] = c0.z;
  }
  // The loop's result with tags instead of shape-0 support points (V0_TAG blocks; resolved by whoever writes the record).
  HFCL_HD void loop_out(const EpaLoop<T>& L, EpaLoopOut<T>& o) const {
    o.status = status;
    o.iterations = L.iterations;
    o.nx = L.outer_n.x; o.ny = L.outer_n.y; o.nz = L.outer_n.z;
    o.depth = L.outer_d;
    const Quad<T> a = m->vw[L.o0], b = m->vw[L.o1], c = m->vw[L.o2];
    o.rw[0] = a.x; o.rw[1] = a.y; o.rw[2] = a.z;
    o.rw[3] = b.x; o.rw[4] = b.y; o.rw[5] = b.z;
    o.rw[6] = c.x; o.rw[7] = c.y; o.rw[8] = c.z;
    o.tag[0] = int(a.w); o.tag[1] = int(b.w); o.tag[2] = int(c.w);
  }

  // GJK::encloseOrigin :437-492 on verts[0..rank) (reference order).  sup(dir) -> (w, w0).  (epa_enclose_origin, below the class, over
  // this block's vertex records.)
  template <class Sup>
  HFCL_HD bool enclose_origin(int& rank, Sup& sup) {
    return epa_enclose_origin<T>(*this, rank, sup);
  }
  // the vertex store epa_enclose_origin works on
  template <class Sup>
  HFCL_HD void eo_support(Sup& sup, const V3<T>& dir, V3<T>& w, V3<T>& w0, int& tag) const { epa_support<TAGGED>(sup, dir, w, w0, tag); }
  HFCL_HD void eo_put(int i, const V3<T>& w, const V3<T>& w0, int tag) {
    Grp::sync();
    set_vert(i, w, w0, tag);
    Grp::sync();
  }

  // EPA::evaluate :1156-1316.  verts[0..rank) must already hold GJK's final simplex in the
  // reference's order (oldest first).  guess = the vector passed as `guess` to evaluate().
  template <class Sup, class Tags = NoTags>
  HFCL_HD void evaluate(int rank, const V3<T>& guess, T ssr_sum, Sup& sup, EpaResult<T>& out, const Tags& tags = Tags()) {
    const int closest0 = begin(rank, guess, sup, out, tags);
    if (closest0 != EPA_NULL) run_loop(closest0, 0, 0, ssr_sum, sup, out, tags);
  }
  // evaluate() up to the loop (:1188-1230): the first closest face, or EPA_NULL when `out` is already
  // final (FallBack :1299-1315).
  template <class Sup, class Tags = NoTags>
  HFCL_HD int begin(int rank, const V3<T>& guess, Sup& sup, EpaResult<T>& out, const Tags& tags = Tags()) {
    const bool enclosed = enclose_origin(rank, sup);
    out.iterations = 0;
    if (rank > 1 && enclosed) {
      status = EPA_VALID;
      num_vertices = 4;
      if (dot(vw(0) - vw(3), cross(vw(1) - vw(3), vw(2) - vw(3))) < T(0)) {
        if constexpr (TAGGED) {  // the tags travel with the records
          const Quad<T> a = m->vw[0], b = m->vw[1];
          Grp::sync();
          m->vw[0] = b;
          m->vw[1] = a;
          Grp::sync();
        } else {
          const V3<T> a = vw(0), a0 = v0(0), b = vw(1), b0 = v0(1);
          Grp::sync();
          set_vert(0, b, b0);
          set_vert(1, a, a0);
          Grp::sync();
        }
      }
      int t0 = new_face(0, 1, 2, true);
      int t1 = new_face(1, 0, 3, true);
      int t2 = new_face(2, 1, 3, true);
      int t3 = new_face(0, 2, 3, true);
      if (hull_count == 4) {
        bind(t0, 0, t1, 0);
        bind(t0, 1, t2, 0);
        bind(t0, 2, t3, 0);
        bind(t1, 1, t3, 2);
        bind(t1, 2, t2, 1);
        bind(t2, 2, t3, 1);
        const int closest0 = find_closest_face();
        status = EPA_VALID;
        return closest0;
      }
    }
    // FallBack :1299-1315
    status = EPA_FALLBACK;
    out.status = status;
    V3<T> n = -guess;
    const T nl = norm(n);
    out.normal = (nl > T(0)) ? (n / nl) : mk<T>(T(1), T(0), T(0));
    out.depth = T(0);
    out.rw0_ = out.rw1_ = out.rw2_ = vw(0);
    out.r00 = out.r01 = out.r02 = v0r(0, tags);
    return EPA_NULL;
  }

  // The expansion loop of evaluate() (:1231-1296) as enter / step / result, so that a kernel can interleave
  // the trips of several polytopes with other work (refilling finished lane groups).  `L` is what the
  // reference keeps in locals across trips: the current best face and the last valid `outer` face.
  HFCL_HD void loop_enter(EpaLoop<T>& L, int closest, int iterations, int pass) const {
    L.closest = closest;
    L.iterations = iterations;
    L.pass = pass;
    L.outer_n = fn(closest);
    L.outer_d = fd(closest);
    const FaceTopo t = m->ft[closest];
    L.o0 = t.vid(0);
    L.o1 = t.vid(1);
    L.o2 = t.vid(2);
  }
  // One trip.  0: go on; 1: the loop is over, `status` is final; 2: the polytope outgrew this block and
  // was prepared for hand-over (overflow && resumable, header written).
  template <class Sup>
  HFCL_HD int step(EpaLoop<T>& L, Sup& sup) {
    // The three ways a trip ends before it starts.  fp32: behind ONE test -- every way out of step() costs the lanes a copy of the loop's state into
    // the registers the caller reads it from, made whether the way is taken or not (k_epa_loop<float,8,17>: 381 -> 347 static v_mov, cfg3 -0.7 %); the
    // fp64 tiers keep the three tests (the same change costs them 1.3 % on cfg5: other live ranges, other copies; profiles/r05_h section 5).
    constexpr bool ONE_TEST = sizeof(T) == 4;
    if constexpr (ONE_TEST) {
      const int early = !(L.iterations < max_iterations) ? 1 : ((L.iterations >= cap_iterations && cap_iterations < max_iterations) ? 2 : (num_vertices >= max_iterations + 4 ? 3 : 0));
      if (early) {
        if (early == 2) {
          overflow = true;
          resumable = true;
          if (Grp::lane() == 0) m->hdr = EpaHeader{L.closest, L.iterations, L.pass, status, num_vertices, hull_count, stock_top, stamp, hw};
          Grp::sync();
          return 2;
        }
        status = early == 1 ? EPA_FAILED : EPA_OUT_OF_VERTICES;
        return 1;
      }
    } else {
      if (!(L.iterations < max_iterations)) {
        status = EPA_FAILED;
        return 1;
      }
      if (L.iterations >= cap_iterations && cap_iterations < max_iterations) {
        // capacity of this scratch block reached before the reference's limit: hand over
        overflow = true;
        resumable = true;
        if (Grp::lane() == 0) m->hdr = EpaHeader{L.closest, L.iterations, L.pass, status, num_vertices, hull_count, stock_top, stamp, hw};
        Grp::sync();
        return 2;
      }
      if (num_vertices >= max_iterations + 4) {
        status = EPA_OUT_OF_VERTICES;
        return 1;
      }
    }
    const int closest = L.closest;
    const int iw = num_vertices++;
    set_pass(closest, ++L.pass);
    const V3<T> cn = fn(closest);
    V3<T> w, w0;
    int tag;
    epa_support<TAGGED>(sup, cn, w, w0, tag);
#if defined(__HIP_DEVICE_COMPILE__) && defined(HFCL_EPA_PHASE_TWICE) && HFCL_EPA_PHASE_TWICE == 1  // phase-cost experiment (tools/epa_phase_costs.sh): the support once more
    {
      V3<T> cn2 = cn, w2, w02;
      int tag2;
      asm volatile("" : "+v"(cn2.x));
      epa_support<TAGGED>(sup, cn2, w2, w02, tag2);
      asm volatile("" ::"v"(w2.x), "v"(w2.y), "v"(w2.z), "v"(tag2));
    }
#endif
    Grp::sync();
    set_vert(iw, w, w0, tag);
    Grp::sync();
    const FaceTopo tcl = m->ft[closest];
    const V3<T> vf1 = vw(tcl.vid(0)), vf2 = vw(tcl.vid(1)), vf3 = vw(tcl.vid(2));
    const T fdist = dot(cn, w - vf1);
    const T wnorm = norm(w);
    const T thr = tolerance + tolerance * wnorm;
    if constexpr (ONE_TEST) {
      // (all four evaluated, no short circuit: three more norms cost less than the state copies in front of three more branches)
      const bool reached = (fdist <= thr) | (norm(w - vf1) <= thr) | (norm(w - vf2) <= thr) | (norm(w - vf3) <= thr);
      if (reached) {
        status = EPA_ACCURACY_REACHED;
        return 1;
      }
    } else {
      if (fdist <= thr) {
        status = EPA_ACCURACY_REACHED;
        return 1;
      }
      if (norm(w - vf1) <= thr || norm(w - vf2) <= thr || norm(w - vf3) <= thr) {
        status = EPA_ACCURACY_REACHED;
        return 1;
      }
    }
    if (!expand_iteration(L.pass, closest, iw)) {
      if (resumable) {  // hand over as of the start of this iteration (vertex iw is recomputed there)
        --num_vertices;
        --L.pass;
        if (Grp::lane() == 0) m->hdr = EpaHeader{closest, L.iterations, L.pass, status, num_vertices, hull_count, stock_top, stamp, hw};
        Grp::sync();
        return 2;
      }
      return 1;
    }
#if defined(__HIP_DEVICE_COMPILE__) && defined(HFCL_EPA_PHASE_TWICE) && HFCL_EPA_PHASE_TWICE == 4  // ... the closest-face scan once more (the second without releases)
    {
      const int c2 = find_closest_face();
      asm volatile("" ::"v"(c2));
      pending_release = -1;
      const int c3 = find_closest_face();
      loop_enter(L, c2 == c3 ? c2 : c3, L.iterations + 1, L.pass);
      return 0;
    }
#endif
    loop_enter(L, find_closest_face(), L.iterations + 1, L.pass);
    return 0;
  }
  template <class Tags = NoTags>
  HFCL_HD void loop_result(const EpaLoop<T>& L, T ssr_sum, EpaResult<T>& out, const Tags& tags = Tags()) const {
    out.status = status;
    out.iterations = L.iterations;
    out.normal = L.outer_n;
    out.depth = L.outer_d + ssr_sum;
    out.rw0_ = vw(L.o0); out.rw1_ = vw(L.o1); out.rw2_ = vw(L.o2);
    out.r00 = v0r(L.o0, tags); out.r01 = v0r(L.o1, tags); out.r02 = v0r(L.o2, tags);
  }
  // The whole loop from (closest, iterations, pass): evaluate() enters at iteration 0, a polytope handed
  // over by a smaller tier where it stopped.  Nothing is written to `out` on a hand-over.
  template <class Sup, class Tags = NoTags>
  HFCL_HD void run_loop(int closest, int iterations, int pass, T ssr_sum, Sup& sup, EpaResult<T>& out, const Tags& tags = Tags()) {
    EpaLoop<T> L;
    loop_enter(L, closest, iterations, pass);
    int r;
    while ((r = step(L, sup)) == 0) {
    }
    if (r == 1) loop_result(L, ssr_sum, out, tags);
  }

  // Continue a polytope saved by a tier with capacity CAP_SRC in this (already reset()) block: vertices and
  // faces keep their indices, the extra faces of the larger block join the stock.
  // (a V0_TAG block continues a V0_TAG block: the vertex records keep their tags and the saved coordinates are not read)
  template <int CAP_SRC>
  HFCL_HD EpaHeader load(const EpaSaved<T, CAP_SRC>* saved) {
    typedef EpaScratch<T, CAP_SRC, V0_TAG> Src;
    const Src* src = &saved->blk;
    const EpaHeader h = src->hdr;
#if defined(__HIP_DEVICE_COMPILE__)
#pragma unroll 1
#endif
    for (int i = Grp::lane(); i < h.num_vertices; i += Grp::W) {
      const Quad<T> q = src->vw[i];
      if constexpr (V0M == V0_TAG) {
        m->vw[i] = q;
      } else {
        m->vw[i] = Quad<T>{q.x, q.y, q.z, T(0)};
        v0p[i] = saved->v0[i];
      }
    }
#if defined(__HIP_DEVICE_COMPILE__)
#pragma unroll 1
#endif
    for (int f = Grp::lane(); f < Src::NF; f += Grp::W) {
      m->fn[f] = src->fn[f];
      m->ft[f] = src->ft[f];
    }
    // the extra slots of the larger block go UNDER the saved stock, lowest index on top of them: they are used once the
    // polytope's own free slots are, in increasing order, so that the high-water mark keeps growing slowly
    const int extra = (2 * cap_iterations + 4) - Src::NF;
    for (int i = Grp::lane(); i < extra; i += Grp::W) m->stock[i] = uint8_t(Src::NF + extra - 1 - i);
    for (int i = Grp::lane(); i < h.stock_top; i += Grp::W) m->stock[extra + i] = src->stock[i];
    status = h.status;
    num_vertices = h.num_vertices;
    hull_count = h.hull_count;
    stock_top = h.stock_top + extra;
    stamp = h.stamp;
    hw = h.hw;
    Grp::sync();
    return h;
  }
};

// Group-cooperative copy of a resumable polytope (scratch block incl. its header) to `dst`, with the shape-0 support
// points of its vertices written out as coordinates (`tags`: int -> V3, V0_TAG blocks only; V0_BLOCK blocks carry them).
// KEEP_TAGS (V0_TAG blocks): the continuing tier works with tags as well -- nothing but the block is written.
template <typename T, class Grp, int CAP, int V0M, class Tags = NoTags, bool KEEP_TAGS = false>
HFCL_HD void epa_save_block(const EpaScratch<T, CAP, V0M>* block, EpaSaved<T, CAP>* dst, const Tags& tags = Tags()) {
  static_assert(V0M != V0_EXTERN, "only the fast tiers hand polytopes over");
  Grp::sync();
  typedef EpaScratch<T, CAP, V0_TAG> Tail;
  static_assert(sizeof(Tail) % 4 == 0, "block is copied in 4-byte words");
  static_assert(sizeof(EpaScratch<T, CAP, V0M>) - sizeof(Tail) == (V0M == V0_BLOCK ? sizeof(Quad<T>) * (CAP + 4) : 0),
                "the members after the support points are laid out as in the V0_TAG block");
  const uint32_t* src = reinterpret_cast<const uint32_t*>(&block->vw[0]);
  uint32_t* d = reinterpret_cast<uint32_t*>(&dst->blk);
  // rolled on purpose: this is the rare path and must not cost the expansion loop registers
#if defined(__HIP_DEVICE_COMPILE__)
#pragma unroll 1
#endif
  for (int i = Grp::lane(); i < int(sizeof(Tail) / 4); i += Grp::W) d[i] = src[i];
  if constexpr (!KEEP_TAGS) {
    const int nv = block->hdr.num_vertices;  // the records beyond hold whatever the LDS held before: no tags to resolve
#if defined(__HIP_DEVICE_COMPILE__)
#pragma unroll 1
#endif
    for (int i = Grp::lane(); i < nv; i += Grp::W) {
      if constexpr (V0M == V0_TAG) {
        const V3<T> p = tags(int(block->vw[i].w));
        dst->v0[i] = Quad<T>{p.x, p.y, p.z, T(0)};
      } else {
        dst->v0[i] = block->v0[i];
      }
    }
  }
}

// EPA::getWitnessPointsAndNormal (:1451-1466) + inflate, shape-0 frame
template <typename T>
HFCL_HD void epa_witness_normal(const EpaResult<T>& r, T r0, T r1, V3<T>& w0, V3<T>& w1, V3<T>& normal) {
  closest_points(3, r.rw0_, r.rw1_, r.rw2_, r.r00, r.r01, r.r02, r.r00 - r.rw0_, r.r01 - r.rw1_, r.r02 - r.rw2_, w0, w1);
  if (norm(w0 - w1) > Lim<T>::dummy()) {
    normal = (r.depth >= T(0)) ? normalized(w0 - w1) : normalized(w1 - w0);
  } else {
    normal = r.normal;
  }
  if (r0 > T(0)) w0 = w0 + r0 * normal;
  if (r1 > T(0)) w1 = w1 - r1 * normal;
}

}  // namespace hfcl
