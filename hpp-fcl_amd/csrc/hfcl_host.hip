// hfcl_host.hip -- host side: library object, batch pipeline and the C-ABI implementation of
// include/hppfcl_amd.h (the kernels live in hfcl_k_gjk.hip / hfcl_k_epa.hip / hfcl_k_bvh.hip, see hfcl_launch.hpp).  No CPU fallback anywhere in this file: every compute entry point
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
//                    explicit DFS stack in LDS (reference order), OBB SAT / RSS bounds, triangle-triangle leaves;
//                    a walk past its step budget is continued by a wave, 64 stack entries per trip
//                    (k_bvh_coop<T> / k_bvh_distance_coop<T>)
//   k_bvh_collide<T,SOLID> + k_shape_obb<T> + k_bvh_shape_coop<T> + k_bvh_shape_finish<T> (collide, first contact) and
//   k_bvh_shape_distance_lane<T> + k_shape_obbrss<T> + k_bvh_shape_distance_coop<T> (distance)
//                    BVHModel<OBBRSS> x convex solid or Plane/Halfspace the same way: one query per lane, per-lane GJK
//                    leaves, the leaves that need EPA from a queue
//   k_bvh_shape<T> / k_bvh_shape_distance<T>   ... one query per 16-lane group, sequential traversal, leaves =
//                    TriangleP-vs-solid GJK + EPA in LDS: requests that keep walking after a contact
//   k_unsupported<T> flags the pairs of a bucket the engine cannot evaluate (never computed elsewhere)
#include <algorithm>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <condition_variable>
#include <map>
#include <mutex>
#include <thread>
#include <string>
#include <vector>

#include "hfcl_dev.hpp"
#include "hfcl_launch.hpp"

// =======================================================================================
// Host side: library object + C ABI
// =======================================================================================
static thread_local std::string g_last_error;
static void set_error(const std::string& s) { g_last_error = s; }
void hfcl_internal_set_error(const char* msg) { g_last_error = msg ? msg : ""; }  // hfcl_multi.hip: a worker thread's message handed to the caller's

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
  std::vector<uint8_t> h_kinds;  // host copy of d_kinds: small host batches are classified on the host (host_batch)
  DShape<double>* d_shapes64 = nullptr;
  DShape<float>* d_shapes32 = nullptr;
  double* d_verts64 = nullptr;
  float* d_verts32 = nullptr;
  uint8_t* d_kinds = nullptr;
  // vertex adjacency of convex shapes (hfcl_lib_set_convex_neighbors): host copies per shape, device image built lazily
  std::vector<double> h_verts;
  struct HostGraph { std::vector<uint32_t> off, ids; };
  std::map<uint32_t, HostGraph> h_graphs;
  bool graph_dirty = false;
  uint32_t* d_graph_base = nullptr;
  std::vector<void*> graph_retired;   // adjacency images replaced by a later registration: batches in flight on other streams may still read
                                      // them, so they are freed where the library waits for the device anyway (set_shapes, destroy)
  hipStream_t upload_stream = nullptr;  // non-blocking stream of the adjacency upload (the host waits for it alone)
  uint32_t* d_graph_off = nullptr;
  NbrEntry<float>* d_graph_ent32 = nullptr;
  NbrEntry<double>* d_graph_ent64 = nullptr;
  uint32_t climb_min = HFCL_CLIMB_MIN;  // HFCL_CLIMB_MIN: hulls of at least this many vertices with a graph hill-climb
  // workspace (grown on demand)
  size_t ws_capacity = 0;  // pairs
  size_t epa_capacity = 0;  // pairs the EPA queues / hand-over area are sized for (0: not allocated yet)
  uint32_t* d_lists = nullptr;
  uint32_t* d_counts = nullptr;
  void* d_epa_queue = nullptr;
  void* d_epa_queue2 = nullptr;
  uint32_t* d_epa_cc_over = nullptr;  // Work::epa_cc_over (resume_cap entries)
  hipStream_t aux = nullptr;     // k_epa_records runs here, beside the tiers that continue the handed-over polytopes
  hipStream_t mesh_st = nullptr;  // the mesh walks of a mixed library's batch run here, beside the solids' kernels (option mesh_beside)
  hipEvent_t ev_mesh_fork = nullptr, ev_mesh_join = nullptr;
  // ... the mesh x mesh walks of such a batch beside its mesh x solid walks (tables of their own: d_bvh2_*), and the helper stream the
  // mesh x solid walks use instead of `aux` (which the solids' EPA section, now beside them, uses)
  hipStream_t mesh_st2 = nullptr, mesh_aux = nullptr;
  bool ran_batch = false;  // h_counts holds the bucket counts of this library's last batch (once its copy has landed)
  hipEvent_t ev_mesh_fork2 = nullptr, ev_mesh_join2 = nullptr;
  BvhTask* d_bvh2_tasks = nullptr;
  void* d_bvh2_sums = nullptr;
  uint32_t* d_bvh2_susp = nullptr;
  uint32_t* d_bvh2_ctr = nullptr;
  size_t bvh2_split_n = 0, bvh2_split_cap = 0;
  // 0 in line; 1 the mesh walks beside the solids' GJK kernels; 2 also mesh x mesh beside mesh x solid, the solids' EPA section beside both
  // (1, 2: when the library's last batch held meshes and solids); 4: as 2 whatever the last batch held
  uint32_t mesh_beside = 2;
  bool mesh_prio = false;  // option mesh_prio = 1: the streams of the mesh x solid walks at the device's highest priority (read when they are created)
  // the solids' GJK kernels of a small batch run beside each other on these (option gjk_beside_max): one bucket's kernel does not fill the chip
  hipStream_t gjk_st[3] = {nullptr, nullptr, nullptr};
  hipEvent_t gjk_fork = nullptr, gjk_join[3] = {nullptr, nullptr, nullptr};
  uint32_t epa_direct_max = 4096;    // largest batch whose EPA seeds all go to the full-capacity tier, the fast tiers not launched (0: never)
  uint32_t gjk_beside_max = 120000;  // largest batch whose GJK kernels fan out (0: never; below the size from which batches run as two halves)
  hipStream_t walk_st[WALK_ROUNDS - 1] = {};  // mesh x mesh collide(): the continuation of what round r of the walk hands over runs on walk_st[r]
  hipEvent_t walk_fork[WALK_ROUNDS - 1] = {}, walk_join[WALK_ROUNDS - 1] = {};
  hipEvent_t ev_aux0 = nullptr, ev_aux1 = nullptr, ev_aux2 = nullptr, ev_aux3 = nullptr;  // fork / join of the EPA tail; of k_bvh_shape_finish's first half
  void* d_epa_ready = nullptr;   // EpaReady<float>[epa_ready_capacity]: the staged convex x convex fast tier (k_epa_prepare / k_epa_loop / k_epa_records)
  size_t epa_ready_capacity = 0;
  void* d_epa_ready_g = nullptr; // EpaReadyG<T>[ws_capacity]: the staged fast tier of the general queues (both precisions)
  size_t epa_ready_g_bytes = 0;
  bool epa_general_staged = false;       // HFCL_EPA_GENERAL_STAGED=1: prepare / loop / records for the general queues too.  Byte-identical
                                         // records; measured slower in wall-clock on cfg2 (0.416 -> 0.479 ms) and cfg5 unsplit (4.77 -> 5.30 ms
                                         // per 1M mixed pairs), faster only on cfg5 split (5.34 -> 5.13): profiles/r05_e_general_staged.md
  size_t epa_general_staged_min = 32768; // HFCL_EPA_GENERAL_STAGED_MIN
  bool records_aside = true;     // HFCL_EPA_RECORDS_ASIDE=0: k_epa_records on the batch's stream
  bool epa64_two_streams = true; // HFCL_EPA64_TWO_STREAMS=0: the two fp64 fast-tier kernels one after the other
  bool epa_cc_staged = true;     // HFCL_EPA_CC_STAGED=0: the one-kernel form (k_epa_stream<.., CC>)
  size_t epa_cc_staged_min = 32768;  // ... which batches below this many pairs keep (two launches less); HFCL_EPA_CC_STAGED_MIN
  void* d_epa_resume = nullptr;
  void* d_epa_v0 = nullptr;
  size_t resume_cap = 0;
  bool shape_finish_tiers = true;  // HFCL_SHAPE_FINISH_TIERS=0: k_bvh_shape_finish in one launch at full capacity
  bool shape_finish_aside = true;  // HFCL_SHAPE_FINISH_ASIDE=0: all of k_bvh_shape_finish behind the last launch of k_bvh_shape_coop
  void* d_shape_defer = nullptr;  // ShapeDeferItem<double>[shape_defer_capacity]: EPA queue of the one-query-per-lane mesh x solid form
  void* d_shape_oq = nullptr;     // ObbQuery<double>[shape_oq_capacity]: the solids' OBBs against the mesh poses, by pair
  size_t shape_defer_capacity = 0, shape_oq_capacity = 0;
  bool bvh_shape_lane = true;     // HFCL_BVH_SHAPE_LANE=0: the group kernels for every request (A/B switch)
  // step budgets of the one-query-per-lane mesh x solid walk (a unit suspends into tasks when it has taken that many BV-test
  // equivalents; a GJK leaf counts shape_leaf_cost): the queries themselves / their tasks.  The steps per query have a heavy
  // tail whatever the batch size (median 1, mean ~60, maximum > 3000 steps with > 1000 leaves), so the walk is always split.
  uint32_t shape_budget0 = 128, shape_budget = 96, shape_leaf_cost = 32, shape_levels = BVH_MAX_LEVELS;
  // Suspended queries are continued by k_bvh_shape_coop (a wave per query, 64 stack entries per trip) instead of task levels
  // (HFCL_SHAPE_COOP=0: the levels); the queries' own budget is then 16 steps (100k queries per kind, budgets 8 / 16 / 32 / 128:
  // sphere 2.0 / 2.0 / 2.4 / 2.6 ms, ellipsoid 7.5 / 8.1 / 8.4 / 8.3, box 1.4 / 1.3 / 1.2 / 1.1; profiles/r03_i)
  bool shape_coop = true;
  // ... and a walk such a kernel has worked on for this many clock ticks is cut into chunk tasks for its next launch (BvhSplit::cut_ticks;
  // HFCL_BVH_CUT_TICKS / HFCL_SHAPE_CUT_TICKS; 0: never).  The records equal the uncut walks' in every field wherever the cuts fall
  // (tools/cut_check.py).  mesh x solid, 600 000 ticks (~2.3x the mean walk): 100k mixed queries 4.4 -> 3.7 ms on one box, the single
  // kinds within +-5 %; shorter budgets lose (200 000: 3.5 against 3.3 at 400 000, 100 000: 6.8 ms -- every cut walks the chunks
  // behind a contact for nothing and pays three launches).  mesh x mesh: off -- its waves are busy 79 % of the kernel's time already
  // and cfg4 went 2.97 -> 3.18 ms (profiles/r04_j)
  uint32_t bvh_cut_ticks = 0, shape_cut_ticks = 350000;  // (600 000 until the queries' own phase became three kernels: profiles/r06_g section 5)
  uint32_t shape_budget0_coop = 16;
  // Mesh x mesh queries past their step budget are continued by k_bvh_coop (a wave per query, 64 stack entries per trip)
  // instead of task levels (HFCL_BVH_COOP=0: the levels).  cfg4, budgets 160 / 192 / 256 / 320 / 384: 100k queries 3.82 / 3.53 /
  // 3.28 / 3.39 / 3.57 ms (levels: 5.03); 1M queries, 256 / 512 / 640 / 1024: 15.1 / 10.95 / 10.86 / 11.9 ms (unsplit stream:
  // 14.7); 250k: 5.03 ms (8.78); 10k: 2.63 (3.74) -- profiles/r03_k.  HFCL_BVH_BUDGET0_COOP overrides both.
  bool bvh_coop = true;
  uint32_t bvh_budget0_coop = 0;  // 0: 256 steps up to 500k queries, 640 beyond
  // distance(): a mesh x mesh walk that has taken this many steps is continued by a wave (k_bvh_distance_coop); 0: never
  uint32_t bvhd_budget = 64;      // HFCL_BVHD_BUDGET: steps a lane walks before its walk goes to k_bvh_distance_pool (cfg4d 100k queries, budgets 16 / 64 / 256: 34.4 / 33.2 / 34.1 ms, profiles/r04_c; with the wave-per-walk form of round 3, HFCL_BVHD_POOL=0, 1024 was best: 55.7 ms)
  uint32_t bvhd_pool = 1;         // HFCL_BVHD_POOL: the walks past the budget are continued by k_bvh_distance_pool (0: k_bvh_distance_coop)
  uint32_t bvhd_pool_leaf_min = 24, bvhd_pool_starve = 32, bvhd_pool_part_min = 48;  // HFCL_BVHD_LEAF_MIN / HFCL_BVHD_STARVE / HFCL_BVHD_PART_MIN
  void* d_dist_susp = nullptr;    // DistSusp<double>[dist_susp_capacity]
  size_t dist_susp_capacity = 0;
  uint32_t pool_rerun = 1;          // HFCL_POOL_RERUN: pooled distance() walks whose result could hang on a rounding error are walked again in order (0: never -- the round-5 behaviour; 2: every walk, a test of the ordered mode)
  uint32_t shape_dist_pool = 1;     // HFCL_SHAPE_DIST_POOL: mesh x solid distance() walks past the budget continue in k_bvh_shape_distance_pool (0: k_bvh_shape_distance_coop)
  uint32_t shape_dist_leaf_min = 48, shape_dist_starve = 16;  // HFCL_SHAPE_DIST_LEAF_MIN / HFCL_SHAPE_DIST_STARVE (a GJK pass is worth waiting for: profiles/r04_i)
  uint32_t shape_dist_budget = 64;  // HFCL_SHAPE_DIST_BUDGET: the same for mesh x solid (a GJK leaf counts 16 steps; k_bvh_shape_distance_coop)
  void* d_shape_dist_susp = nullptr;
  size_t shape_dist_susp_capacity = 0;
  // host-call staging: PIPE_SLOTS device buffer sets of `st_capacity` pairs each (a chunk of a host batch), three streams
  // (H2D | kernels | D2H) and per-slot events / pinned counter blocks (host_batch)
  static constexpr int PIPE_SLOTS = 6;  // (three left the feeder waiting for records to leave: profiles/r03_c)
  struct Staging {
    uint32_t *d_s1 = nullptr, *d_s2 = nullptr;
    double *d_tf1 = nullptr, *d_tf2 = nullptr;
    double *d_qt1 = nullptr, *d_qt2 = nullptr;  // compact host poses (7 doubles), expanded into d_tf1/2 on the device
    hfcl_result* d_out = nullptr;
    hfcl_guess *d_gin = nullptr, *d_gout = nullptr;
    hipEvent_t ev_in = nullptr, ev_in2 = nullptr, ev_done = nullptr;  // inputs of object 1 / object 2 arrived, kernels done
    uint32_t* h_counts = nullptr;  // pinned: bucket populations of the chunk that last ran in this slot
    uint32_t* h_counts2 = nullptr; // ... of its second half when the chunk ran split
    bool split = false;
  };
  Staging stage[PIPE_SLOTS];
  size_t st_capacity = 0;
  // small host batches (<= SMALL_MAX pairs): every input array packed into one pinned block and one device block (one
  // copy in, one copy out, one stream) instead of five copies and the pipeline's threads
  static constexpr size_t SMALL_MAX = 4096;
  char* h_pack = nullptr;  // pinned
  char* d_pack = nullptr;
  uint32_t* h_pack_counts = nullptr;  // pinned: bucket populations of a small batch (and of its second half: never split)
  hipStream_t s_h2d = nullptr, s_h2d2 = nullptr, s_cmp = nullptr, s_d2h = nullptr;
  uint32_t acc_counts[N_COUNTERS] = {0};  // host batches: bucket populations summed over the chunks
  bool last_host = false;                 // the last call was a host batch: acc_counts are its populations
  bool in_host_batch = false;
  size_t pipe_chunk = 0;                  // pairs per chunk (0 = automatic); HFCL_PIPE_CHUNK / hfcl_lib_set_host_chunk
  uint32_t* counts_dst = nullptr;         // where run_batch_one sends the bucket populations (default: h_counts)
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
  bool has_flats = true;             // some shape is a Plane / Halfspace: their "very rough" volumes are no lower bounds, so what a mesh walk
                                     // against them reports depends on the ORDER of its visits -- the ordered continuation, not the pool
  bool has_curved = true;            // some shape is an Ellipsoid / Cone / Cylinder: the curved class of the fp64 EPA tiers can occur
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
  DNodeF* d_fnodes = nullptr;  // 64-byte records of the fp32 separating-axis filter (fp64 collide)
  bool bvh_filter = false;     // HFCL_BVH_FILTER=1: the fp32 filter form of k_bvh_collide (exact, measured slower: profiles/r03_b)
  DRss<double>* d_rss64 = nullptr;
  DRss<float>* d_rss32 = nullptr;
  DNodeD<double>* d_dnodes64 = nullptr;  // the distance() walk's packed node records
  DNodeD<float>* d_dnodes32 = nullptr;
  double* d_bverts64 = nullptr;
  float* d_bverts32 = nullptr;
  uint32_t* d_btris = nullptr;
  DMesh* d_meshes = nullptr;
  // models too large (> 65535 nodes) or too deep for the LDS stacks: 32-bit node ids + per-lane global slabs (BvhSpill)
  uint32_t bvh_max_depth = 0;
  size_t bvh_max_nodes = 0;
  void* d_bvh_slab = nullptr;
  size_t bvh_slab_bytes = 0;
  // split mesh x mesh traversals (BvhSplit): task table, unit summaries, suspended-query list, counters
  BvhTask* d_bvh_tasks = nullptr;
  uint32_t* d_bvh_cut_words = nullptr;  // BvhSplit::cut_words / cut_vals (bvh_split_cap entries each)
  double* d_bvh_cut_vals = nullptr;
  void* d_bvh_sums = nullptr;
  uint32_t* d_bvh_susp = nullptr;
  uint32_t* d_bvh_ctr = nullptr;
  // mesh x mesh collide() in walk / leaves / resolve rounds (hfcl_dev.hpp: WalkRec): records, item list, leaf results, the two query lists, counters
  void* d_walk_recs = nullptr;
  uint32_t* d_walk_items = nullptr;
  void* d_walk_res = nullptr;
  uint32_t* d_walk_lists = nullptr;
  uint32_t* d_walk_order = nullptr;  // BvhSplit::order (one entry per suspended slot)
  bool walk_order = true;            // option bvh_walk_order: the continuation launches draw the queries with the most stack entries first
  uint32_t* d_walk_ctr = nullptr;
  size_t walk_n = 0;
  // ... of the mesh x solid walks (their own: the two kinds of walks of a mixed batch run beside each other); lists: [2 n] + the redo list [n]
  void* d_swalk_recs = nullptr;
  uint32_t* d_swalk_items = nullptr;
  void* d_swalk_res = nullptr;
  uint32_t* d_swalk_lists = nullptr;
  uint32_t* d_swalk_ctr = nullptr;   // 8 words per round, then the 64 words of WalkArgs::hist
  uint32_t* d_swalk_perm = nullptr;  // WalkArgs::perm
  size_t swalk_n = 0;
  bool shape_walk_sort = true;       // option shape_walk_sort = 0: the listed leaves evaluated in the order the walks listed them
  bool shape_walk = true;            // option shape_walk = 0: the queries' own phase of mesh x solid collide() by k_bvh_collide's SOLID form (walk and leaves in one kernel)
  uint32_t shape_walk_budget = 256;  // box tests a query's walk may take before k_bvh_shape_coop continues it
  uint32_t shape_walk_min = 65536;   // batch size from which that phase is used (its eight launches are 0.2 ms of latency: 20k queries 1.63 against 1.42 ms,
                                     // 50k 1.96 / 1.82, 100k 2.43 / 2.70, 200k 3.55 / 4.16)
  size_t epa_resume_slots = 0, bvh_task_slots = 0;  // options epa_resume_slots / bvh_task_slots (0: sized by the batch)
  bool bvh_force_wide = false, pipe_trace = false;   // options bvh_force_wide / pipe_trace
  bool walk_early_coop = true;                               // HFCL_BVH_WALK_EARLY_COOP: the queries round 0 hands over are continued beside the later rounds
  // Rounds of walk / leaves / resolve (option bvh_walk_rounds; 0: k_bvh_collide walks the queries, leaves inline).  Not set: chosen per
  // batch -- ONE round of up to 16 listed leaves and 256 box tests (320 from 120k queries), then the continuation, up to 220k queries
  // (100k: 1.86 against 1.94 ms with two rounds, 20k: 1.20 against 1.78, 50k: 1.46 against 1.74: a round is as long as its longest lane,
  // and with the node records kept the lanes carry the walks far enough in one); TWO rounds (6 then 16 leaves; 320 / 640 then 256 box tests)
  // with the first round's hand-overs continued beside the second above that (250k: 3.32 against 3.36-3.58 ms, 400k: 4.78 against 4.92,
  // 1M: 9.2 against 10.0 with one round).  profiles/r06_a section 6
  bool walk_auto = true;
  uint32_t walk_rounds = 2;
  uint32_t walk_k[WALK_ROUNDS] = {6, 16, 16, 16};           // option bvh_walk_k: leaves a walk lists per round (setting it switches the automatic choice off)
  uint32_t walk_budget[WALK_ROUNDS] = {224, 256, 512, 512};  // option bvh_walk_budget: box tests per round from round 1 on before the walk goes to k_bvh_coop
  size_t bvh_split_n = 0, bvh_split_cap = 0;
  uint32_t bvh_budget0 = HFCL_BVH_BUDGET0;  // HFCL_BVH_BUDGET0: step budget of the queries (level 0); bvh_budget: of the tasks
  // No budget given by the environment: chosen per batch.  A batch that does not fill the chip's lanes more than ~1.5
  // times is a walk with one query per lane whose waves run on with most lanes finished; there the queries are cut at
  // 512 steps and their remainders spread over levels of small tasks (profiles/r02_w: 100k queries 6.4 -> 5.1 ms,
  // 10k 4.7 -> 3.4 ms).  A larger batch keeps its lanes busy by refilling and loses with the split (1M: 68 -> 51 M q/s).
  bool bvh_auto = true;
  uint32_t bvh_budget = HFCL_BVH_BUDGET, bvh_levels = HFCL_BVH_LEVELS;  // HFCL_BVH_BUDGET / HFCL_BVH_LEVELS (1: unsplit)
  // contact list of the last hfcl_collide_batch_contacts call
  hfcl_contact* d_contacts = nullptr;
  size_t contacts_cap = 0;
  uint32_t* d_contacts_count = nullptr;
  BvhParams bvh_params = {1u, nullptr, 0u, nullptr};
  double break_distance = 1e-3;
};

static void share_tables(hfcl_lib* h, const hfcl_lib* lib);
static void free_retired_graphs(hfcl_lib* lib);

// bucket population i of the last batch (both halves of a split batch)
static uint32_t one_count(const uint32_t* c, int i) {
  // the EPA queue of a batch = the general queue + the fp32 convex x convex queue
  // a bucket = its bottom part + its curved part; the EPA queue of a batch = the general queue + the second queue
  return c[i] + (i < B_COUNT ? c[B_CURVED0 + i] : 0u) + (i == B_COUNT ? c[B_COUNT + 3] : 0u);
}
static uint32_t total_count(const hfcl_lib* lib, int i) {
  if (lib->last_host) return one_count(lib->acc_counts, i);
  uint32_t c = lib->h_counts ? one_count(lib->h_counts, i) : 0u;
  if (lib->last_split && lib->helper && lib->helper->h_counts) c += one_count(lib->helper->h_counts, i);
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
int hfcl_has_ab_forms(void) { return HFCL_KEEP_AB_FORMS ? 1 : 0; }
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

static bool validate_shapes(const char* who, const hfcl_shape* shapes, size_t n_shapes, const double* vertices, size_t n_vertices) {
  if (!shapes || n_shapes == 0) {
    set_error(std::string(who) + ": empty shape table");
    return false;
  }
  for (size_t i = 0; i < n_shapes; ++i) {
    const hfcl_shape& s = shapes[i];
    const bool ok_kind = s.type == HFCL_GEOM_BOX || s.type == HFCL_GEOM_SPHERE || s.type == HFCL_GEOM_CAPSULE ||
                         s.type == HFCL_GEOM_ELLIPSOID || s.type == HFCL_GEOM_CONVEX || s.type == HFCL_BV_OBBRSS ||
                         s.type == HFCL_GEOM_TRIANGLE || s.type == HFCL_GEOM_CONE || s.type == HFCL_GEOM_CYLINDER ||
                         s.type == HFCL_GEOM_PLANE || s.type == HFCL_GEOM_HALFSPACE;
    if (!ok_kind) {
      set_error(std::string(who) + ": unsupported shape type " + std::to_string(s.type));
      return false;
    }
    if (s.type == HFCL_GEOM_CONVEX) {
      if (s.num_points == 0 || s.num_points > (uint32_t)HULL_LARGE_MAX) {
        set_error(std::string(who) + ": convex shapes must have 1.." + std::to_string(HULL_LARGE_MAX) + " vertices; got " +
                  std::to_string(s.num_points));
        return false;
      }
      if (size_t(s.vertex_offset) + s.num_points > n_vertices) {
        set_error(std::string(who) + ": convex vertex range out of bounds");
        return false;
      }
    }
    if (s.type == HFCL_GEOM_TRIANGLE && size_t(s.vertex_offset) + 3 > n_vertices) {  // its corners are 3 vertices of the array
      set_error(std::string(who) + ": TriangleP vertex range out of bounds");
      return false;
    }
    if ((s.type == HFCL_GEOM_CONVEX || s.type == HFCL_GEOM_TRIANGLE) && !vertices) {
      set_error(std::string(who) + ": shapes with vertices but no vertex array");
      return false;
    }
  }
  return true;
}

// (Re)build the device shape tables of `lib` from a shape / vertex table: both precisions, the kind bytes of k_classify
// and the set of buckets a pair of these kinds can fall into.
static bool upload_shapes(hfcl_lib* lib, const hfcl_shape* shapes, size_t n_shapes, const double* vertices, size_t n_vertices) {
  lib->n_shapes = n_shapes;
  lib->h_shapes.assign(shapes, shapes + n_shapes);
  lib->h_verts.assign(vertices, vertices + (vertices ? 3 * n_vertices : 0));
  lib->h_graphs.clear();  // shape ids / vertex ranges may have changed: adjacency is registered again by the caller
  lib->graph_dirty = true;
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
    lib->has_curved = present[K_ELLIPSOID] || present[K_CONE] || present[K_CYLINDER];
    lib->has_flats = present[K_PLANE] || present[K_HALFSPACE];
  }
  lib->h_kinds = kinds;
  std::vector<float> v32(3 * n_vertices + 3);
  for (size_t i = 0; i < 3 * n_vertices; ++i) v32[i] = float(vertices[i]);
  hipFree(lib->d_shapes64); hipFree(lib->d_shapes32); hipFree(lib->d_kinds); hipFree(lib->d_verts64); hipFree(lib->d_verts32);
  lib->d_shapes64 = nullptr; lib->d_shapes32 = nullptr; lib->d_kinds = nullptr; lib->d_verts64 = nullptr; lib->d_verts32 = nullptr;
  bool ok = true;
  ok = ok && hipMalloc(&lib->d_shapes64, n_shapes * sizeof(DShape<double>)) == hipSuccess;
  ok = ok && hipMalloc(&lib->d_shapes32, n_shapes * sizeof(DShape<float>)) == hipSuccess;
  ok = ok && hipMalloc(&lib->d_kinds, n_shapes) == hipSuccess;
  ok = ok && hipMalloc(&lib->d_verts64, (3 * n_vertices + 3) * sizeof(double)) == hipSuccess;
  ok = ok && hipMalloc(&lib->d_verts32, (3 * n_vertices + 3) * sizeof(float)) == hipSuccess;
  if (ok) {
    ok = ok && hipMemcpy(lib->d_shapes64, s64.data(), n_shapes * sizeof(DShape<double>), hipMemcpyHostToDevice) == hipSuccess;
    ok = ok && hipMemcpy(lib->d_shapes32, s32.data(), n_shapes * sizeof(DShape<float>), hipMemcpyHostToDevice) == hipSuccess;
    ok = ok && hipMemcpy(lib->d_kinds, kinds.data(), n_shapes, hipMemcpyHostToDevice) == hipSuccess;
    if (n_vertices) {
      ok = ok && hipMemcpy(lib->d_verts64, vertices, 3 * n_vertices * sizeof(double), hipMemcpyHostToDevice) == hipSuccess;
      ok = ok && hipMemcpy(lib->d_verts32, v32.data(), 3 * n_vertices * sizeof(float), hipMemcpyHostToDevice) == hipSuccess;
    }
  }
  return ok;
}


// ---------------------------------------------------------------------------------------
// Tuning options (include/hppfcl_amd.h: hfcl_lib_set_option).  One table: the keys, and what each sets.  Every option is a field of
// the library read when a batch is set up, so an option holds from the next call on; none changes a record (tests/ hold the forms
// against each other), they choose between forms of the same computation and size their budgets.
// ---------------------------------------------------------------------------------------
static const char* const* option_keys() {
  static const char* const keys[] = {
      "closed_staged", "split", "epa_cc_staged", "epa_records_aside", "epa_general_staged", "shape_finish_tiers", "shape_finish_aside",
      "epa_general_staged_min", "epa64_two_streams", "epa_cc_staged_min", "pipe_chunk", "bvh_filter", "bvh_shape_lane", "shape_coop",
      "bvh_cut_ticks", "shape_cut_ticks", "bvh_coop", "bvhd_budget", "bvhd_pool", "shape_dist_pool", "pool_rerun", "bvh_walk_early_coop",
      "bvh_walk_rounds", "bvh_walk_order", "mesh_beside", "mesh_prio", "shape_walk", "shape_walk_sort", "shape_walk_budget", "shape_walk_min", "gjk_beside_max", "epa_direct_max", "bvh_walk_k", "bvh_walk_budget", "shape_dist_leaf_min", "shape_dist_starve", "bvhd_leaf_min", "bvhd_starve",
      "bvhd_part_min", "shape_dist_budget", "bvh_budget0_coop", "shape_budget0", "shape_budget", "shape_leaf_cost", "shape_levels",
      "climb_min", "bvh_budget", "bvh_budget0", "bvh_levels", "cvx_w", "epa_resume_slots", "bvh_task_slots", "bvh_force_wide",
      "pipe_trace", nullptr};
  return keys;
}
// "4,16,16": up to `cap` comma-separated unsigned values into out[first...]; returns how many were read
static int parse_list(const char* v, uint32_t* out, int first, int cap, uint32_t lo, uint32_t hi) {
  int k = first;
  for (const char* p = v; *p && k < cap; ++k) {
    const long long x = atoll(p);
    out[k] = uint32_t(std::min<long long>(std::max<long long>(x, lo), hi));
    while (*p && *p != ',') ++p;
    if (*p == ',') ++p;
  }
  return k - first;
}
static int apply_option(hfcl_lib* lib, const std::string& key, const char* v) {
  const long long i = atoll(v);
  const bool on = i != 0;
  auto u32 = [&](long long lo) { return uint32_t(std::min<long long>(std::max(i, lo), 0xFFFFFFFFll)); };
  if (key == "closed_staged") lib->closed_staged = on;
  else if (key == "split") lib->split = i >= 2 ? 2 : (i == 1 ? 1 : 0);
  else if (key == "epa_cc_staged") lib->epa_cc_staged = on;
  else if (key == "epa_records_aside") lib->records_aside = on;
  else if (key == "epa_general_staged") {
    if (on && !HFCL_KEEP_AB_FORMS) return HFCL_ERR_INVALID_ARGUMENT;  // (not in the product build: hfcl_dev.hpp)
    lib->epa_general_staged = on;
  }
  else if (key == "shape_finish_tiers") lib->shape_finish_tiers = on;
  else if (key == "shape_finish_aside") lib->shape_finish_aside = on;
  else if (key == "epa_general_staged_min") lib->epa_general_staged_min = size_t(std::max(0ll, i));
  else if (key == "epa64_two_streams") lib->epa64_two_streams = on;
  else if (key == "epa_cc_staged_min") lib->epa_cc_staged_min = size_t(std::max(0ll, i));
  else if (key == "pipe_chunk") lib->pipe_chunk = strtoull(v, nullptr, 10);
  else if (key == "bvh_filter") {
    if (on && !HFCL_KEEP_AB_FORMS) return HFCL_ERR_INVALID_ARGUMENT;
    lib->bvh_filter = on;
  }
  else if (key == "bvh_shape_lane") lib->bvh_shape_lane = on;
  else if (key == "shape_coop") lib->shape_coop = on;
  else if (key == "bvh_cut_ticks") lib->bvh_cut_ticks = uint32_t(strtoul(v, nullptr, 10));
  else if (key == "shape_cut_ticks") lib->shape_cut_ticks = uint32_t(strtoul(v, nullptr, 10));
  else if (key == "bvh_coop") lib->bvh_coop = on;
  else if (key == "bvhd_budget") lib->bvhd_budget = u32(0);
  else if (key == "bvhd_pool") lib->bvhd_pool = u32(0);
  else if (key == "shape_dist_pool") lib->shape_dist_pool = u32(0);
  else if (key == "pool_rerun") lib->pool_rerun = u32(0);
  else if (key == "bvh_walk_early_coop") lib->walk_early_coop = on;
  else if (key == "bvh_walk_order") lib->walk_order = on;
  else if (key == "mesh_beside") lib->mesh_beside = u32(0);
  else if (key == "shape_walk") lib->shape_walk = on;
  else if (key == "mesh_prio") lib->mesh_prio = on;
  else if (key == "shape_walk_sort") lib->shape_walk_sort = on;
  else if (key == "shape_walk_budget") lib->shape_walk_budget = u32(0);
  else if (key == "shape_walk_min") lib->shape_walk_min = u32(0);
  else if (key == "gjk_beside_max") lib->gjk_beside_max = u32(0);
  else if (key == "epa_direct_max") lib->epa_direct_max = u32(0);
  else if (key == "bvh_walk_rounds") { lib->walk_rounds = uint32_t(std::min<long long>(std::max(0ll, i), WALK_ROUNDS)); lib->walk_auto = false; }
  else if (key == "bvh_walk_k") { parse_list(v, lib->walk_k, 0, WALK_ROUNDS, 1u, uint32_t(WALK_K)); lib->walk_auto = false; }  // "6,16": per round
  else if (key == "bvh_walk_budget") parse_list(v, lib->walk_budget, 1, WALK_ROUNDS, 0u, 0xFFFFFFFFu);  // rounds 1 ...: box tests (round 0 takes bvh_budget0_coop's)
  else if (key == "shape_dist_leaf_min") lib->shape_dist_leaf_min = u32(1);
  else if (key == "shape_dist_starve") lib->shape_dist_starve = u32(1);  // (>= 1: a window of triangles alone must always run)
  else if (key == "bvhd_leaf_min") lib->bvhd_pool_leaf_min = u32(1);
  else if (key == "bvhd_starve") lib->bvhd_pool_starve = u32(1);
  else if (key == "bvhd_part_min") lib->bvhd_pool_part_min = u32(0);
  else if (key == "shape_dist_budget") lib->shape_dist_budget = u32(0);
  else if (key == "bvh_budget0_coop") lib->bvh_budget0_coop = u32(1);
  else if (key == "shape_budget0") lib->shape_budget0 = lib->shape_budget0_coop = uint32_t(i);
  else if (key == "shape_budget") lib->shape_budget = uint32_t(i);
  else if (key == "shape_leaf_cost") lib->shape_leaf_cost = u32(1);
  else if (key == "shape_levels") lib->shape_levels = uint32_t(std::min<long long>(std::max(1ll, i), BVH_MAX_LEVELS));
  else if (key == "climb_min") lib->climb_min = u32(0);
  else if (key == "bvh_budget") { lib->bvh_budget = lib->bvh_budget0 = u32(0); lib->bvh_auto = false; }
  else if (key == "bvh_budget0") { lib->bvh_budget0 = u32(0); lib->bvh_auto = false; }
  else if (key == "bvh_levels") { lib->bvh_levels = uint32_t(std::min<long long>(BVH_MAX_LEVELS, std::max(1ll, i))); lib->bvh_auto = false; }
  else if (key == "cvx_w") {
    if (i == 0 || i == 2 || i == 4 || i == 8 || i == 16 || i == 32 || i == 64) lib->cvx_w = int(i);
    else return HFCL_ERR_INVALID_ARGUMENT;
  }
  else if (key == "epa_resume_slots") lib->epa_resume_slots = size_t(std::max(0ll, i));
  else if (key == "bvh_task_slots") lib->bvh_task_slots = size_t(std::max(0ll, i));
  else if (key == "bvh_force_wide") lib->bvh_force_wide = on;
  else if (key == "pipe_trace") lib->pipe_trace = on;
  else return HFCL_ERR_INVALID_ARGUMENT;
  return HFCL_OK;
}

hfcl_lib* hfcl_lib_create(const hfcl_shape* shapes, size_t n_shapes, const double* vertices, size_t n_vertices,
                          int device) {
  if (ensure_device(device) != HFCL_OK) return nullptr;
  if (!validate_shapes("hfcl_lib_create", shapes, n_shapes, vertices, n_vertices)) return nullptr;
  hfcl_lib* lib = new hfcl_lib();
  lib->device = device;
  bool ok = upload_shapes(lib, shapes, n_shapes, vertices, n_vertices);
  ok = ok && hipMalloc(&lib->d_counts, N_COUNTERS * sizeof(uint32_t)) == hipSuccess;
  ok = ok && hipHostMalloc((void**)&lib->h_counts, N_COUNTERS * sizeof(uint32_t), hipHostMallocDefault) == hipSuccess;
  if (ok) memset(lib->h_counts, 0, N_COUNTERS * sizeof(uint32_t));
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
  // Tuning options: hfcl_lib_set_option is the interface; the environment (HFCL_<KEY>) is read ONCE, here, as a fallback for
  // processes that cannot call it (A/B runs of an unchanged binary).
  for (const char* const* k = option_keys(); *k; ++k) {
    std::string name = "HFCL_";
    for (const char* c = *k; *c; ++c) name.push_back(char(toupper(static_cast<unsigned char>(*c))));
    if (const char* v = getenv(name.c_str())) apply_option(lib, *k, v);
  }
  return lib;
}

void hfcl_lib_destroy(hfcl_lib* lib) {
  if (!lib) return;
  hipSetDevice(lib->device);
  hipDeviceSynchronize();
  if (lib->helper) hfcl_lib_destroy(lib->helper);
  if (!lib->is_helper) free_retired_graphs(lib);
  if (lib->upload_stream) hipStreamDestroy(lib->upload_stream);
  if (lib->side) hipStreamDestroy(lib->side);
  if (lib->ev_fork) hipEventDestroy(lib->ev_fork);
  if (lib->ev_join) hipEventDestroy(lib->ev_join);
  if (!lib->is_helper) {
    hipFree(lib->d_shapes64);
    hipFree(lib->d_shapes32);
    hipFree(lib->d_kinds);
    hipFree(lib->d_verts64);
    hipFree(lib->d_verts32);
    hipFree(lib->d_graph_base);
    hipFree(lib->d_graph_off);
    hipFree(lib->d_graph_ent32);
    hipFree(lib->d_graph_ent64);
  }
  hipFree(lib->d_counts);
  if (lib->h_counts) hipHostFree(lib->h_counts);
  hipFree(lib->d_lists);
  hipFree(lib->d_epa_queue);
  hipFree(lib->d_epa_queue2);
  hipFree(lib->d_epa_ready);
  hipFree(lib->d_epa_ready_g);
  hipFree(lib->d_epa_cc_over);
  if (lib->mesh_st) hipStreamDestroy(lib->mesh_st);
  if (lib->mesh_st2) hipStreamDestroy(lib->mesh_st2);
  if (lib->mesh_aux) hipStreamDestroy(lib->mesh_aux);
  for (hipEvent_t e : {lib->ev_mesh_fork, lib->ev_mesh_join, lib->ev_mesh_fork2, lib->ev_mesh_join2})
    if (e) hipEventDestroy(e);
  hipFree(lib->d_bvh2_tasks); hipFree(lib->d_bvh2_sums); hipFree(lib->d_bvh2_susp); hipFree(lib->d_bvh2_ctr);
  for (int k = 0; k < 3; ++k) {
    if (lib->gjk_st[k]) hipStreamDestroy(lib->gjk_st[k]);
    if (lib->gjk_join[k]) hipEventDestroy(lib->gjk_join[k]);
  }
  if (lib->gjk_fork) hipEventDestroy(lib->gjk_fork);
  for (int k = 0; k < WALK_ROUNDS - 1; ++k) {
    if (lib->walk_st[k]) hipStreamDestroy(lib->walk_st[k]);
    if (lib->walk_fork[k]) hipEventDestroy(lib->walk_fork[k]);
    if (lib->walk_join[k]) hipEventDestroy(lib->walk_join[k]);
  }
  if (lib->aux) hipStreamDestroy(lib->aux);
  if (lib->ev_aux0) hipEventDestroy(lib->ev_aux0);
  if (lib->ev_aux1) hipEventDestroy(lib->ev_aux1);
  if (lib->ev_aux2) hipEventDestroy(lib->ev_aux2);
  if (lib->ev_aux3) hipEventDestroy(lib->ev_aux3);
  hipFree(lib->d_epa_resume);
  hipFree(lib->d_epa_v0);
  hipFree(lib->d_shape_defer);
  hipFree(lib->d_shape_oq);
  hipFree(lib->d_dist_susp);
  hipFree(lib->d_shape_dist_susp);
  if (lib->h_pack) hipHostFree(lib->h_pack);
  if (lib->h_pack_counts) hipHostFree(lib->h_pack_counts);
  hipFree(lib->d_pack);
  for (auto& sg : lib->stage) {
    hipFree(sg.d_s1);
    hipFree(sg.d_s2);
    hipFree(sg.d_tf1);
    hipFree(sg.d_tf2);
    hipFree(sg.d_qt1);
    hipFree(sg.d_qt2);
    hipFree(sg.d_out);
    hipFree(sg.d_gin);
    hipFree(sg.d_gout);
    if (sg.ev_in) hipEventDestroy(sg.ev_in);
    if (sg.ev_done) hipEventDestroy(sg.ev_done);
    if (sg.ev_in2) hipEventDestroy(sg.ev_in2);
    if (sg.h_counts) hipHostFree(sg.h_counts);
    if (sg.h_counts2) hipHostFree(sg.h_counts2);
  }
  if (lib->s_h2d) hipStreamDestroy(lib->s_h2d);
  if (lib->s_h2d2) hipStreamDestroy(lib->s_h2d2);
  if (lib->s_cmp) hipStreamDestroy(lib->s_cmp);
  if (lib->s_d2h) hipStreamDestroy(lib->s_d2h);
  hipFree(lib->d_nodes64);
  hipFree(lib->d_nodes32);
  hipFree(lib->d_fnodes);
  hipFree(lib->d_rss64);
  hipFree(lib->d_rss32);
  hipFree(lib->d_dnodes64);
  hipFree(lib->d_dnodes32);
  hipFree(lib->d_bverts64);
  hipFree(lib->d_bverts32);
  hipFree(lib->d_btris);
  hipFree(lib->d_meshes);
  hipFree(lib->d_contacts);
  hipFree(lib->d_contacts_count);
  hipFree(lib->d_bvh_tasks);
  hipFree(lib->d_bvh_cut_words);
  hipFree(lib->d_bvh_cut_vals);
  hipFree(lib->d_bvh_slab);
  hipFree(lib->d_bvh_sums);
  hipFree(lib->d_bvh_susp);
  hipFree(lib->d_bvh_ctr);
  hipFree(lib->d_walk_recs); hipFree(lib->d_walk_items); hipFree(lib->d_walk_res); hipFree(lib->d_walk_lists); hipFree(lib->d_walk_ctr); hipFree(lib->d_walk_order);
  hipFree(lib->d_swalk_recs); hipFree(lib->d_swalk_items); hipFree(lib->d_swalk_res); hipFree(lib->d_swalk_lists); hipFree(lib->d_swalk_ctr); hipFree(lib->d_swalk_perm);
  for (auto& t : lib->timers) {
    hipEventDestroy(t.e0);
    hipEventDestroy(t.e1);
  }
  delete lib;
}
// Replace the shape table of a library in place: registered BVH models, workspaces, staging buffers and streams stay
// (a caller that keeps adding geometries -- the hpp::fcl shim -- does not pay a full rebuild with every new shape).
int hfcl_lib_set_shapes(hfcl_lib* lib, const hfcl_shape* shapes, size_t n_shapes, const double* vertices, size_t n_vertices) {
  if (!lib) {
    set_error("null library");
    return HFCL_ERR_INVALID_ARGUMENT;
  }
  if (!validate_shapes("hfcl_lib_set_shapes", shapes, n_shapes, vertices, n_vertices)) return HFCL_ERR_INVALID_ARGUMENT;
  HIP_TRY(hipSetDevice(lib->device));
  HIP_TRY(hipDeviceSynchronize());  // nothing of this library may still be reading the old tables
  free_retired_graphs(lib);
  if (!upload_shapes(lib, shapes, n_shapes, vertices, n_vertices)) {
    set_error("hfcl_lib_set_shapes: HIP allocation/copy failed");
    return HFCL_ERR_HIP;
  }
  if (lib->helper) share_tables(lib->helper, lib);  // the second half of split batches reads the same tables
  return HFCL_OK;
}

// Vertex adjacency of a convex shape (ConvexBase::neighbors, shape/geometric_shapes.h; Convex<PolygonT>::fillNeighbors,
// shape/details/convex.hxx:231-280): offsets[num_points + 1] into neighbors[], vertex indices relative to the shape's first
// vertex.  Hulls of at least HFCL_CLIMB_MIN vertices that have one answer support queries by hill-climbing
// (getShapeSupportLog) instead of scanning all vertices; without one (or below the threshold) nothing changes.
int hfcl_lib_set_convex_neighbors(hfcl_lib* lib, uint32_t shape_id, const uint32_t* offsets, const uint32_t* neighbors) {
  if (!lib || !offsets || !neighbors) {
    set_error("hfcl_lib_set_convex_neighbors: null argument");
    return HFCL_ERR_INVALID_ARGUMENT;
  }
  if (shape_id >= lib->n_shapes || lib->h_shapes[shape_id].type != HFCL_GEOM_CONVEX) {
    set_error("hfcl_lib_set_convex_neighbors: not a convex shape of this library");
    return HFCL_ERR_INVALID_ARGUMENT;
  }
  const uint32_t n = lib->h_shapes[shape_id].num_points;
  if (offsets[0] != 0) {
    set_error("hfcl_lib_set_convex_neighbors: offsets[0] must be 0");
    return HFCL_ERR_INVALID_ARGUMENT;
  }
  for (uint32_t i = 0; i < n; ++i)
    if (offsets[i + 1] < offsets[i]) {
      set_error("hfcl_lib_set_convex_neighbors: offsets must not decrease");
      return HFCL_ERR_INVALID_ARGUMENT;
    }
  if (offsets[n] == 0) {
    set_error("hfcl_lib_set_convex_neighbors: empty adjacency");
    return HFCL_ERR_INVALID_ARGUMENT;
  }
  for (uint32_t k = 0; k < offsets[n]; ++k)
    if (neighbors[k] >= n) {
      set_error("hfcl_lib_set_convex_neighbors: neighbour index outside the shape's vertices");
      return HFCL_ERR_INVALID_ARGUMENT;
    }
  hfcl_lib::HostGraph& g = lib->h_graphs[shape_id];
  g.off.assign(offsets, offsets + n + 1);
  g.ids.assign(neighbors, neighbors + offsets[n]);
  lib->graph_dirty = true;
  return HFCL_OK;
}
size_t hfcl_lib_num_shapes(const hfcl_lib* lib) { return lib ? lib->n_shapes : 0; }
int hfcl_lib_device(const hfcl_lib* lib) { return lib ? lib->device : -1; }
uint32_t hfcl_lib_climb_min(const hfcl_lib* lib) { return lib ? lib->climb_min : 0u; }

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
  if (n_nodes > 0x7FFFFFF0u || n_vertices > 0xFFFFFFF0u) {
    set_error("hfcl_lib_add_bvh: model too large (node / vertex ids are 32-bit)");
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
  {  // depth of the tree (the traversal stacks are sized from it) -- iterative: degenerate models are as deep as they are big
    std::vector<std::pair<uint32_t, uint32_t>> todo;
    todo.emplace_back(0u, 1u);
    uint32_t depth = 1;
    while (!todo.empty()) {
      const auto [i, d] = todo.back();
      todo.pop_back();
      depth = std::max(depth, d);
      const int fc = nodes[i].first_child;
      if (fc > 0) {
        todo.emplace_back(uint32_t(fc), d + 1);
        todo.emplace_back(uint32_t(fc) + 1u, d + 1);
      }
    }
    lib->bvh_max_depth = std::max(lib->bvh_max_depth, depth);
    lib->bvh_max_nodes = std::max<size_t>(lib->bvh_max_nodes, n_nodes);
  }
  lib->bvh_dirty = true;
  return int(lib->h_meshes.size() - 1);
}

}  // extern "C"

// Device workspace of a batch of n pairs.  The bucket lists (4 B per pair and bucket the library's shape kinds can reach)
// are always needed; the EPA queues (two seeds of ~230 B per pair) and the hand-over area (one slot of ~4 KB per 8 pairs)
// only when the batch has a GJK bucket and asks for penetration data -- a closed-form or mesh-only library never pays
// for them.  ~0.05 KB per pair without EPA, ~1 KB with (it was 1.8 KB for every library).
static int ensure_workspace(hfcl_lib* lib, size_t n, bool need_epa) {
  if (n > lib->ws_capacity) {
    hipFree(lib->d_lists);
    lib->d_lists = nullptr;
    lib->ws_capacity = 0;
    const size_t cap = n + n / 8 + 1024;
    HIP_TRY(hipMalloc(&lib->d_lists, size_t(B_COUNT) * cap * sizeof(uint32_t)));
    lib->ws_capacity = cap;
  }
  if (need_epa && lib->ws_capacity > lib->epa_capacity) {
    hipFree(lib->d_epa_queue);
    hipFree(lib->d_epa_queue2);
    hipFree(lib->d_epa_resume);
    hipFree(lib->d_epa_cc_over);
    lib->d_epa_queue = nullptr;
    lib->d_epa_queue2 = nullptr;
    lib->d_epa_resume = nullptr;
    lib->d_epa_cc_over = nullptr;
    lib->resume_cap = 0;
    lib->epa_capacity = 0;
    const size_t cap = lib->ws_capacity;
    HIP_TRY(hipMalloc(&lib->d_epa_queue, cap * sizeof(EpaItem<double>)));
    HIP_TRY(hipMalloc(&lib->d_epa_queue2, cap * sizeof(EpaItem<double>)));
    // saved polytopes for the tier hand-over: room for an eighth of the batch (cfg5: 4 % of the pairs outgrow the fast
    // tier; beyond the area the full tier simply redoes the pair from its seed)
    size_t rcap = std::min(cap, std::max<size_t>(65536, cap / 8));
    if (lib->epa_resume_slots) rcap = std::max<size_t>(1, std::min<size_t>(cap, lib->epa_resume_slots));  // test knob (option epa_resume_slots)
    HIP_TRY(hipMalloc(&lib->d_epa_resume, rcap * std::max(epa_resume_stride<double>, epa_resume_stride<float>)));
    HIP_TRY(hipMalloc((void**)&lib->d_epa_cc_over, rcap * sizeof(uint32_t)));
    lib->resume_cap = rcap;
    lib->epa_capacity = cap;
  }
  return HFCL_OK;
}

// Tables of a split traversal (mesh x mesh, mesh x solid) for a batch of n queries: room for 16 tasks per query (a long query suspends
// with a stack of ~20 entries, one query in five is long; mesh x solid walks are cut finer) -- ~2.4 KB of device memory per query in fp64.
static int ensure_bvh_split(hfcl_lib* lib, size_t n) {
  if (n <= lib->bvh_split_n) return HFCL_OK;
  hipFree(lib->d_bvh_tasks); hipFree(lib->d_bvh_sums); hipFree(lib->d_bvh_susp); hipFree(lib->d_bvh_cut_words); hipFree(lib->d_bvh_cut_vals);
  lib->d_bvh_tasks = nullptr; lib->d_bvh_sums = nullptr; lib->d_bvh_susp = nullptr; lib->d_bvh_cut_words = nullptr; lib->d_bvh_cut_vals = nullptr;
  lib->bvh_split_n = 0;
  size_t per_query = 16;
  if (lib->bvh_task_slots) per_query = std::max<size_t>(1, lib->bvh_task_slots);  // test / tuning knob (option bvh_task_slots)
  const size_t nq = n + n / 8 + 1024, cap = per_query * nq + 65536;
  HIP_TRY(hipMalloc(&lib->d_bvh_tasks, cap * sizeof(BvhTask)));
  HIP_TRY(hipMalloc(&lib->d_bvh_sums, (nq + cap) * sizeof(BvhSum<double>)));
  HIP_TRY(hipMalloc(&lib->d_bvh_cut_words, cap * sizeof(uint32_t)));
  HIP_TRY(hipMalloc(&lib->d_bvh_cut_vals, cap * sizeof(double)));
  HIP_TRY(hipMalloc(&lib->d_bvh_susp, nq * sizeof(uint32_t)));
  if (!lib->d_bvh_ctr) HIP_TRY(hipMalloc(&lib->d_bvh_ctr, BVH_CTR_WORDS * sizeof(uint32_t)));
  lib->bvh_split_n = nq;
  lib->bvh_split_cap = cap;
  return HFCL_OK;
}

static int ensure_walk(hfcl_lib* lib, size_t n) {
  if (n <= lib->walk_n) return HFCL_OK;
  hipFree(lib->d_walk_recs); hipFree(lib->d_walk_items); hipFree(lib->d_walk_res); hipFree(lib->d_walk_lists); hipFree(lib->d_walk_order);
  lib->d_walk_order = nullptr;
  lib->d_walk_recs = nullptr; lib->d_walk_items = nullptr; lib->d_walk_res = nullptr; lib->d_walk_lists = nullptr;
  lib->walk_n = 0;
  const size_t nq = n + n / 8 + 1024;
  HIP_TRY(hipMalloc(&lib->d_walk_recs, nq * sizeof(WalkRec<double>)));
  HIP_TRY(hipMalloc(&lib->d_walk_items, nq * WALK_K * sizeof(uint32_t)));
  HIP_TRY(hipMalloc(&lib->d_walk_res, nq * WALK_K * 10 * sizeof(double)));  // TriLeafOut<double>: distance, p1, p2, n
  HIP_TRY(hipMalloc(&lib->d_walk_lists, 2 * nq * sizeof(uint32_t)));
  HIP_TRY(hipMalloc(&lib->d_walk_order, nq * sizeof(uint32_t)));
  if (!lib->d_walk_ctr) HIP_TRY(hipMalloc(&lib->d_walk_ctr, 8 * WALK_ROUNDS * sizeof(uint32_t)));
  lib->walk_n = nq;
  return HFCL_OK;
}

static int ensure_swalk(hfcl_lib* lib, size_t n) {
  if (n <= lib->swalk_n) return HFCL_OK;
  hipFree(lib->d_swalk_recs); hipFree(lib->d_swalk_items); hipFree(lib->d_swalk_res); hipFree(lib->d_swalk_lists); hipFree(lib->d_swalk_perm);
  lib->d_swalk_recs = nullptr; lib->d_swalk_items = nullptr; lib->d_swalk_res = nullptr; lib->d_swalk_lists = nullptr; lib->d_swalk_perm = nullptr;
  lib->swalk_n = 0;
  const size_t nq = n + n / 8 + 1024;
  HIP_TRY(hipMalloc(&lib->d_swalk_recs, nq * sizeof(WalkRec<double>)));
  HIP_TRY(hipMalloc(&lib->d_swalk_items, nq * WALK_K * sizeof(uint32_t)));
  HIP_TRY(hipMalloc(&lib->d_swalk_res, nq * WALK_K * 10 * sizeof(double)));  // TriLeafOut<double>: distance, p1, p2, n
  HIP_TRY(hipMalloc(&lib->d_swalk_lists, 3 * nq * sizeof(uint32_t)));
  HIP_TRY(hipMalloc(&lib->d_swalk_perm, (nq * WALK_K + nq) * sizeof(uint32_t)));
  if (!lib->d_swalk_ctr) HIP_TRY(hipMalloc(&lib->d_swalk_ctr, (8 * WALK_ROUNDS + 64) * sizeof(uint32_t)));
  lib->swalk_n = nq;
  return HFCL_OK;
}

// How the mesh x mesh traversals of this library keep their stacks.  A stack never holds more than depth1 + depth2 + 2
// entries (every step pops one entry and pushes at most two, one level deeper in one of the trees).
static int make_bvh_spill(hfcl_lib* lib, BvhSpill& sp, bool distance) {
  memset(&sp, 0, sizeof(sp));
  const size_t need = 2 * size_t(lib->bvh_max_depth) + 4;
  // collide(): a full LDS stack first suspends into tasks (HFCL_BVH_LEVELS levels of BVH_STACK entries); distance() has
  // no task form: anything deeper than its LDS stack takes the wide form with slabs
  const size_t narrow_holds = distance ? size_t(BVHD_STACK) : size_t(std::min(BVH_STACK, BVH_STACK_FILT)) * HFCL_BVH_LEVELS;
  sp.wide = (lib->bvh_max_nodes > 65535 || need > narrow_holds) ? 1u : 0u;
  if (lib->bvh_force_wide) sp.wide = 1u;  // test knob (option bvh_force_wide): the wide form (and its slabs) on small models
  if (!sp.wide || need <= size_t(std::min(BVH_STACK, BVH_STACK_FILT)) / 2) return HFCL_OK;  // the LDS stack of the wide form suffices
  const size_t cap = ((need + 63) / 64) * 64;                     // entries per lane
  const size_t per_block = size_t(BVH_BLOCK) * cap * 2 * sizeof(uint64_t);  // (entry, bound) records: k_bvh_distance
  size_t blocks = std::min<size_t>(size_t(lib->n_cus) * 16, std::max<size_t>(1, (size_t(2) << 30) / per_block));
  const size_t bytes = blocks * per_block;
  if (bytes > lib->bvh_slab_bytes) {
    hipFree(lib->d_bvh_slab);
    lib->d_bvh_slab = nullptr;
    lib->bvh_slab_bytes = 0;
    HIP_TRY(hipMalloc(&lib->d_bvh_slab, bytes));
    lib->bvh_slab_bytes = bytes;
  }
  sp.slab = lib->d_bvh_slab;
  sp.cap = uint32_t(cap);
  sp.max_blocks = uint32_t(blocks);
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

// Device image of the registered vertex adjacencies: per shape [14 warm-start vertices][num_points + 1 offsets] in
// d_graph_off, the neighbours with their coordinates inline in d_graph_ent32/64 (one fetch per hop instead of two).
// The warm-start vertices are the support vertices along the 14 directions of ConvexBase::buildSupportWarmStart
// (src/shape/geometric_shapes.cpp: the six axis directions and the eight cube diagonals).
static void free_retired_graphs(hfcl_lib* lib) {  // (the caller has waited for the device)
  for (void* p : lib->graph_retired) hipFree(p);
  lib->graph_retired.clear();
}
// No device-wide wait (round 4): the previous image is retired, not freed, the new one is copied on a non-blocking stream
// of its own and the host waits for that stream only -- hipFree and the null-stream hipMemcpy the first version used both
// stall every stream of the device, inside an entry point documented as asynchronous.
static int upload_graph(hfcl_lib* lib) {
  if (!lib->graph_dirty) return HFCL_OK;
  HIP_TRY(hipSetDevice(lib->device));
  // (an application that re-registers neighbours between batches must not grow device memory without bound: after four retired
  // images the device is waited for once and they are freed)
  if (lib->graph_retired.size() >= 16) {
    HIP_TRY(hipDeviceSynchronize());
    free_retired_graphs(lib);
  }
  for (void* p : {(void*)lib->d_graph_base, (void*)lib->d_graph_off, (void*)lib->d_graph_ent32, (void*)lib->d_graph_ent64})
    if (p) lib->graph_retired.push_back(p);
  lib->d_graph_base = nullptr; lib->d_graph_off = nullptr; lib->d_graph_ent32 = nullptr; lib->d_graph_ent64 = nullptr;
  lib->graph_dirty = false;
  if (lib->h_graphs.empty()) {
    if (lib->helper) share_tables(lib->helper, lib);
    return HFCL_OK;
  }
  std::vector<uint32_t> base(lib->n_shapes, HFCL_NO_GRAPH), off;
  std::vector<NbrEntry<float>> e32;
  std::vector<NbrEntry<double>> e64;
  static const double dirs[HFCL_WARM_STARTS][3] = {{1, 0, 0}, {-1, 0, 0}, {0, 1, 0}, {0, -1, 0}, {0, 0, 1}, {0, 0, -1},
                                                   {1, 1, 1}, {-1, 1, 1}, {1, -1, 1}, {-1, -1, 1}, {1, 1, -1}, {-1, 1, -1}, {1, -1, -1}, {-1, -1, -1}};
  for (const auto& kv : lib->h_graphs) {
    const hfcl_shape& sh = lib->h_shapes[kv.first];
    const hfcl_lib::HostGraph& g = kv.second;
    const double* v = lib->h_verts.data() + 3 * size_t(sh.vertex_offset);
    for (int k = 0; k < HFCL_WARM_STARTS; ++k) {
      uint32_t bi = 0;
      double best = -1.7976931348623157e+308;
      for (uint32_t j = 0; j < sh.num_points; ++j) {
        if (g.off[j + 1] == g.off[j]) continue;  // not a hull vertex: no way on from there
        const double d = v[3 * size_t(j)] * dirs[k][0] + v[3 * size_t(j) + 1] * dirs[k][1] + v[3 * size_t(j) + 2] * dirs[k][2];
        if (d > best) {
          best = d;
          bi = j;
        }
      }
      off.push_back(bi);
    }
    base[kv.first] = uint32_t(off.size());
    const size_t e0 = e64.size();
    if (e0 + g.ids.size() > 0xFFFFFFF0u) {
      set_error("hfcl_lib_set_convex_neighbors: more than 2^32 adjacency entries in one library");
      return HFCL_ERR_INVALID_ARGUMENT;
    }
    for (uint32_t o : g.off) off.push_back(uint32_t(e0) + o);
    for (uint32_t id : g.ids) {
      const double* p = v + 3 * size_t(id);
      NbrEntry<double> a;
      a.x = p[0]; a.y = p[1]; a.z = p[2]; a.id = id; a.pad_ = 0;
      NbrEntry<float> b;
      b.x = float(p[0]); b.y = float(p[1]); b.z = float(p[2]); b.id = id;
      e64.push_back(a);
      e32.push_back(b);
    }
  }
  bool ok = hipMalloc(&lib->d_graph_base, base.size() * sizeof(uint32_t)) == hipSuccess;
  ok = ok && hipMalloc(&lib->d_graph_off, off.size() * sizeof(uint32_t)) == hipSuccess;
  ok = ok && hipMalloc(&lib->d_graph_ent32, (e32.size() + 1) * sizeof(NbrEntry<float>)) == hipSuccess;
  ok = ok && hipMalloc(&lib->d_graph_ent64, (e64.size() + 1) * sizeof(NbrEntry<double>)) == hipSuccess;
  if (!lib->upload_stream) ok = ok && hipStreamCreateWithFlags(&lib->upload_stream, hipStreamNonBlocking) == hipSuccess;
  hipStream_t us = lib->upload_stream;
  ok = ok && hipMemcpyAsync(lib->d_graph_base, base.data(), base.size() * sizeof(uint32_t), hipMemcpyHostToDevice, us) == hipSuccess;
  ok = ok && hipMemcpyAsync(lib->d_graph_off, off.data(), off.size() * sizeof(uint32_t), hipMemcpyHostToDevice, us) == hipSuccess;
  ok = ok && (e32.empty() || hipMemcpyAsync(lib->d_graph_ent32, e32.data(), e32.size() * sizeof(NbrEntry<float>), hipMemcpyHostToDevice, us) == hipSuccess);
  ok = ok && (e64.empty() || hipMemcpyAsync(lib->d_graph_ent64, e64.data(), e64.size() * sizeof(NbrEntry<double>), hipMemcpyHostToDevice, us) == hipSuccess);
  ok = ok && hipStreamSynchronize(us) == hipSuccess;  // the host vectors go out of scope; kernels launched after this see the image
  if (!ok) {  // nothing half-built stays behind: the next batch retries the upload from the host copies
    hipFree(lib->d_graph_base); hipFree(lib->d_graph_off); hipFree(lib->d_graph_ent32); hipFree(lib->d_graph_ent64);
    lib->d_graph_base = nullptr; lib->d_graph_off = nullptr; lib->d_graph_ent32 = nullptr; lib->d_graph_ent64 = nullptr;
    lib->graph_dirty = true;
    if (lib->helper) share_tables(lib->helper, lib);
    set_error("vertex adjacency: HIP allocation/copy failed");
    return HFCL_ERR_HIP;
  }
  if (lib->helper) share_tables(lib->helper, lib);
  return HFCL_OK;
}

static int upload_bvh(hfcl_lib* lib) {
  if (!lib->bvh_dirty) return HFCL_OK;
  hipFree(lib->d_nodes64); hipFree(lib->d_nodes32); hipFree(lib->d_bverts64); hipFree(lib->d_bverts32);
  hipFree(lib->d_btris); hipFree(lib->d_meshes); hipFree(lib->d_rss64); hipFree(lib->d_rss32); hipFree(lib->d_fnodes);
  hipFree(lib->d_dnodes64); hipFree(lib->d_dnodes32);
  lib->d_dnodes64 = nullptr; lib->d_dnodes32 = nullptr;
  lib->d_rss64 = nullptr; lib->d_rss32 = nullptr; lib->d_fnodes = nullptr;
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
  {  // filter records: sizes compared through their rank among all nodes of the library (exact, hfcl_bvh.hpp)
    std::vector<uint32_t> rank(nn);
    obbf_size_ranks(lib->h_bvh_nodes.data(), nn, rank.data());
    std::vector<DNodeF> fn(nn);
    for (size_t i = 0; i < nn; ++i) fn[i] = pack_fnode(lib->h_bvh_nodes[i], rank[i]);
    HIP_TRY(hipMalloc(&lib->d_fnodes, nn * sizeof(DNodeF)));
    HIP_TRY(hipMemcpy(lib->d_fnodes, fn.data(), nn * sizeof(DNodeF), hipMemcpyHostToDevice));
    // the distance() walk's records: axes + RSS + child link + the same ranks
    std::vector<DNodeD<double>> d64(nn);
    std::vector<DNodeD<float>> d32(nn);
    for (size_t i = 0; i < nn; ++i) {
      d64[i].axes = n64[i].axes; d64[i].Tr = r64[i].Tr; d64[i].l0 = r64[i].l0; d64[i].l1 = r64[i].l1; d64[i].r = r64[i].r;
      d64[i].first_child = n64[i].first_child; d64[i].rank = rank[i];
      d32[i].axes = n32[i].axes; d32[i].Tr = r32[i].Tr; d32[i].l0 = r32[i].l0; d32[i].l1 = r32[i].l1; d32[i].r = r32[i].r;
      d32[i].first_child = n32[i].first_child; d32[i].rank = rank[i];
    }
    HIP_TRY(hipMalloc(&lib->d_dnodes64, nn * sizeof(DNodeD<double>)));
    HIP_TRY(hipMalloc(&lib->d_dnodes32, nn * sizeof(DNodeD<float>)));
    HIP_TRY(hipMemcpy(lib->d_dnodes64, d64.data(), nn * sizeof(DNodeD<double>), hipMemcpyHostToDevice));
    HIP_TRY(hipMemcpy(lib->d_dnodes32, d32.data(), nn * sizeof(DNodeD<float>), hipMemcpyHostToDevice));
  }
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
// the helper stream of a library (tail kernels beside the main ones) and its fork / join events, made on first use
static int ensure_aux(hfcl_lib* lib) {
  if (lib->aux) return HFCL_OK;
  // created into locals and committed together: a failure half way leaves the library without a helper stream, not with null events
  hipStream_t s = nullptr;
  hipEvent_t ev[4] = {nullptr, nullptr, nullptr, nullptr};
  hipError_t e = hipStreamCreateWithFlags(&s, hipStreamNonBlocking);
  for (int k = 0; k < 4 && e == hipSuccess; ++k) e = hipEventCreateWithFlags(&ev[k], hipEventDisableTiming);
  if (e != hipSuccess) {
    for (int k = 0; k < 4; ++k)
      if (ev[k]) hipEventDestroy(ev[k]);
    if (s) hipStreamDestroy(s);
    HIP_TRY(e);
  }
  lib->ev_aux0 = ev[0];
  lib->ev_aux1 = ev[1];
  lib->ev_aux2 = ev[2];
  lib->ev_aux3 = ev[3];
  lib->aux = s;
  return HFCL_OK;
}
static int ensure_mesh_stream(hfcl_lib* lib) {
  if (lib->mesh_st) return HFCL_OK;  // (committed last)
  hipStream_t st[3] = {nullptr, nullptr, nullptr};
  hipEvent_t ev[4] = {nullptr, nullptr, nullptr, nullptr};
  hipError_t e = hipSuccess;
  // option mesh_prio: [0] the mesh x solid walks and [2] their helper at the device's highest priority -- hardware queues of their own (the runtime
  // maps the streams of one priority onto four queues; two chains that share one run one after the other): cfgmix 2.90 -> 2.72 ms, but a process
  // that has created them runs cfg4s's in-line batches 1 ms slower (3.7 against 2.65 ms; profiles/r06_g).  Off.
  int prio_lo = 0, prio_hi = 0;
  if (lib->mesh_prio) (void)hipDeviceGetStreamPriorityRange(&prio_lo, &prio_hi);
  for (int k = 0; k < 3 && e == hipSuccess; ++k) e = hipStreamCreateWithPriority(&st[k], hipStreamNonBlocking, k == 1 ? 0 : prio_hi);
  for (int k = 0; k < 4 && e == hipSuccess; ++k) e = hipEventCreateWithFlags(&ev[k], hipEventDisableTiming);
  if (e != hipSuccess) {
    for (hipEvent_t x : ev)
      if (x) hipEventDestroy(x);
    for (hipStream_t x : st)
      if (x) hipStreamDestroy(x);
    HIP_TRY(e);
  }
  lib->ev_mesh_fork = ev[0];
  lib->ev_mesh_join = ev[1];
  lib->ev_mesh_fork2 = ev[2];
  lib->ev_mesh_join2 = ev[3];
  lib->mesh_st2 = st[1];
  lib->mesh_aux = st[2];
  lib->mesh_st = st[0];
  return HFCL_OK;
}
// the second set of split-traversal tables (what k_bvh_walk / k_bvh_resolve / k_bvh_coop use of them: tasks, summaries, suspended list,
// counters): the mesh x mesh walks of a batch that also holds mesh x solid pairs run beside those on these
static int ensure_bvh_split2(hfcl_lib* lib, size_t n) {
  if (n <= lib->bvh2_split_n) return HFCL_OK;
  hipFree(lib->d_bvh2_tasks); hipFree(lib->d_bvh2_sums); hipFree(lib->d_bvh2_susp);
  lib->d_bvh2_tasks = nullptr; lib->d_bvh2_sums = nullptr; lib->d_bvh2_susp = nullptr;
  lib->bvh2_split_n = 0;
  const size_t nq = n + n / 8 + 1024, cap = 16 * nq + 65536;
  HIP_TRY(hipMalloc(&lib->d_bvh2_tasks, cap * sizeof(BvhTask)));
  HIP_TRY(hipMalloc(&lib->d_bvh2_sums, (nq + cap) * sizeof(BvhSum<double>)));
  HIP_TRY(hipMalloc(&lib->d_bvh2_susp, nq * sizeof(uint32_t)));
  if (!lib->d_bvh2_ctr) HIP_TRY(hipMalloc(&lib->d_bvh2_ctr, BVH_CTR_WORDS * sizeof(uint32_t)));
  lib->bvh2_split_n = nq;
  lib->bvh2_split_cap = cap;
  return HFCL_OK;
}
static int ensure_walk_streams(hfcl_lib* lib) {
  if (lib->walk_st[WALK_ROUNDS - 2]) return HFCL_OK;  // (committed last)
  hipStream_t s[WALK_ROUNDS - 1] = {};
  hipEvent_t ev[2 * (WALK_ROUNDS - 1)] = {};
  hipError_t e = hipSuccess;
  for (int k = 0; k < WALK_ROUNDS - 1 && e == hipSuccess; ++k) e = hipStreamCreateWithFlags(&s[k], hipStreamNonBlocking);
  for (int k = 0; k < 2 * (WALK_ROUNDS - 1) && e == hipSuccess; ++k) e = hipEventCreateWithFlags(&ev[k], hipEventDisableTiming);
  if (e != hipSuccess) {
    for (auto x : ev)
      if (x) hipEventDestroy(x);
    for (auto x : s)
      if (x) hipStreamDestroy(x);
    HIP_TRY(e);
  }
  for (int k = 0; k < WALK_ROUNDS - 1; ++k) {
    lib->walk_fork[k] = ev[2 * k];
    lib->walk_join[k] = ev[2 * k + 1];
  }
  for (int k = 0; k < WALK_ROUNDS - 1; ++k) lib->walk_st[k] = s[k];
  return HFCL_OK;
}
static int ensure_gjk_streams(hfcl_lib* lib) {
  if (lib->gjk_fork) return HFCL_OK;  // (committed last)
  hipStream_t s[3] = {nullptr, nullptr, nullptr};
  hipEvent_t ev[4] = {nullptr, nullptr, nullptr, nullptr};
  hipError_t e = hipSuccess;
  for (int k = 0; k < 3 && e == hipSuccess; ++k) e = hipStreamCreateWithFlags(&s[k], hipStreamNonBlocking);
  for (int k = 0; k < 4 && e == hipSuccess; ++k) e = hipEventCreateWithFlags(&ev[k], hipEventDisableTiming);
  if (e != hipSuccess) {
    for (auto x : ev)
      if (x) hipEventDestroy(x);
    for (auto x : s)
      if (x) hipStreamDestroy(x);
    HIP_TRY(e);
  }
  for (int k = 0; k < 3; ++k) {
    lib->gjk_st[k] = s[k];
    lib->gjk_join[k] = ev[k];
  }
  lib->gjk_fork = ev[3];
  return HFCL_OK;
}
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
  launch_gjk_cvx<T>(M, w, q.guess_mode == HFCL_GUESS_BOUNDING_VOLUME, int(b), st, wk, lv, io, q);
}
// (next_stream, optional: called in front of every kernel, returns the stream it goes on -- the solids' kernels of a small batch fan out)
template <typename T, class Next>
static void launch_cvx(hfcl_lib* lib, const Work& wk, const LibView<T>& lv, const IO<T>& io, const QParams<T>& q,
                       hipStream_t st, size_t& ti, size_t n, Next&& next_stream) {
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
    st = next_stream();
    tbeg("k_gjk_cvx<cc>");
    launch_cvx_m<T, 0>(lib, wk, lv, io, q, st, n);
    tend();
  }
  if ((lib->possible_buckets >> B_PC) & 1u) {
    st = next_stream();
    tbeg("k_gjk_cvx<pc>");
    launch_cvx_m<T, 1>(lib, wk, lv, io, q, st, n);
    tend();
  }
  if ((lib->possible_buckets >> B_CP) & 1u) {
    st = next_stream();
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
  const bool any_gjk_bucket = ((lib->possible_buckets >> B_PRIM) | (lib->possible_buckets >> B_CC) | (lib->possible_buckets >> B_PC) |
                               (lib->possible_buckets >> B_CP) | (lib->possible_buckets >> B_LARGE)) & 1u;
  int rc = ensure_workspace(lib, n, any_gjk_bucket && q.compute_penetration);
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
  wk.shape_defer = nullptr;
  wk.shape_defer_cap = 0;
  wk.shape_finish_over = nullptr;
  wk.shape_oq = nullptr;
  wk.epa_ready = nullptr;
  wk.epa_ready_g = nullptr;
  wk.epa_cc_over = lib->d_epa_cc_over;
  // fp32 slots are shorter than the area's stride (the fp64 slot): the slots past resume_cap are the convex x convex tier's own
  wk.cc_resume_base = wk.resume_cap;
  {
    const size_t fslots = lib->resume_cap * std::max(epa_resume_stride<double>, epa_resume_stride<float>) / epa_resume_stride<float>;
    wk.cc_resume_cap = std::is_same<T, float>::value && fslots > lib->resume_cap ? uint32_t(std::min<size_t>(fslots - lib->resume_cap, lib->resume_cap)) : 0u;
  }
  LibView<T> lv;
  lv.shapes = std::is_same<T, double>::value ? (const DShape<T>*)lib->d_shapes64 : (const DShape<T>*)lib->d_shapes32;
  lv.verts = std::is_same<T, double>::value ? (const T*)lib->d_verts64 : (const T*)lib->d_verts32;
  lv.kinds = lib->d_kinds;
  lv.n_shapes = uint32_t(lib->n_shapes);
  lv.graph_base = lib->d_graph_base;
  lv.graph_off = lib->d_graph_off;
  lv.graph_ent = std::is_same<T, double>::value ? (const NbrEntry<T>*)lib->d_graph_ent64 : (const NbrEntry<T>*)lib->d_graph_ent32;
  lv.climb_min = lib->climb_min;

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
  HIP_TRY(hipMemsetAsync(lib->d_counts, 0, N_COUNTERS * sizeof(uint32_t), st));
  tbeg("k_classify");
  launch_classify(blocks_for(n, CLS_BLOCK * 8), st, wk, lib->d_kinds, uint32_t(lib->n_shapes), q.mode != 1);
  tend();

  // the solids' kernels of the batch (closed forms, GJK; their EPA follows below, behind the mesh walks: a mesh x solid leaf can queue for it)
  auto launch_solids = [&]() -> int {
    // A small batch (option gjk_beside_max) of a library without meshes: its buckets' kernels -- independent of each other, each a chain of
    // GJK trips on a chip it does not fill -- fan out over the caller's stream and three helpers, joined in front of the EPA section
    // (cfg5 at 20 000 pairs: four GJK kernels of 40-116 us in a row).
    // From three iterative kernels on: a fork and a join cost ~0.08 ms themselves (cfg2, closed forms + one GJK kernel: 0.10 -> 0.20 ms with them).
    int kernels = 0;
    for (int b : {int(B_PRIM), int(B_CC), int(B_PC), int(B_CP), int(B_LARGE), int(B_TRI)}) kernels += may(b) ? 1 : 0;
    bool fan = lib->gjk_beside_max && n <= lib->gjk_beside_max && lib->h_meshes.empty() && kernels >= 3;
    if (fan && ensure_gjk_streams(lib) != HFCL_OK) fan = false;
    hipStream_t const caller = st;
    int fan_i = 0;
    uint32_t used = 0;
    if (fan) HIP_TRY(hipEventRecord(lib->gjk_fork, caller));
    auto next_stream = [&]() -> hipStream_t {
      if (!fan) return caller;
      const int k = fan_i;
      fan_i = (fan_i + 1) % 4;
      if (k && !(used & (1u << (k - 1)))) {
        used |= 1u << (k - 1);
        (void)hipStreamWaitEvent(lib->gjk_st[k - 1], lib->gjk_fork, 0);
      }
      return k ? lib->gjk_st[k - 1] : caller;
    };
    if (may(B_CLOSED)) {
      st = next_stream();
      tbeg("k_closed");
      launch_closed<T>(blocks_for(n, 256), st, wk, lv, io, q, lib->closed_staged);
      tend();
    }
    if (may(B_PRIM)) {
      st = next_stream();
      tbeg("k_gjk_prim");
      launch_gjk_prim<T>(blocks_for(n, 256), st, wk, lv, io, q, bvg);
      tend();
    }

    launch_cvx<T>(lib, wk, lv, io, q, caller, ti, n, next_stream);

    if (may(B_LARGE)) {
      st = next_stream();
      tbeg("k_gjk_large");
      launch_gjk_large<T>(blocks_for(n, 256 / LARGE_W), st, wk, lv, io, q, bvg);
      tend();
    }

    if (may(B_TRI)) {
      st = next_stream();
      tbeg("k_triangle");
      launch_triangle<T>(blocks_for(n / 8 + 1, 64 / BS_W), st, wk, lv, io, q);
      tend();
    }
    st = caller;
    for (int k = 0; k < 3; ++k)
      if (used & (1u << k)) {
        HIP_TRY(hipEventRecord(lib->gjk_join[k], lib->gjk_st[k]));
        HIP_TRY(hipStreamWaitEvent(caller, lib->gjk_join[k], 0));
      }
    return HFCL_OK;
  };
  // the mesh walks of the batch
  bool meshes_on_own_stream = false;
  auto launch_meshes = [&]() -> int {
    if (!lib->h_meshes.empty() && (may(B_BVH) || may(B_BVHSHAPE))) {
      // every BVH shape must name a registered model: checked on the host, the kernels index the mesh table with it
      const hfcl_lib* owner = lib;  // (helpers never run mesh batches)
      for (const hfcl_shape& sh : owner->h_shapes)
        if (sh.type == HFCL_BV_OBBRSS && (sh.bvh_index < 0 || size_t(sh.bvh_index) >= owner->h_meshes.size())) {
          set_error("BVH shape with bvh_index " + std::to_string(sh.bvh_index) + " but only " + std::to_string(owner->h_meshes.size()) +
                    " BVHModel(s) registered (hfcl_lib_add_bvh)");
          return HFCL_ERR_INVALID_ARGUMENT;
        }
      rc = upload_bvh(lib);
      if (rc) return rc;
      BvhView<T> bv;
      bv.nodes = std::is_same<T, double>::value ? (const DNode<T>*)lib->d_nodes64 : (const DNode<T>*)lib->d_nodes32;
      bv.fnodes = (std::is_same<T, double>::value && lib->bvh_filter) ? lib->d_fnodes : nullptr;
      bv.rss = std::is_same<T, double>::value ? (const DRss<T>*)lib->d_rss64 : (const DRss<T>*)lib->d_rss32;
      bv.dnodes = std::is_same<T, double>::value ? (const DNodeD<T>*)lib->d_dnodes64 : (const DNodeD<T>*)lib->d_dnodes32;
      bv.verts = std::is_same<T, double>::value ? (const T*)lib->d_bverts64 : (const T*)lib->d_bverts32;
      bv.tris = lib->d_btris;
      bv.meshes = lib->d_meshes;
      bv.n_meshes = uint32_t(lib->h_meshes.size());
      BvhSpill spill;
      rc = make_bvh_spill(lib, spill, q.mode != 1);
      if (rc) return rc;
      // long traversals are cut into tasks when the batch is large enough for the tail to matter and the request keeps no
      // query-wide contact count (mesh x mesh and the one-query-per-lane form of mesh x solid alike)
      auto make_split = [&](BvhSplit& split, bool want, bool solid, bool own_tables = false) -> int {
        memset(&split, 0, sizeof(split));
        split.leaf_cost = lib->shape_leaf_cost;
        if (!(want && (solid ? lib->shape_levels : lib->bvh_levels) > 1 && lib->bvh_params.num_max_contacts == 1 && !lib->bvh_params.contacts)) return HFCL_OK;
        int r = own_tables ? ensure_bvh_split2(lib, n) : ensure_bvh_split(lib, n);
        if (r) return r;
        HIP_TRY(hipMemsetAsync(own_tables ? lib->d_bvh2_ctr : lib->d_bvh_ctr, 0, BVH_CTR_WORDS * sizeof(uint32_t), st));
        split.tasks = own_tables ? lib->d_bvh2_tasks : lib->d_bvh_tasks;
        split.sums = own_tables ? lib->d_bvh2_sums : lib->d_bvh_sums;
        split.suspended = own_tables ? lib->d_bvh2_susp : lib->d_bvh_susp;
        split.ctr = own_tables ? lib->d_bvh2_ctr : lib->d_bvh_ctr;
        split.cap = uint32_t(std::min<size_t>(own_tables ? lib->bvh2_split_cap : lib->bvh_split_cap, 0x7FFFFFFFu));
        split.n_queries = uint32_t(own_tables ? lib->bvh2_split_n : lib->bvh_split_n);
        split.budget = lib->bvh_budget;
        split.budget0 = lib->bvh_budget0;
        split.n_levels = lib->bvh_levels;
        if (lib->bvh_auto && 2 * n <= 3 * size_t(lib->n_cus) * 512) {  // (8 waves of 64 lanes per CU are resident)
          split.budget0 = 512;
          split.budget = 16;
          split.n_levels = BVH_MAX_LEVELS;
        }
        split.coop_grid = uint32_t(lib->n_cus) * 8u;
        split.cut_ticks = solid ? lib->shape_cut_ticks : (own_tables ? 0u : lib->bvh_cut_ticks);  // (the second set has no chunk tables)
        split.cut_cap = split.cap;
        // (mesh x solid: the EPA queue has room for one item per query and per chunk -- shape_defer_cap entries, sized before the walk)
        split.cut_task_cap = solid ? (wk.shape_defer_cap > n ? uint32_t(std::min<size_t>(wk.shape_defer_cap - n, split.cap)) : 0u) : split.cap;
        split.cut_words = lib->d_bvh_cut_words;
        split.cut_vals = lib->d_bvh_cut_vals;
        if (!solid && lib->bvh_coop) {
          split.coop = 1u;
          const bool one_round = lib->walk_auto && n <= 220000;
          const uint32_t rounds = lib->walk_auto ? (one_round ? 1u : 2u) : lib->walk_rounds;
          split.budget0 = lib->bvh_budget0_coop ? lib->bvh_budget0_coop
                                                : (n > 500000 ? 640u : (one_round ? (n > 120000 ? 320u : 256u) : (rounds ? (n > 150000 ? std::max(320u, lib->walk_budget[0]) : lib->walk_budget[0]) : 256u)));
          // the queries' own phase as walk / leaves / resolve rounds (narrow node ids; rec indices travel in 28 bits)
          if (rounds && n < (size_t(1) << 28)) {
            r = ensure_walk(lib, n);
            if (r) return r;
            HIP_TRY(hipMemsetAsync(lib->d_walk_ctr, 0, 8 * WALK_ROUNDS * sizeof(uint32_t), st));
            split.walk.recs = lib->d_walk_recs;
            split.walk.items = lib->d_walk_items;
            split.walk.res = lib->d_walk_res;
            split.walk.ctr = lib->d_walk_ctr;
            split.walk.list_in = lib->d_walk_lists;
            split.walk.list_out = lib->d_walk_lists;
            split.walk.item_cap = uint32_t(std::min<size_t>(lib->walk_n * WALK_K, 0xFFFFFFFFu));
            split.walk.list_stride = uint32_t(lib->walk_n);
            split.order = lib->walk_order ? lib->d_walk_order : nullptr;
            split.walk_rounds = std::min<uint32_t>(rounds, WALK_ROUNDS);
            for (int k = 0; k < WALK_ROUNDS; ++k) {
              split.walk_k[k] = (one_round && k == 0) ? uint32_t(WALK_K) : lib->walk_k[k];
              split.walk_budget[k] = k == 0 ? split.budget0 : lib->walk_budget[k];
            }
          }
        }
        if (solid) {
          split.coop = lib->shape_coop ? 1u : 0u;
          split.budget0 = lib->shape_coop ? lib->shape_budget0_coop : lib->shape_budget0;
          split.budget = lib->shape_budget;
          split.n_levels = lib->shape_levels;
          // the queries' own phase as walk / leaves / resolve (one round; what is left of a walk is k_bvh_shape_coop's)
          if (split.coop && lib->shape_walk && n >= lib->shape_walk_min && n < (size_t(1) << 28)) {
            r = ensure_swalk(lib, n);
            if (r) return r;
            HIP_TRY(hipMemsetAsync(lib->d_swalk_ctr, 0, (8 * WALK_ROUNDS + 64) * sizeof(uint32_t), st));
            split.walk.hist = lib->d_swalk_ctr + 8 * WALK_ROUNDS;
            split.walk.perm = lib->shape_walk_sort ? lib->d_swalk_perm : nullptr;
            split.walk.recs = lib->d_swalk_recs;
            split.walk.items = lib->d_swalk_items;
            split.walk.res = lib->d_swalk_res;
            split.walk.ctr = lib->d_swalk_ctr;
            split.walk.list_in = lib->d_swalk_lists;
            split.walk.list_out = lib->d_swalk_lists;
            split.walk.redo = lib->d_swalk_lists + 2 * lib->swalk_n;
            split.walk.item_cap = uint32_t(std::min<size_t>(lib->swalk_n * WALK_K, 0xFFFFFFFFu));
            split.walk.list_stride = uint32_t(lib->swalk_n);
            split.walk_rounds = 1u;
            split.walk_k[0] = uint32_t(WALK_K);
            split.walk_budget[0] = lib->shape_walk_budget;
          }
        }
        return HFCL_OK;
      };
      // mesh x solid: one query per lane (k_bvh_collide's SOLID form) where the request lets a leaf that needs EPA end the
      // walk (hfcl_bvh_shape.hpp: mesh_shape_lane_request) and the lanes' stacks hold the models; the 16-lane group kernel
      // otherwise
      const bool shape_fast = q.mode == 1 && may(B_BVHSHAPE) && lib->bvh_shape_lane && size_t(lib->bvh_max_depth) + 1 <= size_t(BVH_STACK) &&
                              mesh_shape_lane_request(q, lib->bvh_params.num_max_contacts);
      // distance(): a leaf that needs EPA always ends the walk; models deeper than the lanes' stacks take the group kernel
      const bool shape_fast_d = q.mode != 1 && may(B_BVHSHAPE) && lib->bvh_shape_lane && size_t(lib->bvh_max_depth) + 1 <= size_t(BVHD_STACK);
      if (shape_fast || shape_fast_d) {
        // one EPA item per unit at most (a contact ends the unit): a query, or -- when suspended walks are cut into task levels
        // instead of being continued by a wave (HFCL_SHAPE_COOP=0) -- every task of the split's table as well
        size_t need = lib->ws_capacity;
        // (the chunks of a cut walk are units too, and every unit can queue one item: room for four chunks per query -- 336 B each --; a walk
        // whose chunks would not fit is not cut, BvhSplit::cut_task_cap.  cfg4s makes ~0.6 chunks per query; n / 2 was too tight: cuts refused,
        // 3.5 -> 4.4 ms)
        if (shape_fast && lib->shape_coop && lib->shape_cut_ticks) need = std::max(need, n + 4 * n + 4096);
        if (shape_fast && !lib->shape_coop && n >= 256) {
          rc = ensure_bvh_split(lib, n);
          if (rc) return rc;
          need = std::max(need, n + lib->bvh_split_cap);
        }
        if (need > lib->shape_defer_capacity) {
          hipFree(lib->d_shape_defer);
          lib->d_shape_defer = nullptr;
          lib->shape_defer_capacity = 0;
          HIP_TRY(hipMalloc(&lib->d_shape_defer, need * (sizeof(ShapeDeferItem<double>) + 2 * sizeof(uint32_t))));  // (+ the two lists of k_bvh_shape_finish's second tier)
          lib->shape_defer_capacity = need;
        }
        // (the solids' boxes are indexed by pair: one per pair of the workspace, not one per EPA item -- 1M pairs: 0.14 GB instead of 0.7)
        if (lib->ws_capacity > lib->shape_oq_capacity) {
          hipFree(lib->d_shape_oq);
          lib->d_shape_oq = nullptr;
          lib->shape_oq_capacity = 0;
          HIP_TRY(hipMalloc(&lib->d_shape_oq, lib->ws_capacity * std::max(sizeof(ObbQuery<double>), sizeof(RssQuery<double>))));
          lib->shape_oq_capacity = lib->ws_capacity;
        }
        wk.shape_defer = lib->d_shape_defer;
        wk.shape_defer_cap = uint32_t(std::min<size_t>(lib->shape_defer_capacity, 0xFFFFFFFFu));
        wk.shape_finish_over = lib->shape_finish_tiers ? reinterpret_cast<uint32_t*>(static_cast<char*>(lib->d_shape_defer) + lib->shape_defer_capacity * sizeof(ShapeDeferItem<double>)) : nullptr;
        wk.shape_oq = lib->d_shape_oq;
      }
      if (q.mode == 1) {
        // Both kinds of mesh pairs in the batch's library, and the mesh walks on streams of their own: the mesh x mesh walks (tables of
        // their own) on a second one, beside the mesh x solid walks -- the two share nothing else, and each is a chain that leaves the chip
        // half empty (cfgmix: 1.45 ms of mesh x mesh behind 2.3 ms of mesh x solid)
        const bool mm_beside = meshes_on_own_stream && lib->mesh_beside >= 2 && may(B_BVH) && may(B_BVHSHAPE) && !spill.wide && !lib->bvh_cut_ticks && lib->bvh_coop && (lib->walk_auto || lib->walk_rounds != 0);
        auto mesh_mesh = [&](bool own_tables) -> int {
          tbeg("k_bvh_collide");
          BvhSplit split;
          rc = make_split(split, may(B_BVH) && !spill.wide && (n >= 256 || 2 * size_t(lib->bvh_max_depth) + 4 > size_t(std::min(BVH_STACK, BVH_STACK_FILT))), false, own_tables);
          if (rc) return rc;
          AsideStream beside[WALK_ROUNDS - 1];
          memset(beside, 0, sizeof(beside));
          const bool early = split.walk.recs && split.walk_rounds > 1 && lib->walk_early_coop;
          if (early) {
            rc = ensure_walk_streams(lib);
            if (rc) return rc;
            for (int k = 0; k < WALK_ROUNDS - 1; ++k) beside[k] = AsideStream{lib->walk_st[k], lib->walk_fork[k], lib->walk_join[k]};
          }
          launch_bvh_collide<T>(blocks_for(n, BVH_BLOCK), st, wk, lv, bv, io, q, lib->bvh_params, T(lib->break_distance * lib->break_distance), split, spill, early ? beside : nullptr);
          tend();
          return HFCL_OK;
        };
        auto mesh_mesh_beside = [&]() -> int {
          hipStream_t const ms = st;
          st = lib->mesh_st2;
          rc = mesh_mesh(true);
          st = ms;
          if (rc) return rc;
          HIP_TRY(hipEventRecord(lib->ev_mesh_join2, lib->mesh_st2));
          return HFCL_OK;
        };
        if (mm_beside) {
          HIP_TRY(hipEventRecord(lib->ev_mesh_fork2, st));
          HIP_TRY(hipStreamWaitEvent(lib->mesh_st2, lib->ev_mesh_fork2, 0));
          rc = mesh_mesh_beside();
          if (rc) return rc;
        }
        tbeg("k_bvh_shape");
        if (shape_fast) {
          // tasks re-start the leaf solver from the request's guess: a walk whose leaves hand the cached guess on, or whose
          // final guess is read, stays in one piece
          BvhSplit split;
          rc = make_split(split, n >= 256 && q.guess_mode != HFCL_GUESS_CACHED && !io.gout, true);
          if (rc) return rc;
          AsideStream aside = {nullptr, nullptr, nullptr};
          if (lib->shape_finish_aside && wk.shape_finish_over && split.tasks && split.coop && split.cut_ticks) {
            rc = ensure_aux(lib);
            if (rc) return rc;
            aside = AsideStream{meshes_on_own_stream ? lib->mesh_aux : lib->aux, lib->ev_aux2, lib->ev_aux3};  // (`aux` is the solids' EPA section's, beside)
          }
          launch_bvh_shape_fast<T>(blocks_for(n, BVH_BLOCK), blocks_for(n / 8 + 1, 64 / BS_W), int(std::min<size_t>(n / 4 + 1, size_t(lib->n_cus) * 8)), st, wk, lv, bv, io, q, lib->bvh_params,
                                   T(lib->break_distance * lib->break_distance), split, spill, aside.stream ? &aside : nullptr);
        } else {
          launch_bvh_shape<T>(blocks_for(n / 8 + 1, 64 / BS_W), st, wk, lv, bv, io, q, lib->bvh_params, T(lib->break_distance * lib->break_distance));
        }
        tend();
        if (mm_beside) {
          HIP_TRY(hipStreamWaitEvent(st, lib->ev_mesh_join2, 0));
        } else {
          rc = mesh_mesh(false);
          if (rc) return rc;
        }
      } else {
        tbeg("k_bvh_shape_distance");
        if (shape_fast_d) {
          // long walks are handed to waves -- unless their leaves hand a cached guess on, or the final guess is read
          BvhSpill ss;
          memset(&ss, 0, sizeof(ss));
          if (lib->shape_dist_budget && q.guess_mode != HFCL_GUESS_CACHED && !io.gout) {
            if (lib->ws_capacity > lib->shape_dist_susp_capacity) {
              hipFree(lib->d_shape_dist_susp);
              lib->d_shape_dist_susp = nullptr;
              lib->shape_dist_susp_capacity = 0;
              HIP_TRY(hipMalloc(&lib->d_shape_dist_susp, lib->ws_capacity * sizeof(ShapeDistSusp<double>)));
              lib->shape_dist_susp_capacity = lib->ws_capacity;
            }
            ss.susp = lib->d_shape_dist_susp;
            ss.rerun_count = lib->pool_rerun ? lib->d_counts + CTR_SHAPE_DIST_RERUN : nullptr;
            ss.rerun_all = lib->pool_rerun >= 2 ? 1u : 0u;
            ss.susp_count = lib->d_counts + CTR_SHAPE_DIST_SUSP;
            ss.budget = lib->shape_dist_budget;
            ss.max_blocks = uint32_t(lib->n_cus) * 8u;
            ss.pool = lib->has_flats ? 0u : lib->shape_dist_pool;
            ss.pool_ticket = lib->d_counts + CTR_SHAPE_DIST_TICKET;
            ss.pool_leaf_min = lib->shape_dist_leaf_min;
            ss.pool_starve = lib->shape_dist_starve;
          }
          launch_bvh_shape_distance_fast<T>(blocks_for(n, BVHD_BLOCK), blocks_for(n / 8 + 1, 64 / BS_W), st, wk, lv, bv, io, q, ss);
        }
        else
          launch_bvh_shape_distance<T>(blocks_for(n / 8 + 1, 64 / BS_W), st, wk, lv, bv, io, q);
        tend();
        tbeg("k_bvh_distance");
        if (may(B_BVH) && !spill.wide && lib->bvhd_budget) {
          if (lib->ws_capacity > lib->dist_susp_capacity) {
            hipFree(lib->d_dist_susp);
            lib->d_dist_susp = nullptr;
            lib->dist_susp_capacity = 0;
            HIP_TRY(hipMalloc(&lib->d_dist_susp, lib->ws_capacity * sizeof(DistSusp<double>)));
            lib->dist_susp_capacity = lib->ws_capacity;
          }
          spill.susp = lib->d_dist_susp;
          spill.rerun_count = lib->pool_rerun ? lib->d_counts + CTR_DIST_RERUN : nullptr;
          spill.rerun_all = lib->pool_rerun >= 2 ? 1u : 0u;
          spill.susp_count = lib->d_counts + CTR_DIST_SUSP;
          spill.budget = lib->bvhd_budget;
          spill.max_blocks = uint32_t(lib->n_cus) * 8u;
          spill.pool = lib->bvhd_pool;
          spill.pool_ticket = lib->d_counts + CTR_DIST_TICKET;
          spill.pool_leaf_min = lib->bvhd_pool_leaf_min;
          spill.pool_starve = lib->bvhd_pool_starve;
          spill.pool_part_min = lib->bvhd_pool_part_min;
          if (lib->bvh_max_nodes > 32767) spill.pool = 0;  // 15-bit node ids in the pool's entry word (POOL_MAX_NODES)
        }
        launch_bvh_distance<T>(blocks_for(n, BVHD_BLOCK), st, wk, lv, bv, io, q, spill);
        tend();
      }
    }
    return HFCL_OK;
  };
  bool mesh_join_pending = false;
  // A library with meshes AND solids: the mesh walks on a stream of their own BESIDE the solids' kernels -- the walks are chains of dependent
  // steps that leave the chip half empty (section 3 item 6f), and every kernel of a bucket the library COULD fill is launched whether or not
  // this batch fills it (a mesh-only batch of a mixed library used to wait for ~0.08 ms of empty GJK launches in front of its walks).
  {
    const bool any_mesh = !lib->h_meshes.empty() && (may(B_BVH) || may(B_BVHSHAPE));
    const bool any_solid = may(B_CLOSED) || any_gjk || may(B_TRI);
    bool beside = any_mesh && any_solid && lib->mesh_beside;
    // ... when the batch holds both: a batch of mesh pairs alone pays for the solids' empty launches when they stand BESIDE its walks (grids sized
    // for the batch, every block waiting for a wave slot of a full chip: cfg4s 2.67 -> 2.80 ms) and nothing when they stand in front of them
    // (4 us each on an empty chip).  The witness is the library's batch before this one (its bucket counts, read without waiting for them:
    // they only choose between two orders of the same launches); the first batch runs beside.
    if (beside && lib->mesh_beside < 4 && lib->ran_batch && lib->h_counts) {
      uint32_t solids_before = 0, meshes_before = one_count(lib->h_counts, int(B_BVH)) + one_count(lib->h_counts, int(B_BVHSHAPE));
      for (int b : {int(B_CLOSED), int(B_PRIM), int(B_CC), int(B_PC), int(B_CP), int(B_LARGE), int(B_TRI)}) solids_before += one_count(lib->h_counts, b);
      if (!solids_before || !meshes_before) beside = false;
    }
    if (beside && ensure_mesh_stream(lib) != HFCL_OK) beside = false;  // (no helper stream: one after the other, as before)
    if (beside) {
      hipStream_t const caller = st;
      HIP_TRY(hipEventRecord(lib->ev_mesh_fork, caller));
      HIP_TRY(hipStreamWaitEvent(lib->mesh_st, lib->ev_mesh_fork, 0));
      st = lib->mesh_st;  // (the lambdas above launch on `st`)
      meshes_on_own_stream = true;
      rc = launch_meshes();
      st = caller;
      if (rc) return rc;
      HIP_TRY(hipEventRecord(lib->ev_mesh_join, lib->mesh_st));
      rc = launch_solids();
      if (rc) return rc;
      if (lib->mesh_beside < 2) HIP_TRY(hipStreamWaitEvent(caller, lib->ev_mesh_join, 0));
      else mesh_join_pending = true;  // (the solids' EPA section first: no mesh kernel feeds its queues; joined in front of the batch's last launches)
    } else {
      rc = launch_solids();
      if (rc) return rc;
      rc = launch_meshes();
      if (rc) return rc;
    }
  }

  if (q.compute_penetration && any_gjk) {
    // ---- EPA.  Fast tiers in three stages (hfcl_k_epa.hip) for batches large enough to pay for the extra launches: one lane per polytope
    // prepares it (encloseOrigin, first tetrahedron) and writes its record; the loop kernels between them do nothing but expand.
    //   fp32 convex x convex (the top queue): k_epa_prepare / k_epa_loop / k_epa_records, k_epa_resume_cc for the polytopes that outgrow the block
    //   every other queue, both precisions:    k_epa_prepare_general / k_epa_loop_general / k_epa_records_general, the full-capacity tier behind them
    // Otherwise the one-kernel forms (launch_epa_fast: fp32 streams, fp64 lockstep kernels).
    constexpr bool F32 = std::is_same<T, float>::value;
    const bool general_q = may(B_PRIM) || may(B_PC) || may(B_CP) || (!F32 && may(B_CC));
    bool cc_staged = false;
    if constexpr (F32) cc_staged = may(B_CC) && lib->epa_cc_staged && n >= lib->epa_cc_staged_min;
    bool gen_staged = general_q && lib->epa_general_staged && n >= lib->epa_general_staged_min;
    // A very small batch: the full-capacity tier alone, over every seed (k_epa_requeue) -- the batch is as long as its longest polytope either way, and
    // the fast tier in front of the full one is a second such chain (cfg5's mix at 2 000 pairs: 0.18 + 0.22 ms)
    // (fp64 only: its tiers are compiled without contraction and agree bit for bit; the fp32 tiers are different instantiations of contracted code)
    const bool direct = sizeof(T) == 8 && lib->epa_direct_max && n <= lib->epa_direct_max;
    auto need_aux = [&]() -> int { return ensure_aux(lib); };
    auto tbeg_on = [&](const char* name, hipStream_t s) {
      if (!lib->kernel_timing) return;
      t = timer_slot(lib, ti++, name);
      hipEventRecord(t->e0, s);
    };
    auto tend_on = [&](hipStream_t s) {
      if (lib->kernel_timing) hipEventRecord(t->e1, s);
    };
    if (direct) cc_staged = gen_staged = false;
    if (cc_staged) {
      if (lib->ws_capacity > lib->epa_ready_capacity) {
        hipFree(lib->d_epa_ready);
        lib->d_epa_ready = nullptr;
        lib->epa_ready_capacity = 0;
        HIP_TRY(hipMalloc(&lib->d_epa_ready, lib->ws_capacity * sizeof(EpaReady<float>)));
        lib->epa_ready_capacity = lib->ws_capacity;
      }
      wk.epa_ready = lib->d_epa_ready;
    }
    if (gen_staged) {
      if (lib->ws_capacity * sizeof(EpaReadyG<T>) > lib->epa_ready_g_bytes) {
        hipFree(lib->d_epa_ready_g);
        lib->d_epa_ready_g = nullptr;
        lib->epa_ready_g_bytes = 0;
        HIP_TRY(hipMalloc(&lib->d_epa_ready_g, lib->ws_capacity * sizeof(EpaReadyG<T>)));
        lib->epa_ready_g_bytes = lib->ws_capacity * sizeof(EpaReadyG<T>);
      }
      wk.epa_ready_g = lib->d_epa_ready_g;
    }
    if constexpr (F32) {
      if (cc_staged) {
        tbeg("k_epa_prepare");
        launch_epa_prepare(blocks_for(n / 4 + 1, 256), st, wk, lv, io, q);
        tend();
      }
    }
    if (gen_staged) {
      tbeg("k_epa_prepare_general");
      launch_epa_prepare_general<T>(blocks_for(n / 4 + 1, 256), st, wk, lv, io, q, F32);
      tend();
    }
    if (direct) {
      tbeg("k_epa<full>");
      launch_epa_requeue<T>(st, wk);
      // (the grid: a lane group per seed, up to what k_epa's shape-0 support point area holds -- n_cus * 16 blocks)
      launch_epa_full<T>(int(std::min<size_t>(blocks_for(n / 2 + 1, 64 / epa_we2<T>), size_t(lib->n_cus) * 16)), st, wk, lv, io, q);
      tend();
    } else {
    const int epa_batches = int(std::min<size_t>((n + 64 / EPA_WE - 1) / (64 / EPA_WE), size_t(1) << 22));
    tbeg("k_epa<fast>");
    // fp64 with both classes of pairs: their fast-tier kernels on two streams (each one's tail under the other's body)
    hipStream_t st2 = nullptr;
    if constexpr (!F32) {
      if (lib->epa64_two_streams && lib->has_curved && general_q) {
        if (int rc2 = need_aux()) return rc2;
        st2 = lib->aux;
        HIP_TRY(hipEventRecord(lib->ev_aux0, st));
        HIP_TRY(hipStreamWaitEvent(st2, lib->ev_aux0, 0));
      }
    }
    // (the launchers size the grids of the persistent forms themselves: here only the number of wave-sized batches)
    if constexpr (F32) {
      if (cc_staged) launch_epa_loop(epa_batches, st, wk, lv, q, lib->n_cus);
    }
    if (gen_staged) launch_epa_loop_general<T>(epa_batches, st, st2, wk, lv, q, lib->n_cus, lib->has_curved);
    if ((F32 && may(B_CC) && !cc_staged) || (general_q && !gen_staged))
      launch_epa_fast<T>(epa_batches, st, wk, lv, io, q, may(B_CC) && !cc_staged, general_q && !gen_staged, lib->n_cus, lib->has_curved, st2);
    if (st2) {
      HIP_TRY(hipEventRecord(lib->ev_aux1, st2));
      HIP_TRY(hipStreamWaitEvent(st, lib->ev_aux1, 0));
    }
    tend();
    if (cc_staged || gen_staged) {
      // What ends the batch: the records of the finished polytopes (bound by memory), the continuation of the handed-over ones
      // (k_epa_resume_cc; as long as its longest chain of iterations) and the full-capacity tier (likewise).  The records run on a stream of
      // their own beside the latter two (with a convex x convex tier the full-capacity tier joins them there, beside the continuation).
      const bool aside = lib->records_aside;
      if (aside) {
        if (int rc2 = need_aux()) return rc2;
        HIP_TRY(hipEventRecord(lib->ev_aux0, st));
        HIP_TRY(hipStreamWaitEvent(lib->aux, lib->ev_aux0, 0));
      }
      hipStream_t rs = aside ? lib->aux : st;
      tbeg_on("k_epa_records", rs);
      if constexpr (F32) {
        if (cc_staged) launch_epa_records(blocks_for(n / 4 + 1, 256), rs, wk, lv, io, q);
      }
      if (gen_staged) launch_epa_records_general<T>(blocks_for(n / 4 + 1, 256), rs, wk, lv, io, q, F32);
      tend_on(rs);
      // (without a continuation kernel of its own the batch's stream takes the full-capacity tier)
      hipStream_t fs = cc_staged ? rs : st;
      tbeg_on("k_epa<full>", fs);
      launch_epa_full<T>(blocks_for(n / 16 + 1, 64 / epa_we2<T>), fs, wk, lv, io, q);
      tend_on(fs);
      if (aside) HIP_TRY(hipEventRecord(lib->ev_aux1, rs));
      if constexpr (F32) {
        if (cc_staged) {
          tbeg("k_epa_resume_cc");
          launch_epa_resume_cc(blocks_for(n / 16 + 1, 64 / HFCL_EPA_CC_RESUME_WE), st, wk, lv, io, q);
          tend();
        }
      }
      if (aside) HIP_TRY(hipStreamWaitEvent(st, lib->ev_aux1, 0));
    } else {
      tbeg("k_epa<full>");
      launch_epa_full<T>(blocks_for(n / 16 + 1, 64 / epa_we2<T>), st, wk, lv, io, q);
      tend();
    }
    }
  }
  // last: a launch of a few waves that, between the GJK and the EPA kernels, only waited for a free CU while the other
  // half of a split batch had the chip (0.2 ms of this stream's timeline on cfg5)
  if (mesh_join_pending) HIP_TRY(hipStreamWaitEvent(st, lib->ev_mesh_join, 0));
  tbeg("k_unsupported");
  if (may(B_UNSUPPORTED)) launch_unsupported<T>(blocks_for(n, 256 * 64), st, wk, io, int(B_UNSUPPORTED));
  if (lib->h_meshes.empty()) {  // BVH shapes without any registered mesh: flagged, never left unwritten
    if (may(B_BVHSHAPE)) launch_unsupported<T>(blocks_for(n, 256 * 64), st, wk, io, int(B_BVHSHAPE));
    if (may(B_BVH)) launch_unsupported<T>(blocks_for(n, 256 * 64), st, wk, io, int(B_BVH));
  }
  tend();
  HIP_TRY(hipMemcpyAsync(lib->counts_dst ? lib->counts_dst : lib->h_counts, lib->d_counts, N_COUNTERS * sizeof(uint32_t), hipMemcpyDeviceToHost, st));
  lib->ran_batch = lib->counts_dst == nullptr;
  HIP_TRY(hipGetLastError());
  return HFCL_OK;
}

// shallow clone for the second half of a split batch: shares the device shape tables, owns everything else
// the device tables a split batch's second half shares with its owner
static void share_tables(hfcl_lib* h, const hfcl_lib* lib) {
  h->n_shapes = lib->n_shapes;
  h->d_shapes64 = lib->d_shapes64;
  h->d_shapes32 = lib->d_shapes32;
  h->d_verts64 = lib->d_verts64;
  h->d_verts32 = lib->d_verts32;
  h->d_kinds = lib->d_kinds;
  h->possible_buckets = lib->possible_buckets;
  h->has_curved = lib->has_curved;
  h->has_flats = lib->has_flats;
  h->d_graph_base = lib->d_graph_base;
  h->d_graph_off = lib->d_graph_off;
  h->d_graph_ent32 = lib->d_graph_ent32;
  h->d_graph_ent64 = lib->d_graph_ent64;
  h->climb_min = lib->climb_min;
}
static hfcl_lib* make_helper(hfcl_lib* lib) {
  hfcl_lib* h = new hfcl_lib;
  h->is_helper = true;
  h->device = lib->device;
  share_tables(h, lib);
  h->cvx_w = lib->cvx_w;
  h->epa_resume_slots = lib->epa_resume_slots;
  h->closed_staged = lib->closed_staged;
  h->epa_cc_staged = lib->epa_cc_staged;
  h->records_aside = lib->records_aside;
  h->epa_general_staged = lib->epa_general_staged;
  h->shape_finish_tiers = lib->shape_finish_tiers;
  h->shape_finish_aside = lib->shape_finish_aside;
  h->shape_cut_ticks = lib->shape_cut_ticks;
  h->epa_general_staged_min = lib->epa_general_staged_min;
  h->epa64_two_streams = lib->epa64_two_streams;
  h->epa_cc_staged_min = lib->epa_cc_staged_min;
  h->n_cus = lib->n_cus;
  bool ok = hipMalloc(&h->d_counts, N_COUNTERS * sizeof(uint32_t)) == hipSuccess;
  ok = ok && hipHostMalloc((void**)&h->h_counts, N_COUNTERS * sizeof(uint32_t), hipHostMallocDefault) == hipSuccess;
  ok = ok && hipMalloc(&h->d_epa_v0, size_t(h->n_cus) * 16 * (64 / EPA_WE2) * EPA_MAX_VERTS * sizeof(Quad<double>)) == hipSuccess;
  if (!ok) {
    hfcl_lib_destroy(h);
    return nullptr;
  }
  memset(h->h_counts, 0, N_COUNTERS * sizeof(uint32_t));
  return h;
}

template <typename T> static IO<T> io_at(const IO<T>& io, size_t lo);
template <> IO<double> io_at(const IO<double>& io, size_t lo) {
  return IO<double>{io.tf1 + 12 * lo, io.tf2 + 12 * lo, io.out + lo, io.gin ? io.gin + lo : nullptr, io.gout ? io.gout + lo : nullptr};
}
template <> IO<float> io_at(const IO<float>& io, size_t lo) {
  return IO<float>{io.tf1 + 7 * lo, io.tf2 + 7 * lo, io.out + lo, nullptr, nullptr};
}

// Does a batch of n pairs of this library run as two halves on two streams?  Automatic choice: a library whose pairs
// spread over three or more of the iterative buckets (mixed scenes: cfg5 4.05 -> 3.80 ms) -- the halves then run different
// kernels side by side; with one or two kernels in the batch the halves only share the machine phase by phase and the
// doubled fixed costs lose 3 % (cfg2, cfg3).  A/B in profiles/r01_k_two_stream_overlap.txt.
static bool batch_splits(const hfcl_lib* lib, size_t n) {
  constexpr size_t MIN_SPLIT = 1u << 17;
  int parts = lib->split;
  if (parts == 0) {
    int kinds = 0;
    for (int b : {int(B_PRIM), int(B_CC), int(B_PC), int(B_CP), int(B_LARGE)}) kinds += (lib->possible_buckets >> b) & 1u;
    parts = kinds >= 3 ? 2 : 1;
  }
  // meshes keep query-wide side state (contact lists, pair ids in them): they run unsplit
  return parts >= 2 && n >= MIN_SPLIT && lib->h_meshes.empty();
}
static int ensure_helper(hfcl_lib* lib) {
  if (lib->helper) return HFCL_OK;
  HIP_TRY(hipSetDevice(lib->device));
  lib->helper = make_helper(lib);
  bool ok = lib->helper != nullptr;
  ok = ok && (lib->side || hipStreamCreateWithFlags(&lib->side, hipStreamNonBlocking) == hipSuccess);
  ok = ok && (lib->ev_fork || hipEventCreateWithFlags(&lib->ev_fork, hipEventDisableTiming) == hipSuccess);
  ok = ok && (lib->ev_join || hipEventCreateWithFlags(&lib->ev_join, hipEventDisableTiming) == hipSuccess);
  if (!ok) {  // leave nothing half-made behind: the next call retries cleanly
    if (lib->helper) hfcl_lib_destroy(lib->helper);
    lib->helper = nullptr;
    set_error("split batches: HIP allocation failed");
    return HFCL_ERR_HIP;
  }
  return HFCL_OK;
}

template <typename T>
static int run_batch(hfcl_lib* lib, const uint32_t* d_s1, const uint32_t* d_s2, IO<T> io, size_t n, QParams<T> q,
                     hipStream_t st) {
  lib->last_split = false;
  if (!lib->in_host_batch) lib->last_host = false;
  if (lib->graph_dirty) {
    const int rcg = upload_graph(lib);
    if (rcg) return rcg;
  }
  if (!batch_splits(lib, n)) return run_batch_one<T>(lib, d_s1, d_s2, io, n, q, st);
  int rc0 = ensure_helper(lib);
  if (rc0) return rc0;
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

// full records -> compact records (hfcl_result_compact), for the multi-GPU exchange of results
template <typename R, typename C>
static int compact_results(hfcl_lib* lib, const R* d_records, size_t n, C* d_out, void* stream) {
  if (!lib) {
    set_error("null library");
    return HFCL_ERR_INVALID_ARGUMENT;
  }
  if (n == 0) return HFCL_OK;
  if (!d_records || !d_out) {
    set_error("hfcl_compact_results_device: null buffer");
    return HFCL_ERR_INVALID_ARGUMENT;
  }
  if (n > 0xFFFFFFF0ull) {
    set_error("batch too large (max 2^32-16 pairs per call)");
    return HFCL_ERR_LIMIT;
  }
  HIP_TRY(hipSetDevice(lib->device));
  launch_compact_records((hipStream_t)stream, d_records, d_out, uint32_t(n));
  HIP_TRY(hipGetLastError());
  return HFCL_OK;
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
    if (n) launch_fill_skipped(st, d_out, uint32_t(n));
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
    if (n) launch_fill_skipped(st, d_out, uint32_t(n));
    return HFCL_OK;
  }
  IO<float> io{d_pose1, d_pose2, d_out, nullptr, nullptr};
  lib->bvh_params.num_max_contacts = req->num_max_contacts;
  lib->break_distance = req->break_distance;
  return run_batch<float>(lib, d_shape1, d_shape2, io, n, q, st);
}

int hfcl_compact_results_device(hfcl_lib* lib, const hfcl_result* d_records, size_t n, hfcl_result_compact* d_out, void* stream) {
  return compact_results(lib, d_records, n, d_out, stream);
}
int hfcl_compact_results_device_f32(hfcl_lib* lib, const hfcl_result_f32* d_records, size_t n, hfcl_result_compact_f32* d_out,
                                    void* stream) {
  return compact_results(lib, d_records, n, d_out, stream);
}

static int ensure_staging(hfcl_lib* lib, size_t n, bool gin, bool gout, bool compact) {
  if (!lib->s_cmp) {
    HIP_TRY(hipStreamCreateWithFlags(&lib->s_h2d, hipStreamNonBlocking));
    HIP_TRY(hipStreamCreateWithFlags(&lib->s_h2d2, hipStreamNonBlocking));
    HIP_TRY(hipStreamCreateWithFlags(&lib->s_cmp, hipStreamNonBlocking));
    HIP_TRY(hipStreamCreateWithFlags(&lib->s_d2h, hipStreamNonBlocking));
    for (auto& sg : lib->stage) {
      HIP_TRY(hipEventCreateWithFlags(&sg.ev_in, hipEventDisableTiming));
      HIP_TRY(hipEventCreateWithFlags(&sg.ev_done, hipEventDisableTiming));
      HIP_TRY(hipEventCreateWithFlags(&sg.ev_in2, hipEventDisableTiming));
      HIP_TRY(hipHostMalloc((void**)&sg.h_counts, N_COUNTERS * sizeof(uint32_t), hipHostMallocDefault));
      HIP_TRY(hipHostMalloc((void**)&sg.h_counts2, N_COUNTERS * sizeof(uint32_t), hipHostMallocDefault));
    }
  }
  if (n > lib->st_capacity) {
    lib->st_capacity = 0;
    const size_t cap = n + n / 8 + 256;
    for (auto& sg : lib->stage) {
      hipFree(sg.d_s1); hipFree(sg.d_s2); hipFree(sg.d_tf1); hipFree(sg.d_tf2); hipFree(sg.d_out);
      hipFree(sg.d_gin); hipFree(sg.d_gout); hipFree(sg.d_qt1); hipFree(sg.d_qt2);
      sg.d_qt1 = sg.d_qt2 = nullptr;
      sg.d_s1 = sg.d_s2 = nullptr;
      sg.d_tf1 = sg.d_tf2 = nullptr;
      sg.d_out = nullptr;
      sg.d_gin = sg.d_gout = nullptr;
      HIP_TRY(hipMalloc(&sg.d_s1, cap * sizeof(uint32_t)));
      HIP_TRY(hipMalloc(&sg.d_s2, cap * sizeof(uint32_t)));
      HIP_TRY(hipMalloc(&sg.d_tf1, cap * 12 * sizeof(double)));
      HIP_TRY(hipMalloc(&sg.d_tf2, cap * 12 * sizeof(double)));
      HIP_TRY(hipMalloc(&sg.d_out, cap * sizeof(hfcl_result)));
    }
    lib->st_capacity = cap;
  }
  for (auto& sg : lib->stage) {
    if (gin && !sg.d_gin) HIP_TRY(hipMalloc(&sg.d_gin, lib->st_capacity * sizeof(hfcl_guess)));
    if (gout && !sg.d_gout) HIP_TRY(hipMalloc(&sg.d_gout, lib->st_capacity * sizeof(hfcl_guess)));
    if (compact && !sg.d_qt1) {
      HIP_TRY(hipMalloc(&sg.d_qt1, lib->st_capacity * 7 * sizeof(double)));
      HIP_TRY(hipMalloc(&sg.d_qt2, lib->st_capacity * 7 * sizeof(double)));
    }
  }
  return HFCL_OK;
}

// what a host batch reports after its records are back: pairs without an evaluator, requests the reference rejects
static int host_batch_checks(hfcl_lib* lib, const hfcl_collision_request* creq, const hfcl_distance_request* dreq) {
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

// A small batch from a library of many shape kinds pays a launch for every bucket the LIBRARY can reach (up to a dozen
// dependent launches, ~5 us each) although its pairs fall into one or two: the host has the ids, so it classifies them
// itself and, for the lifetime of this object, only the kernels of buckets that hold a pair are launched (a single
// box x box pair: 5 launches instead of 12).
struct SmallBatchBuckets {
  hfcl_lib* lib;
  uint32_t library_buckets;
  SmallBatchBuckets(hfcl_lib* l, const uint32_t* s1, const uint32_t* s2, size_t n, bool distance_mode) : lib(l), library_buckets(l->possible_buckets) {
    if (n > 1024 || lib->h_kinds.size() != lib->n_shapes) return;
    uint32_t mask = 0;
    for (size_t i = 0; i < n; ++i)
      mask |= 1u << ((s1[i] < lib->n_shapes && s2[i] < lib->n_shapes) ? bucket_of(lib->h_kinds[s1[i]], lib->h_kinds[s2[i]], distance_mode)
                                                                       : int(B_UNSUPPORTED));
    lib->possible_buckets = mask & (library_buckets | (1u << B_UNSUPPORTED));
    if (lib->helper) lib->helper->possible_buckets = lib->possible_buckets;
  }
  ~SmallBatchBuckets() {
    lib->possible_buckets = library_buckets;
    if (lib->helper) lib->helper->possible_buckets = library_buckets;
  }
};

// Host batch of at most hfcl_lib::SMALL_MAX pairs: the per-call cost is what counts (a hpp::fcl::collide() caller sends one
// pair).  Every input array is packed into one pinned block, which crosses the link as ONE copy; the kernels and the one
// copy back run on the same stream; the call waits for that stream.  (Five pageable copies, the event hand-overs between
// three streams and the copy back cost ~85 us of host time per call; profiles/r03_d.)
static int host_batch_small(hfcl_lib* lib, const uint32_t* s1, const uint32_t* s2, const double* tf1, const double* tf2, size_t n,
                            const hfcl_collision_request* creq, const hfcl_distance_request* dreq, hfcl_result* out,
                            const hfcl_guess* gin, hfcl_guess* gout, bool compact) {
  constexpr size_t C = hfcl_lib::SMALL_MAX;
  auto up = [](size_t x) { return (x + 255) & ~size_t(255); };
  const size_t pose_in = compact ? 7 : 12;
  // block layout for n pairs: [pose1][pose2][guess in][ids 1][ids 2] | [records][guess out] | [expanded poses 1][2] (device only)
  const size_t o_p1 = 0, o_p2 = o_p1 + n * pose_in * 8, o_gin = o_p2 + n * pose_in * 8, o_s1 = o_gin + (gin ? n * sizeof(hfcl_guess) : 0),
               o_s2 = o_s1 + up(n * 4), in_bytes = o_s2 + up(n * 4);
  const size_t o_out = up(in_bytes), o_gout = o_out + n * sizeof(hfcl_result), out_bytes = n * sizeof(hfcl_result) + (gout ? n * sizeof(hfcl_guess) : 0);
  const size_t o_tf1 = up(o_out + out_bytes), o_tf2 = o_tf1 + n * 96;
  if (!lib->d_pack) {
    const size_t cap = up(C * (2 * 96 + sizeof(hfcl_guess)) + 2 * up(C * 4)) + up(C * (sizeof(hfcl_result) + sizeof(hfcl_guess))) + 2 * C * 96 + 1024;
    HIP_TRY(hipHostMalloc((void**)&lib->h_pack, cap, hipHostMallocDefault));
    HIP_TRY(hipHostMalloc((void**)&lib->h_pack_counts, 2 * N_COUNTERS * sizeof(uint32_t), hipHostMallocDefault));
    HIP_TRY(hipMalloc(&lib->d_pack, cap));
  }
  if (!lib->s_cmp) {  // (the pipeline's streams; this path uses the compute stream only)
    const int rc0 = ensure_staging(lib, 256, false, false, false);
    if (rc0) return rc0;
  }
  char* h = lib->h_pack;
  char* d = lib->d_pack;
  memcpy(h + o_p1, tf1, n * pose_in * 8);
  memcpy(h + o_p2, tf2, n * pose_in * 8);
  if (gin) memcpy(h + o_gin, gin, n * sizeof(hfcl_guess));
  memcpy(h + o_s1, s1, n * 4);
  memcpy(h + o_s2, s2, n * 4);
  hipStream_t st = lib->s_cmp;
  HIP_TRY(hipMemcpyAsync(d, h, in_bytes, hipMemcpyHostToDevice, st));
  const double *d_tf1 = reinterpret_cast<const double*>(d + o_p1), *d_tf2 = reinterpret_cast<const double*>(d + o_p2);
  if (compact) {
    launch_expand_poses(st, d_tf1, reinterpret_cast<double*>(d + o_tf1), uint32_t(n));
    launch_expand_poses(st, d_tf2, reinterpret_cast<double*>(d + o_tf2), uint32_t(n));
    d_tf1 = reinterpret_cast<const double*>(d + o_tf1);
    d_tf2 = reinterpret_cast<const double*>(d + o_tf2);
  }
  memset(lib->acc_counts, 0, sizeof(lib->acc_counts));
  memset(lib->h_pack_counts, 0, 2 * N_COUNTERS * sizeof(uint32_t));
  lib->in_host_batch = true;
  lib->counts_dst = lib->h_pack_counts;
  if (lib->helper) lib->helper->counts_dst = lib->h_pack_counts + N_COUNTERS;  // (a batch this small is never split)
  int rc;
  {
    SmallBatchBuckets batch_buckets(lib, s1, s2, n, dreq != nullptr);
    const uint32_t* d_s1 = reinterpret_cast<const uint32_t*>(d + o_s1);
    const uint32_t* d_s2 = reinterpret_cast<const uint32_t*>(d + o_s2);
    const hfcl_guess* d_gin = gin ? reinterpret_cast<const hfcl_guess*>(d + o_gin) : nullptr;
    hfcl_guess* d_gout = gout ? reinterpret_cast<hfcl_guess*>(d + o_gout) : nullptr;
    if (creq)
      rc = hfcl_collide_batch_device(lib, d_s1, d_s2, d_tf1, d_tf2, n, creq, reinterpret_cast<hfcl_result*>(d + o_out), d_gin, d_gout, st);
    else
      rc = hfcl_distance_batch_device(lib, d_s1, d_s2, d_tf1, d_tf2, n, dreq, reinterpret_cast<hfcl_result*>(d + o_out), d_gin, d_gout, st);
  }
  hipError_t e = hipSuccess;
  if (rc == HFCL_OK) e = hipMemcpyAsync(h + o_out, d + o_out, out_bytes, hipMemcpyDeviceToHost, st);
  const hipError_t e2 = hipStreamSynchronize(st);  // (also after a failed launch sequence: nothing of this call stays in flight)
  lib->in_host_batch = false;
  lib->counts_dst = nullptr;
  if (lib->helper) lib->helper->counts_dst = nullptr;
  if (rc) return rc;
  if (e != hipSuccess || e2 != hipSuccess) {
    set_error(std::string("host batch (small): ") + hipGetErrorString(e != hipSuccess ? e : e2));
    return HFCL_ERR_HIP;
  }
  memcpy(out, h + o_out, n * sizeof(hfcl_result));
  if (gout) memcpy(gout, h + o_gout, n * sizeof(hfcl_guess));
  for (int i = 0; i < N_COUNTERS; ++i) lib->acc_counts[i] = lib->h_pack_counts[i] + (lib->last_split ? lib->h_pack_counts[N_COUNTERS + i] : 0u);
  lib->last_host = true;
  return host_batch_checks(lib, creq, dreq);
}

// Host-buffer entry point = the drop-in boundary a user of hpp::fcl::collide() / distance() gets.  The batch is cut into
// chunks that flow through a three-stage pipeline on three streams: H2D of chunk k+1 | kernels of chunk k | D2H of chunk
// k-1, over PIPE_SLOTS device buffer sets.  The caller's thread feeds the pipeline (copies in, launches); a helper thread
// drains it (waits for a chunk's kernels, adds up its bucket populations, copies the records out, releases the slot).  The
// copies go straight from / to the caller's (pageable) arrays: the runtime moves them at the link rate (tools/valu_peak.hip:
// 56 GB/s pageable vs 57 GB/s pinned, 37.6 GB/s per direction when both run), an extra staging memcpy would only add a
// 33 GB/s single-thread stage.  No device-wide synchronisation: other streams of the process are not disturbed.
// pipelined = false (contact lists: their pair ids are batch-wide) runs the batch as one chunk.
static int host_batch(hfcl_lib* lib, const uint32_t* s1, const uint32_t* s2, const double* tf1, const double* tf2,
                      size_t n, const hfcl_collision_request* creq, const hfcl_distance_request* dreq, hfcl_result* out,
                      const hfcl_guess* gin, hfcl_guess* gout, bool pipelined = true, bool compact = false, bool f32 = false) {
  // f32 (hfcl_*_batch_f32): tf1 / tf2 are 7-FLOAT poses and `out` hfcl_result_f32 records, both smaller than what the slots' buffers hold for the
  // fp64 formats, so the same staging serves; the chunks go through the fp32 device path.  No guesses in that format.
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
  if (pipelined && n <= hfcl_lib::SMALL_MAX && !lib->pipe_chunk && !f32) return host_batch_small(lib, s1, s2, tf1, tf2, n, creq, dreq, out, gin, gout, compact);
  // Chunks: large enough that a chunk's fixed costs (a dozen launches, ~0.1 ms) vanish, small enough that the pipeline
  // has several chunks to overlap.  The link is busy from the first byte to the last only if the first chunk is small
  // (nothing computes until it has arrived) and the last one too (nothing overlaps its way back): the sizes ramp up
  // geometrically from 16k pairs to the steady size and down again (1M pairs: 6.4 -> see profiles/r03_c).
  std::vector<size_t> bounds;  // chunk k = [bounds[k], bounds[k + 1])
  bounds.push_back(0);
  size_t max_chunk = n;
  if (pipelined && lib->pipe_chunk) {
    for (size_t lo = 0; lo < n; lo += lib->pipe_chunk) bounds.push_back(std::min(n, lo + lib->pipe_chunk));
    max_chunk = std::min(n, lib->pipe_chunk);
  } else if (pipelined && f32 && n > (size_t(1) << 16)) {
    // fp32: 108 B per pair cross the link -- a quarter of the time the kernels take -- and those kernels live on latency, so a chunk a quarter the
    // size takes 0.44 of the time, not 0.25: few, large chunks (1M convex32 pairs: three chunks 3.7 ms, the fp64 policy's ten 6.4 ms, one chunk 4.1 ms)
    const size_t c = std::min<size_t>(std::max<size_t>((n + 2) / 3, size_t(1) << 16), size_t(1) << 19);
    for (size_t lo = 0; lo < n; lo += c) bounds.push_back(std::min(n, lo + c));
    max_chunk = std::min(n, c);
  } else if (pipelined && n > (size_t(1) << 16)) {
    const size_t steady = std::min<size_t>(std::max<size_t>(n / 6, size_t(1) << 16), size_t(1) << 18);
    std::vector<size_t> up;    // 16k, 32k, ... below the steady size
    for (size_t c = size_t(1) << 14; c < steady; c *= 2) up.push_back(c);
    size_t ramp = 0;
    for (size_t c : up) ramp += c;
    while (!up.empty() && 2 * ramp + steady > n) {  // a batch too small for the whole ramp: shorten it from the top
      ramp -= up.back();
      up.pop_back();
    }
    size_t lo = 0;
    for (size_t c : up) bounds.push_back(lo += c);
    const size_t mid_end = n - ramp;
    while (mid_end - lo > steady + steady / 2) bounds.push_back(lo += steady);
    if (mid_end > lo) bounds.push_back(lo = mid_end);
    for (size_t k = up.size(); k-- > 0;) bounds.push_back(lo += up[k]);
    max_chunk = 0;
    for (size_t k = 0; k + 1 < bounds.size(); ++k) max_chunk = std::max(max_chunk, bounds[k + 1] - bounds[k]);
  } else {
    bounds.push_back(n);
  }
  const size_t n_chunks = bounds.size() - 1;
  int rc = ensure_staging(lib, max_chunk, gin != nullptr, gout != nullptr, compact);
  if (rc) return rc;
  SmallBatchBuckets batch_buckets(lib, s1, s2, n, dreq != nullptr);
  constexpr int S = hfcl_lib::PIPE_SLOTS;
  memset(lib->acc_counts, 0, sizeof(lib->acc_counts));
  lib->in_host_batch = true;

  // Host threads keep the streams busy.  A copy between pageable memory and the device holds its caller until the data
  // has moved (above a few MB; the first time a range of host memory is used it is also pinned, several times slower),
  // so every stream that copies has a thread of its own: two feeders (the arrays of object 1 and of object 2, on two
  // streams: one blocking copy at a time leaves the link idle between copies, 42 instead of 50 GB/s), the drainer
  // (records out, queued behind the chunk's kernels on the stream; bucket populations), and the caller's thread, which
  // launches the kernels of a chunk as soon as its inputs are on their way (profiles/r03_c).
  std::mutex mu;
  std::condition_variable cv;
  size_t copied[2] = {0, 0}, issued = 0, drained = 0;  // chunks whose inputs are queued / whose kernels are launched / whose records are back
  int side_rc = HFCL_OK;                       // first failure of the feeder or the drainer
  std::string side_err;
  bool abort_all = false;
  auto side_fail = [&](const char* who, hipError_t e) {
    std::lock_guard<std::mutex> lk(mu);
    if (side_rc == HFCL_OK) {
      side_rc = HFCL_ERR_HIP;
      side_err = std::string("host batch (") + who + "): " + hipGetErrorString(e);
    }
    abort_all = true;
    cv.notify_all();
  };

  // option pipe_trace: per-chunk time line of the three threads on stderr (ms since the call started)
  const bool trace = lib->pipe_trace;
  const auto t_call = std::chrono::steady_clock::now();
  auto ms_now = [&]() { return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t_call).count(); };
  std::vector<double> tr(trace ? 6 * n_chunks : 0);
  // `copied[h]`: chunks whose arrays of object h + 1 (ids, poses; h = 0 also the guesses) are queued
  auto feed = [&](int h) {
    hipSetDevice(lib->device);
    hipStream_t st = h ? lib->s_h2d2 : lib->s_h2d;
    for (size_t k = 0; k < n_chunks; ++k) {
      {
        std::unique_lock<std::mutex> lk(mu);
        cv.wait(lk, [&] { return drained + S > k || abort_all; });  // the slot's previous chunk must be back on the host
        if (abort_all) return;
      }
      hfcl_lib::Staging& sg = lib->stage[k % S];
      const size_t lo = bounds[k], m = bounds[k + 1] - lo;
      if (trace && !h) tr[6 * k] = ms_now();
      const uint32_t* ids = h ? s2 : s1;
      const double* tf = h ? tf2 : tf1;
      hipError_t e = hipMemcpyAsync(h ? sg.d_s2 : sg.d_s1, ids + lo, m * sizeof(uint32_t), hipMemcpyHostToDevice, st);
      if (f32) {
        if (e == hipSuccess) e = hipMemcpyAsync(h ? sg.d_tf2 : sg.d_tf1, reinterpret_cast<const float*>(tf) + 7 * lo, m * 7 * sizeof(float), hipMemcpyHostToDevice, st);
      } else if (compact) {  // tf1 / tf2 are 7-double poses: expanded to Transform3f images by the first kernel of the chunk
        if (e == hipSuccess) e = hipMemcpyAsync(h ? sg.d_qt2 : sg.d_qt1, tf + 7 * lo, m * 7 * sizeof(double), hipMemcpyHostToDevice, st);
      } else {
        if (e == hipSuccess) e = hipMemcpyAsync(h ? sg.d_tf2 : sg.d_tf1, tf + 12 * lo, m * 12 * sizeof(double), hipMemcpyHostToDevice, st);
      }
      if (e == hipSuccess && gin && !h) e = hipMemcpyAsync(sg.d_gin, gin + lo, m * sizeof(hfcl_guess), hipMemcpyHostToDevice, st);
      if (e == hipSuccess) e = hipEventRecord(h ? sg.ev_in2 : sg.ev_in, st);
      if (e != hipSuccess) {
        side_fail("feed", e);
        return;
      }
      if (trace && !h) tr[6 * k + 1] = ms_now();
      std::lock_guard<std::mutex> lk(mu);
      copied[h] = k + 1;
      cv.notify_all();
    }
  };
  auto drain = [&]() {
    hipSetDevice(lib->device);
    for (size_t k = 0; k < n_chunks; ++k) {
      {
        std::unique_lock<std::mutex> lk(mu);
        cv.wait(lk, [&] { return issued > k || abort_all; });
        if (abort_all && issued <= k) return;
      }
      hfcl_lib::Staging& sg = lib->stage[k % S];
      const size_t lo = bounds[k], m = bounds[k + 1] - lo;
      // the records' way back is queued behind the chunk's kernels on the third stream (no host round trip between the two)
      hipError_t e = trace ? hipEventSynchronize(sg.ev_done) : hipSuccess;
      if (trace) tr[6 * k + 4] = ms_now();
      if (e == hipSuccess) e = hipStreamWaitEvent(lib->s_d2h, sg.ev_done, 0);
      if (e == hipSuccess)
        e = f32 ? hipMemcpyAsync(reinterpret_cast<hfcl_result_f32*>(out) + lo, sg.d_out, m * sizeof(hfcl_result_f32), hipMemcpyDeviceToHost, lib->s_d2h)
                : hipMemcpyAsync(out + lo, sg.d_out, m * sizeof(hfcl_result), hipMemcpyDeviceToHost, lib->s_d2h);
      if (e == hipSuccess && gout) e = hipMemcpyAsync(gout + lo, sg.d_gout, m * sizeof(hfcl_guess), hipMemcpyDeviceToHost, lib->s_d2h);
      if (e == hipSuccess) e = hipStreamSynchronize(lib->s_d2h);  // (ev_done has passed: the counters are on the host too)
      if (e == hipSuccess)
        for (int i = 0; i < N_COUNTERS; ++i) lib->acc_counts[i] += sg.h_counts[i] + (sg.split ? sg.h_counts2[i] : 0u);
      if (e != hipSuccess) {
        side_fail("drain", e);
        return;
      }
      if (trace) tr[6 * k + 5] = ms_now();
      std::lock_guard<std::mutex> lk(mu);
      drained = k + 1;
      cv.notify_all();
    }
  };
  const bool threaded = n_chunks > 1;
  std::thread feeder, feeder2, drainer;
  if (threaded) {
    feeder = std::thread(feed, 0);
    feeder2 = std::thread(feed, 1);
    drainer = std::thread(drain);
  }

  auto fail = [&](int code) {  // stop the pipeline and leave
    {
      std::lock_guard<std::mutex> lk(mu);
      abort_all = true;
    }
    cv.notify_all();
    if (threaded) {
      feeder.join();
      feeder2.join();
      drainer.join();
    }
    hipStreamSynchronize(lib->s_h2d);
    hipStreamSynchronize(lib->s_h2d2);
    hipStreamSynchronize(lib->s_cmp);
    hipStreamSynchronize(lib->s_d2h);
    lib->in_host_batch = false;
    lib->counts_dst = nullptr;
    if (lib->helper) lib->helper->counts_dst = nullptr;
    return code;
  };
#define PIPE_TRY(expr)                                                       \
  do {                                                                       \
    hipError_t _e = (expr);                                                  \
    if (_e != hipSuccess) {                                                  \
      set_error(std::string(#expr) + ": " + hipGetErrorString(_e));          \
      return fail(HFCL_ERR_HIP);                                             \
    }                                                                        \
  } while (0)

  if (!threaded) {
    feed(0);
    feed(1);
  }
  for (size_t k = 0; k < n_chunks; ++k) {
    hfcl_lib::Staging& sg = lib->stage[k % S];
    const size_t m = bounds[k + 1] - bounds[k];
    {
      std::unique_lock<std::mutex> lk(mu);
      cv.wait(lk, [&] { return (copied[0] > k && copied[1] > k) || abort_all; });
      if (side_rc) {
        lk.unlock();
        set_error(side_err);
        return fail(side_rc);
      }
    }
    if (trace) tr[6 * k + 2] = ms_now();
    PIPE_TRY(hipStreamWaitEvent(lib->s_cmp, sg.ev_in, 0));
    PIPE_TRY(hipStreamWaitEvent(lib->s_cmp, sg.ev_in2, 0));
    if (compact) {
      launch_expand_poses(lib->s_cmp, sg.d_qt1, sg.d_tf1, uint32_t(m));
      launch_expand_poses(lib->s_cmp, sg.d_qt2, sg.d_tf2, uint32_t(m));
    }
    memset(sg.h_counts, 0, N_COUNTERS * sizeof(uint32_t));  // (a skipped batch -- -inf margin -- copies no counters)
    memset(sg.h_counts2, 0, N_COUNTERS * sizeof(uint32_t));
    lib->counts_dst = sg.h_counts;
    if (batch_splits(lib, m)) {
      rc = ensure_helper(lib);
      if (rc) return fail(rc);
      lib->helper->counts_dst = sg.h_counts2;
    }
    if (f32 && creq)
      rc = hfcl_collide_batch_device_f32(lib, sg.d_s1, sg.d_s2, reinterpret_cast<const float*>(sg.d_tf1), reinterpret_cast<const float*>(sg.d_tf2), m, creq,
                                         reinterpret_cast<hfcl_result_f32*>(sg.d_out), lib->s_cmp);
    else if (f32)
      rc = hfcl_distance_batch_device_f32(lib, sg.d_s1, sg.d_s2, reinterpret_cast<const float*>(sg.d_tf1), reinterpret_cast<const float*>(sg.d_tf2), m, dreq,
                                          reinterpret_cast<hfcl_result_f32*>(sg.d_out), lib->s_cmp);
    else if (creq)
      rc = hfcl_collide_batch_device(lib, sg.d_s1, sg.d_s2, sg.d_tf1, sg.d_tf2, m, creq, sg.d_out, gin ? sg.d_gin : nullptr,
                                     gout ? sg.d_gout : nullptr, lib->s_cmp);
    else
      rc = hfcl_distance_batch_device(lib, sg.d_s1, sg.d_s2, sg.d_tf1, sg.d_tf2, m, dreq, sg.d_out, gin ? sg.d_gin : nullptr,
                                      gout ? sg.d_gout : nullptr, lib->s_cmp);
    if (rc) return fail(rc);
    sg.split = lib->last_split;
    PIPE_TRY(hipEventRecord(sg.ev_done, lib->s_cmp));
    if (trace) tr[6 * k + 3] = ms_now();
    {
      std::lock_guard<std::mutex> lk(mu);
      issued = k + 1;
    }
    cv.notify_all();
    if (!threaded) drain();
  }
#undef PIPE_TRY
  if (threaded) {
    feeder.join();
    feeder2.join();
    drainer.join();
  }
  if (trace) {
    fprintf(stderr, "[hfcl pipe] %zu pairs, %zu chunks, %.3f ms\n", n, n_chunks, ms_now());
    for (size_t k = 0; k < n_chunks; ++k)
      fprintf(stderr, "[hfcl pipe] chunk %2zu %7zu pairs: copy-in issued %.3f..%.3f  launches %.3f..%.3f  kernels done %.3f  records out %.3f\n", k,
              bounds[k + 1] - bounds[k], tr[6 * k], tr[6 * k + 1], tr[6 * k + 2], tr[6 * k + 3], tr[6 * k + 4], tr[6 * k + 5]);
  }
  lib->in_host_batch = false;
  lib->counts_dst = nullptr;
  if (lib->helper) lib->helper->counts_dst = nullptr;
  lib->last_host = true;
  if (side_rc) {
    set_error(side_err);
    return side_rc;
  }
  return host_batch_checks(lib, creq, dreq);
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

int hfcl_collide_batch_f32(hfcl_lib* lib, const uint32_t* shape1, const uint32_t* shape2, const float* pose1, const float* pose2, size_t n,
                           const hfcl_collision_request* req, hfcl_result_f32* out) {
  if (!req) {
    set_error("null request");
    return HFCL_ERR_INVALID_ARGUMENT;
  }
  return host_batch(lib, shape1, shape2, reinterpret_cast<const double*>(pose1), reinterpret_cast<const double*>(pose2), n, req, nullptr,
                    reinterpret_cast<hfcl_result*>(out), nullptr, nullptr, true, false, true);
}
int hfcl_distance_batch_f32(hfcl_lib* lib, const uint32_t* shape1, const uint32_t* shape2, const float* pose1, const float* pose2, size_t n,
                            const hfcl_distance_request* req, hfcl_result_f32* out) {
  if (!req) {
    set_error("null request");
    return HFCL_ERR_INVALID_ARGUMENT;
  }
  return host_batch(lib, shape1, shape2, reinterpret_cast<const double*>(pose1), reinterpret_cast<const double*>(pose2), n, nullptr, req,
                    reinterpret_cast<hfcl_result*>(out), nullptr, nullptr, true, false, true);
}

int hfcl_collide_batch_qt(hfcl_lib* lib, const uint32_t* shape1, const uint32_t* shape2, const double* pose1,
                          const double* pose2, size_t n, const hfcl_collision_request* req, hfcl_result* out,
                          const hfcl_guess* guess_in, hfcl_guess* guess_out) {
  if (!req) {
    set_error("null request");
    return HFCL_ERR_INVALID_ARGUMENT;
  }
  return host_batch(lib, shape1, shape2, pose1, pose2, n, req, nullptr, out, guess_in, guess_out, true, true);
}
int hfcl_distance_batch_qt(hfcl_lib* lib, const uint32_t* shape1, const uint32_t* shape2, const double* pose1,
                           const double* pose2, size_t n, const hfcl_distance_request* req, hfcl_result* out,
                           const hfcl_guess* guess_in, hfcl_guess* guess_out) {
  if (!req) {
    set_error("null request");
    return HFCL_ERR_INVALID_ARGUMENT;
  }
  return host_batch(lib, shape1, shape2, pose1, pose2, n, nullptr, req, out, guess_in, guess_out, true, true);
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
  int rc = host_batch(lib, shape1, shape2, tf1, tf2, n, req, nullptr, out, nullptr, nullptr, /*pipelined=*/false);
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
// (diagnostic, not part of the ABI: the counters of the last batch's walk rounds -- WalkArgs::ctr, 8 words per round -- of the mesh x mesh (solid = 0) or the
// mesh x solid walks, and BVH_CTR_TASKS / BVH_CTR_SUSPENDED of their split tables behind them: tools/dbg/walk_counters.py)
extern "C" int hfcl_debug_walk_counters(hfcl_lib* lib, int solid, uint32_t* out34) {
  if (!lib || !out34) return HFCL_ERR_INVALID_ARGUMENT;
  if (hipSetDevice(lib->device) != hipSuccess || hipDeviceSynchronize() != hipSuccess) return HFCL_ERR_HIP;
  memset(out34, 0, 34 * sizeof(uint32_t));
  const uint32_t* src = solid ? lib->d_swalk_ctr : lib->d_walk_ctr;
  if (src && hipMemcpy(out34, src, 8 * WALK_ROUNDS * sizeof(uint32_t), hipMemcpyDeviceToHost) != hipSuccess) return HFCL_ERR_HIP;
  uint32_t ctr[BVH_CTR_WORDS] = {0};
  // (a mixed batch run beside walks its mesh x mesh pairs on the second set of tables; otherwise the kind walked last owns the first)
  const uint32_t* bsrc = (!solid && lib->d_bvh2_ctr) ? lib->d_bvh2_ctr : lib->d_bvh_ctr;
  if (bsrc && hipMemcpy(ctr, bsrc, sizeof(ctr), hipMemcpyDeviceToHost) != hipSuccess) return HFCL_ERR_HIP;
  out34[32] = ctr[BVH_CTR_TASKS];
  out34[33] = ctr[BVH_CTR_SUSPENDED];
  return HFCL_OK;
}
// pairs per chunk of the host-buffer pipeline (0 = automatic: n/8 clamped to 32k .. 256k)
void hfcl_lib_set_host_chunk(hfcl_lib* lib, size_t pairs) {
  if (lib) lib->pipe_chunk = pairs;
}
// Options by name (the list: option_keys above; INTEGRATION.md describes them).  Keys are case-insensitive, an "HFCL_" prefix -- the
// spelling of the environment fallback -- is accepted.  Holds from the next batch on; the caller does not call it while a batch of this
// library is being set up on another thread (the library has no lock of its own, like every other hfcl_lib_set_*).
int hfcl_lib_set_option(hfcl_lib* lib, const char* key, const char* value) {
  if (!lib || !key || !value) {
    set_error("hfcl_lib_set_option: null argument");
    return HFCL_ERR_INVALID_ARGUMENT;
  }
  std::string k;
  for (const char* c = key; *c; ++c) k.push_back(char(tolower(static_cast<unsigned char>(*c))));
  if (k.compare(0, 5, "hfcl_") == 0) k.erase(0, 5);
  const int rc = apply_option(lib, k, value);
  if (rc != HFCL_OK) {
    set_error("hfcl_lib_set_option: unknown option or value out of range: " + std::string(key) + "=" + value);
    return rc;
  }
  // buffers sized by an option are sized again by the next batch
  if (k == "epa_resume_slots") lib->epa_capacity = 0;
  if (k == "bvh_task_slots") lib->bvh_split_n = 0;
  if (lib->helper) apply_option(lib->helper, k, value);
  return HFCL_OK;
}
// The option names, one per call: index 0, 1, ... until nullptr.
const char* hfcl_lib_option_key(int index) {
  if (index < 0) return nullptr;
  const char* const* k = option_keys();
  for (int i = 0; k[i]; ++i)
    if (i == index) return k[i];
  return nullptr;
}
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

// Walks of the last distance() batch that the pooled continuations walked again in the reference's order (BvhSpill::rerun_count): out4 = mesh x mesh
// walks continued by waves, of those re-run in order; the same two for mesh x solid.  Waits for the device like hfcl_last_bucket_counts.
void hfcl_last_ordered_reruns(hfcl_lib* lib, uint32_t* out4) {
  static const int idx[4] = {CTR_DIST_SUSP, CTR_DIST_RERUN, CTR_SHAPE_DIST_SUSP, CTR_SHAPE_DIST_RERUN};
  if (lib) {
    hipSetDevice(lib->device);
    hipDeviceSynchronize();
  }
  for (int k = 0; k < 4; ++k) {
    uint32_t c = 0;
    if (lib) {
      if (lib->last_host)
        c = lib->acc_counts[idx[k]];
      else {
        c = lib->h_counts ? lib->h_counts[idx[k]] : 0u;
        if (lib->last_split && lib->helper && lib->helper->h_counts) c += lib->helper->h_counts[idx[k]];
      }
    }
    out4[k] = c;
  }
}

}  // extern "C"
