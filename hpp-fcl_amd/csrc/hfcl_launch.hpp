// hfcl_launch.hpp -- host-callable launchers of the HIP kernels.  One translation unit per kernel family, so that a
// change to one kernel recompiles in parallel with nothing else and A/B builds of a single family are cheap:
//
//   hfcl_k_gjk.hip   k_classify       pair -> bucket lists (block-aggregated atomics), one pass over the shape ids
//                    k_closed<T>      closed forms (sphere / capsule / cylinder / box-sphere pairs, every Plane /
//                                     Halfspace row), one pair per lane (fp64: poses / records staged through LDS)
//                    k_gjk_prim<T>    GJK for Box/Capsule/Cone/Cylinder/Ellipsoid/Sphere pairs, one pair per lane
//                    k_gjk_cvx<W,M>   GJK with hulls of <= 32 vertices: one pair per W-lane group, hull vertices in the
//                                     group's registers, support = per-lane dots + DPP-butterfly arg-max
//                    k_gjk_large<T>   GJK when a hull has more than 32 vertices (scan / hill-climb from memory)
//                    k_unsupported<T>, k_fill_skipped
//   hfcl_k_epa.hip   k_epa<T,WE,CAP,TIER>, k_epa_stream<T,WE,CAP>   EPA on the pairs GJK left in `Collision`
//                    k_epa_prepare / k_epa_loop / k_epa_records   the fp32 convex x convex fast tier in three stages
//   hfcl_k_bvh.hip   k_bvh_collide<T>   BVHModel<OBBRSS> x BVHModel<OBBRSS> collide(), one query per lane for a step budget;
//                                     k_bvh_coop<T>: the queries past it, a wave each, 64 stack entries per trip
//                                     (BvhSplit::cut_ticks: a walk a wave has had for that long is cut into chunk tasks for the kernel's next
//                                     launch, k_bvh_combine folds them back; on for mesh x solid, k_bvh_shape_coop)
//                                     (k_bvh_combine<T>: the task-level alternative)
//   hfcl_k_bvhd.hip  k_bvh_distance<T>  ... distance(), one query per lane for a step budget (this unit is built without
//                                     contraction: triangle ids equal to the reference's); k_bvh_distance_pool<T, PQ>: the walks past
//                                     it, PQ per wave, their box and triangle tests pooled, DFS order kept by a marker
//                                     (k_bvh_distance_coop<T>: the ordered wave-per-walk form)
//                    k_shape_obb<T>, k_bvh_collide<T, ., ., SOLID>, k_bvh_shape_coop<T>, k_bvh_shape_finish<T>
//                                     BVHModel<OBBRSS> x convex solid, first-contact collide(): the solids' OBBs, the walk
//                                     (lane, then wave), the leaves that need EPA
//                    k_shape_obbrss<T>, k_bvh_shape_distance_lane<T>, k_bvh_shape_distance_pool<T>   ... distance()
//                                     (k_bvh_shape_distance_coop<T>: the ordered form, libraries with a Plane / Halfspace)
//                    k_bvh_shape<T> / k_bvh_shape_distance<T>   the same rows, one 16-lane group per query: requests that
//                                     keep walking after a contact, models deeper than the lanes' stacks
//                    k_triangle<T>    top-level TriangleP pairs
//
// Every launcher is asynchronous on `st` and does no error checking of its own (run_batch_one checks hipGetLastError once).
#pragma once
#include "hfcl_dev.hpp"

void launch_classify(int grid, hipStream_t st, const Work& wk, const uint8_t* kinds, uint32_t n_shapes, bool distance_mode);
template <typename T> void launch_unsupported(int grid, hipStream_t st, const Work& wk, const IO<T>& io, int bucket);
template <typename T> void launch_closed(int grid, hipStream_t st, const Work& wk, const LibView<T>& lv, const IO<T>& io, const QParams<T>& q, bool staged);
template <typename T> void launch_gjk_prim(int grid, hipStream_t st, const Work& wk, const LibView<T>& lv, const IO<T>& io, const QParams<T>& q, bool bvg);
// m: 0 = convex-convex, 1 = prim-convex, 2 = convex-prim; w: lanes per pair (2 / 4 / 8 / 16 / 32 / 64)
template <typename T> void launch_gjk_cvx(int m, int w, bool bvg, int grid, hipStream_t st, const Work& wk, const LibView<T>& lv, const IO<T>& io, const QParams<T>& q);
template <typename T> void launch_gjk_large(int grid, hipStream_t st, const Work& wk, const LibView<T>& lv, const IO<T>& io, const QParams<T>& q, bool bvg);
// 7-double (quaternion w,x,y,z + translation) poses -> 12-double Transform3f images
void launch_expand_poses(hipStream_t st, const double* qt, double* tf, uint32_t n);
void launch_fill_skipped(hipStream_t st, hfcl_result* out, uint32_t n);
void launch_fill_skipped(hipStream_t st, hfcl_result_f32* out, uint32_t n);
// hfcl_k_util.hip: full records -> compact records (hfcl_result_compact{,_f32})
void launch_compact_records(hipStream_t st, const hfcl_result* in, hfcl_result_compact* out, uint32_t n);
void launch_compact_records(hipStream_t st, const hfcl_result_f32* in, hfcl_result_compact_f32* out, uint32_t n);

// tier 1 (fp32: streaming form) and tier 2 of EPA; the grids are in blocks of one wavefront
// cc_queue / general_queue (fp32): which of the two streaming forms have anything to do (convex x convex pairs have a
// queue and a kernel of their own); curved_class (fp64): the library has a shape whose support is not a vertex, i.e. the
// curved class of pairs -- a queue and a fast-tier kernel of its own -- can occur
// st2 (fp64, both classes present): the polytope-class kernel goes there, beside the curved-class kernel on st (the caller forks and joins)
template <typename T> void launch_epa_fast(int grid, hipStream_t st, const Work& wk, const LibView<T>& lv, const IO<T>& io, const QParams<T>& q, bool cc_queue, bool general_queue, int n_cus, bool curved_class = true, hipStream_t st2 = nullptr);
// the fp32 convex x convex fast tier in three stages (Work::epa_ready set): one lane per polytope before and after the loop kernel
void launch_epa_prepare(int grid, hipStream_t st, const Work& wk, const LibView<float>& lv, const IO<float>& io, const QParams<float>& q);
void launch_epa_loop(int grid, hipStream_t st, const Work& wk, const LibView<float>& lv, const QParams<float>& q, int n_cus);
void launch_epa_resume_cc(int grid, hipStream_t st, const Work& wk, const LibView<float>& lv, const IO<float>& io, const QParams<float>& q);
void launch_epa_records(int grid, hipStream_t st, const Work& wk, const LibView<float>& lv, const IO<float>& io, const QParams<float>& q);
// ... and for pairs of any convex kinds, both precisions (Work::epa_ready_g set)
template <typename T> void launch_epa_prepare_general(int grid, hipStream_t st, const Work& wk, const LibView<T>& lv, const IO<T>& io, const QParams<T>& q, bool skip_top);
template <typename T> void launch_epa_loop_general(int grid, hipStream_t st, hipStream_t st2, const Work& wk, const LibView<T>& lv, const QParams<T>& q, int n_cus, bool curved_class);
template <typename T> void launch_epa_records_general(int grid, hipStream_t st, const Work& wk, const LibView<T>& lv, const IO<T>& io, const QParams<T>& q, bool skip_top);
template <typename T> void launch_epa_requeue(hipStream_t st, const Work& wk);
template <typename T> void launch_epa_full(int grid, hipStream_t st, const Work& wk, const LibView<T>& lv, const IO<T>& io, const QParams<T>& q);

// A helper stream with its fork / join events: k_bvh_shape_finish for the items of whole walks runs there, beside the launches that walk the
// chunks of the cut ones; k_bvh_coop for the queries round 0 of a mesh x mesh walk handed over, beside the later rounds
struct AsideStream {
  hipStream_t stream;
  hipEvent_t fork, join;
};
// split: the task tables of a split traversal (tasks == nullptr: single pass)
// aside: nullptr, or WALK_ROUNDS - 1 helper streams (the continuation of what round r of the walk hands over runs on aside[r])
template <typename T> void launch_bvh_collide(int grid, hipStream_t st, const Work& wk, const LibView<T>& lv, const BvhView<T>& bv, const IO<T>& io, const QParams<T>& q, const BvhParams& bp, T break_distance2, BvhSplit split, BvhSpill spill, const AsideStream* aside = nullptr);
template <typename T> void launch_bvh_distance(int grid, hipStream_t st, const Work& wk, const LibView<T>& lv, const BvhView<T>& bv, const IO<T>& io, const QParams<T>& q, BvhSpill spill);
template <typename T> void launch_bvh_shape(int grid, hipStream_t st, const Work& wk, const LibView<T>& lv, const BvhView<T>& bv, const IO<T>& io, const QParams<T>& q, const BvhParams& bp, T break_distance2);
template <typename T> void launch_bvh_shape_distance_fast(int grid, int grid_finish, hipStream_t st, const Work& wk, const LibView<T>& lv, const BvhView<T>& bv, const IO<T>& io, const QParams<T>& q, BvhSpill spill);
template <typename T> void launch_bvh_shape_fast(int grid, int grid_finish, int coop_grid, hipStream_t st, const Work& wk, const LibView<T>& lv, const BvhView<T>& bv, const IO<T>& io, const QParams<T>& q, const BvhParams& bp, T break_distance2, BvhSplit split, BvhSpill spill, const AsideStream* aside = nullptr);
template <typename T> void launch_bvh_shape_distance(int grid, hipStream_t st, const Work& wk, const LibView<T>& lv, const BvhView<T>& bv, const IO<T>& io, const QParams<T>& q);
template <typename T> void launch_triangle(int grid, hipStream_t st, const Work& wk, const LibView<T>& lv, const IO<T>& io, const QParams<T>& q);
