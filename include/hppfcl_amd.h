/*
 * hppfcl_amd.h -- C ABI of the MI355X batched narrow-phase engine.
 *
 * This is the drop-in boundary for hpp-fcl's narrow phase.  hpp-fcl has no FFI
 * layer of its own; its seam is the C++ pair of free functions
 *     hpp::fcl::collide (o1, tf1, o2, tf2, CollisionRequest, CollisionResult&)
 *         -- /root/reference/include/hpp/fcl/collision.h:58-70, src/collision.cpp:69-130
 *     hpp::fcl::distance(o1, tf1, o2, tf2, DistanceRequest,  DistanceResult&)
 *         -- include/hpp/fcl/distance.h:53-65,  src/distance.cpp:60-109
 * and below them the uniform function-pointer signature of the dispatch tables
 * (include/hpp/fcl/collision_func_matrix.h:61-67).  The entry points declared here
 * are the *batched* form of exactly those two calls: N independent
 * (geometry, pose, geometry, pose) queries under one request, N result records.
 * The header-only shim include/hppfcl_amd_compat.hpp re-exposes them under the
 * reference's own names (hpp::fcl::collide / distance / CollisionRequest / ...).
 *
 * Plain C: pointers and sizes only.  No torch / Eigen / STL types.
 * All floating point at this boundary is fp64 unless a name ends in _f32.
 */
#ifndef HPPFCL_AMD_H
#define HPPFCL_AMD_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define HFCL_ABI_VERSION 5  /* 4: the hfcl_multi_* entry points (several devices in one process); 5: hfcl_lib_set_option / hfcl_multi_set_option, hfcl_multi_last_gather */

/* ---- geometry kinds: numeric values are hpp-fcl's NODE_TYPE
 *      (include/hpp/fcl/collision_object.h:65-89) so a caller can pass
 *      CollisionGeometry::getNodeType() straight through. ------------------------ */
enum {
  HFCL_BV_OBBRSS     = 5,  /* BVHModel<OBBRSS>                                   */
  HFCL_GEOM_BOX      = 9,  /* params = halfSide[3]      (geometric_shapes.h:164) */
  HFCL_GEOM_SPHERE   = 10, /* params[0] = radius        (geometric_shapes.h:238) */
  HFCL_GEOM_CAPSULE  = 11, /* params[0] = radius, [1] = halfLength (:381-400)    */
  HFCL_GEOM_CONE     = 12, /* params[0] = radius, [1] = halfLength (geometric_shapes.h:437-500) */
  HFCL_GEOM_CYLINDER = 13, /* params[0] = radius, [1] = halfLength (:505-570)           */
  HFCL_GEOM_CONVEX   = 14, /* num_points vertices at vertex_offset (:638-872)    */
  HFCL_GEOM_PLANE    = 15, /* params[0..2] = unit normal n, params[3] = d: n.x = d (:968-1049) */
  HFCL_GEOM_HALFSPACE= 16, /* params[0..2] = unit normal n, params[3] = d: n.x <= d (:873-962) */
  HFCL_GEOM_TRIANGLE = 17, /* 3 vertices at vertex_offset (TriangleP, :109)      */
  HFCL_GEOM_ELLIPSOID= 19  /* params = radii[3]         (geometric_shapes.h:303) */
};

/* One entry of the shape library.  Mirrors the data members of ShapeBase and its
 * subclasses that the narrow phase reads (include/hpp/fcl/shape/geometric_shapes.h). */
typedef struct hfcl_shape {
  int32_t  type;                /* one of HFCL_GEOM_* / HFCL_BV_OBBRSS                  */
  uint32_t num_points;          /* CONVEX: #vertices; TRIANGLE: 3; BVH: #vertices        */
  uint32_t vertex_offset;       /* first vertex in the library vertex array (units: vertices) */
  uint32_t bvh_index;           /* BVH only: index into the mesh table (hfcl_lib_add_bvh) */
  double   params[4];
  double   swept_sphere_radius; /* ShapeBase::getSweptSphereRadius(), geometric_shapes.h:59-102 */
} hfcl_shape;

/* Pose = memory image of hpp::fcl::Transform3f (include/hpp/fcl/math/transform.h:56-61):
 * Matrix3f R stored column-major (Eigen default) followed by Vec3f T -> 12 doubles.
 *   pose[0..8]  = R(0,0) R(1,0) R(2,0) R(0,1) R(1,1) R(2,1) R(0,2) R(1,2) R(2,2)
 *   pose[9..11] = T
 * so `reinterpret_cast<const double*>(&tf)` of a std::vector<Transform3f> is valid input. */
#define HFCL_POSE_DOUBLES 12

/* Compact fp32 pose for the fp32 device-resident path: unit quaternion (w,x,y,z) +
 * translation = 7 floats = 28 B (SURVEY.md 8d byte accounting). */
#define HFCL_POSE_F32_FLOATS 7

/* ---- enums of include/hpp/fcl/data_types.h:85-98 (same numeric values) ---------- */
enum { HFCL_GUESS_DEFAULT = 0, HFCL_GUESS_CACHED = 1, HFCL_GUESS_BOUNDING_VOLUME = 2 };
enum { HFCL_GJK_DEFAULT = 0, HFCL_GJK_POLYAK = 1, HFCL_GJK_NESTEROV = 2 };
enum { HFCL_CRIT_DEFAULT = 0, HFCL_CRIT_DUALITY_GAP = 1, HFCL_CRIT_HYBRID = 2 };
enum { HFCL_CRIT_RELATIVE = 0, HFCL_CRIT_ABSOLUTE = 1 };

/* GJK::Status, include/hpp/fcl/narrowphase/gjk.h:95-102 */
enum {
  HFCL_GJK_DID_NOT_RUN = 0, HFCL_GJK_FAILED = 1, HFCL_GJK_NO_COLLISION_EARLY_STOPPED = 2,
  HFCL_GJK_NO_COLLISION = 3, HFCL_GJK_COLLISION_WITH_PENETRATION_INFORMATION = 4,
  HFCL_GJK_COLLISION = 5
};
/* EPA::Status, gjk.h:330-341 (DidNotRun is -1 there; stored here as 15 in 4 bits) */
enum {
  HFCL_EPA_FAILED = 0, HFCL_EPA_VALID = 1, HFCL_EPA_ACCURACY_REACHED = 3,
  HFCL_EPA_DEGENERATED = 2, HFCL_EPA_NON_CONVEX = 4, HFCL_EPA_INVALID_HULL = 6,
  HFCL_EPA_OUT_OF_FACES = 8, HFCL_EPA_OUT_OF_VERTICES = 10, HFCL_EPA_FALLBACK = 12,
  HFCL_EPA_DID_NOT_RUN = 15
};

/* QueryRequest, include/hpp/fcl/collision_data.h:171-273 (same names, same defaults;
 * defaults are filled by hfcl_*_request_init). */
typedef struct hfcl_query_request {
  int32_t  gjk_initial_guess;               /* DefaultGuess                           */
  int32_t  gjk_variant;                     /* DefaultGJK                             */
  int32_t  gjk_convergence_criterion;       /* Default (VDB)                          */
  int32_t  gjk_convergence_criterion_type;  /* Relative                               */
  uint32_t gjk_max_iterations;              /* 128  narrowphase_defaults.h:47         */
  uint32_t epa_max_iterations;              /* 64   narrowphase_defaults.h:60.  DEVICE LIMIT: <= 64 -- the polytope blocks in LDS hold the
                                             * reference's default capacity (68 vertices, 132 faces; 8-bit indices); a larger value is refused with
                                             * HFCL_ERR_LIMIT, never clamped silently (the reference takes any value, collision_data.h:171-237) */
  double   gjk_tolerance;                   /* 1e-6                                   */
  double   epa_tolerance;                   /* 1e-6                                   */
  double   collision_distance_threshold;    /* Eigen dummy_precision<double> = 1e-12, collision_data.h:236 */
  double   cached_gjk_guess[3];             /* (1,0,0)                                */
  int32_t  cached_support_func_guess[2];    /* (0,0)                                  */
} hfcl_query_request;

/* CollisionRequest, collision_data.h:312-383 */
typedef struct hfcl_collision_request {
  hfcl_query_request q;
  uint32_t num_max_contacts;    /* 1 ; 0 is an error (src/collision.cpp:82-85)          */
  int32_t  enable_contact;      /* true                                                 */
  double   security_margin;     /* 0 ; -inf => no test, cleared result (collision.cpp:73) */
  double   break_distance;      /* 1e-3                                                 */
  double   distance_upper_bound;/* DBL_MAX                                              */
} hfcl_collision_request;

/* DistanceRequest, collision_data.h:987-1050 */
typedef struct hfcl_distance_request {
  hfcl_query_request q;
  int32_t  enable_nearest_points;  /* deprecated in the reference; always computed      */
  int32_t  enable_signed_distance; /* true => EPA on penetration                        */
  double   rel_err;                /* 0 (BVH distance pruning)                          */
  double   abs_err;                /* 0                                                 */
} hfcl_distance_request;

/* Packed per-pair status word (both result formats):
 *   bits  0..2  gjk status        bits  3..6  epa status (HFCL_EPA_*, 15 = did not run)
 *   bit   7     contact flag (collide: a Contact was added; distance: min_distance <= 0)
 *   bits  8..15 gjk iterations    bits 16..22 epa iterations
 *   bit  23     operands were swapped internally and swapped back
 *   bit  31     record not computed (unsupported pair / -inf margin)                    */
#define HFCL_STATUS_GJK(s)       ((s) & 7u)
#define HFCL_STATUS_EPA(s)       (((s) >> 3) & 15u)
#define HFCL_STATUS_CONTACT(s)   (((s) >> 7) & 1u)
#define HFCL_STATUS_GJK_ITERS(s) (((s) >> 8) & 255u)
#define HFCL_STATUS_EPA_ITERS(s) (((s) >> 16) & 127u)
#define HFCL_STATUS_SKIPPED(s)   (((s) >> 31) & 1u)

/* One fp64 result record = the fields of DistanceResult (collision_data.h:1053-1174)
 * or, for collide, of CollisionResult + its (at most one, for shape-shape) Contact
 * (collision_data.h:59-166, 391-494):
 *   distance():  min_distance = distance, normal, nearest_points = p1,p2, b1,b2
 *   collide():   Contact.penetration_depth = distance (NOT margin-adjusted,
 *                shape_shape_func.h:155), Contact.normal/nearest_points = normal,p1,p2,
 *                Contact.pos = (p1+p2)/2, CollisionResult.distance_lower_bound =
 *                distance - security_margin, numContacts() = num_contacts.
 * World frame; normal points from o1 to o2.  NaN where the reference yields NaN. */
typedef struct hfcl_result {
  double   distance;
  double   normal[3];
  double   p1[3];
  double   p2[3];
  int32_t  b1, b2;          /* Contact::NONE = -1 for primitives; triangle ids for BVH */
  uint32_t status;          /* packed, see above                                       */
  int32_t  num_contacts;    /* collide only                                            */
} hfcl_result;              /* 96 bytes                                                */
/* Mesh distance(): which triangle pair is reported.  Triangles that share the closest vertex or edge are at the minimal distance bit
 * for bit, and DistanceResult::update (collision_data.h:1099-1125) keeps the first one the reference's walk (distanceRecurse,
 * traversal_recurse.cpp:153-203) meets.  b1 / b2 are that pair, in every record and in every run: the default continuation kernels of long
 * walks (several walks per wave, their tests pooled, evaluated out of the reference's order) reproduce the choice through a marker rule and
 * verify it -- a walk whose reported pair stands under a bound that exceeds its own distance (a rounding error of the two computations,
 * where the choice could depend on the minimum the sequential walk held at that very entry) is walked again inside the same kernel in the
 * reference's order, ~0.6 % of cfg4d's walks at ~0.5 % of its time (hfcl_last_ordered_reruns counts them).  Measured: every byte of every
 * record equal to the lanes' sequential walk in 1.5 M mesh x mesh and mesh x solid queries (tools/distance_order_soak.py). */

/* Compact fp32 record for the fp32 device-resident path: 44 bytes.  It has NO b1 / b2: a BVHModel pair run through the
 * fp32 entry points reports its contact flag, distance, witness points and normal, not the triangle ids (Contact::b1 / b2,
 * DistanceResult::b1 / b2) -- callers that need the primitive ids of mesh pairs use the fp64 entry points. */
typedef struct hfcl_result_f32 {
  float    distance;
  float    p1[3];
  float    p2[3];
  float    normal[3];
  uint32_t status;
} hfcl_result_f32;

/* Compact records: what a caller that only folds the answers (CollisionResult::isCollision(), numContacts(),
 * distance_lower_bound, Contact::b1 / b2 -- collision_data.h:391-494; DistanceResult::min_distance, b1, b2 --
 * :1053-1090) needs of a record, without the witness points and the normal.  They exist for the multi-GPU exchange:
 * an all-gather of full records moves 96 B (44 B fp32) per pair to every rank, which at the measured rates is as long
 * as the compute step itself on 8 GPUs; the compact form is 24 B (8 B).  Produced on the device from
 * full records by hfcl_compact_results_device{,_f32}; every field is a bit copy of the full record's field. */
typedef struct hfcl_result_compact {
  double   distance;
  int32_t  b1, b2;
  uint32_t status;
  int32_t  num_contacts;
} hfcl_result_compact;      /* 24 bytes */
typedef struct hfcl_result_compact_f32 {
  float    distance;
  uint32_t status;
} hfcl_result_compact_f32;  /* 8 bytes */

/* Warm-start cache (QueryRequest::cached_gjk_guess / cached_support_func_guess flowing
 * request -> solver -> result -> request, src/collision.cpp:125-127). Optional arrays. */
typedef struct hfcl_guess {
  double  gjk_guess[3];
  int32_t support_guess[2];
} hfcl_guess;

/* A contact of a mesh-mesh collide query (num_max_contacts > 1). */
typedef struct hfcl_contact {
  uint32_t pair;            /* index of the query in the batch */
  int32_t  b1, b2;
  uint32_t _pad;
  double   penetration_depth;
  double   normal[3];
  double   p1[3];
  double   p2[3];
} hfcl_contact;

typedef struct hfcl_lib hfcl_lib;   /* opaque: shape library resident on one GPU */

/* Error codes (the C++ shim maps them back to the exceptions the reference throws). */
enum {
  HFCL_OK = 0,
  HFCL_ERR_INVALID_ARGUMENT = 1,   /* std::invalid_argument in the reference            */
  HFCL_ERR_UNSUPPORTED_PAIR = 2,   /* "Collision function between node type ... not yet supported" */
  HFCL_ERR_NO_DEVICE = 3,          /* no HIP device / HIP runtime error: fail loudly     */
  HFCL_ERR_HIP = 4,
  HFCL_ERR_LIMIT = 5               /* epa_max_iterations > 64, convex > 64 vertices, ... */
};

/* ---- library lifetime --------------------------------------------------------------- */
int  hfcl_abi_version(void);
/* number of visible HIP devices (0 => every compute entry point returns HFCL_ERR_NO_DEVICE) */
int  hfcl_device_count(void);
/* 1 if a (node_type1, node_type2) pair has an evaluator, 0 otherwise: the lookup the reference makes in its
 * function matrices (getCollisionFunctionLookTable / getDistanceFunctionLookTable, src/collision.cpp:132-158,
 * src/distance.cpp:111-137).  for_distance selects the distance matrix, which -- unlike the collision matrix --
 * has no GEOM_TRIANGLE entries (src/distance_func_matrix.cpp).  Host-only, needs no device. */
int  hfcl_pair_supported(int32_t node_type1, int32_t node_type2, int for_distance);
const char* hfcl_last_error(void);

void hfcl_collision_request_init(hfcl_collision_request* r);
void hfcl_distance_request_init(hfcl_distance_request* r);

/* Copy the shape table and vertex array (n_vertices x 3 doubles) to `device`.
 * Returns NULL (and sets hfcl_last_error) on failure -- never a CPU fallback. */
hfcl_lib* hfcl_lib_create(const hfcl_shape* shapes, size_t n_shapes,
                          const double* vertices, size_t n_vertices, int device);
/* Replace the shape / vertex table of an existing library (same rules as hfcl_lib_create); registered BVH models,
 * workspaces and streams are kept.  Waits for the device.  Shape ids of pairs refer to the new table. */
int hfcl_lib_set_shapes(hfcl_lib* lib, const hfcl_shape* shapes, size_t n_shapes,
                        const double* vertices, size_t n_vertices);
/* Vertex adjacency of a convex shape -- ConvexBase::neighbors (include/hpp/fcl/shape/geometric_shapes.h:ConvexBase,
 * built by Convex<PolygonT>::fillNeighbors, shape/details/convex.hxx:231-280): offsets[num_points + 1] into neighbors[],
 * indices relative to the shape's first vertex.  A hull of at least HFCL_CLIMB_MIN (default 512, environment) vertices
 * that has one answers GJK / EPA support queries by neighbour hill-climbing from the previous answer, as
 * getShapeSupportLog does (src/narrowphase/support_functions.cpp:323-397), instead of scanning every vertex; smaller
 * hulls and hulls without adjacency are unaffected.  hfcl_lib_set_shapes drops all registered adjacencies.
 * The device image is (re)built by the first batch after a registration: the calling thread waits for that upload (a copy
 * stream of the library's own), nothing else on the device is synchronised; the previous image stays allocated until
 * hfcl_lib_set_shapes / hfcl_lib_destroy (batches in flight on other streams may still be reading it). */
int hfcl_lib_set_convex_neighbors(hfcl_lib* lib, uint32_t shape_id, const uint32_t* offsets,
                                  const uint32_t* neighbors);
void      hfcl_lib_destroy(hfcl_lib* lib);
size_t    hfcl_lib_num_shapes(const hfcl_lib* lib);
int       hfcl_lib_device(const hfcl_lib* lib);
/* Smallest hull (vertices) that answers support queries by hill-climbing a registered adjacency (HFCL_CLIMB_MIN, default 512). */
uint32_t  hfcl_lib_climb_min(const hfcl_lib* lib);

/* Register a BVHModel<OBBRSS> (include/hpp/fcl/BVH/BVH_model.h:66-360, BV_node.h:52-148):
 * nodes: n_nodes records of 32 doubles in the reference's field order
 *   [first_child, first_primitive, num_primitives, 0,
 *    obb.axes(col-major 9), obb.To(3), obb.extent(3),
 *    rss.axes(col-major 9) -- ignored, shared with obb --, rss.Tr(3), rss.length[2], rss.radius]
 * (see hfcl_bvh_node below), vertices n_vertices x 3, triangles n_tris x 3 (uint32).
 * Returns the bvh_index to store in hfcl_shape.bvh_index, or -1. */
typedef struct hfcl_bvh_node {
  int32_t first_child;      /* >0: children at first_child, first_child+1; <0: leaf, primitive = -(first_child+1)  (BV_node.h:57-101) */
  int32_t first_primitive;
  int32_t num_primitives;
  int32_t _pad;
  double  obb_axes[9];      /* column-major, OBB.h:52-126 */
  double  obb_To[3];
  double  obb_extent[3];
  double  rss_axes[9];      /* column-major, RSS.h:54-150 */
  double  rss_Tr[3];
  double  rss_length[2];
  double  rss_radius;
} hfcl_bvh_node;

int hfcl_lib_add_bvh(hfcl_lib* lib, const hfcl_bvh_node* nodes, size_t n_nodes,
                     const double* vertices, size_t n_vertices,
                     const uint32_t* triangles, size_t n_tris);

/* Host-side construction of the node array of a BVHModel<OBBRSS> from a triangle soup, as
 * BVHModel<OBBRSS>::beginModel/addSubModel/endModel does (src/BVH/BVH_model.cpp:440-576,858-960;
 * fit src/BVH/BV_fitter.cpp:501-531; split SPLIT_METHOD_MEAN src/BVH/BV_splitter.cpp:81-118,276-279).
 * nodes_out: 2*n_tris-1 records; primitive_indices_out: n_tris (the model's permutation array).
 * n_threads <= 0: pick automatically.  Runs on the host (the reference builds on the host too);
 * needs no GPU. */
int hfcl_bvh_build(const double* vertices, size_t n_vertices, const uint32_t* triangles, size_t n_tris,
                   hfcl_bvh_node* nodes_out, uint32_t* primitive_indices_out, int n_threads);

/* ---- host broadphase: pair-list producer (north_star: "the broadphase ... stays on host and feeds
 * pair lists to the device") -----------------------------------------------------------------
 * hfcl_world_aabbs: world AABB of every posed object as CollisionObject::computeAABB does
 * (include/hpp/fcl/collision_object.h:259-276) from the shape's local AABB
 * (src/shape/geometric_shapes.cpp:145-254).  aabbs_out: n_objects x 6 doubles (min xyz, max xyz).
 * hfcl_broadphase_self_pairs: every (i < j) whose AABBs overlap -- the pairs for which
 * DynamicAABBTreeCollisionManager::collide(callback) invokes the callback
 * (src/broadphase/broadphase_dynamic_AABB_tree.cpp:252-293,393-406,713-721), i.e. what a
 * CollisionCallBackCollect (src/broadphase/default_broadphase_callbacks.cpp:91-100) would hold; order:
 * i ascending, then j ascending.  hfcl_broadphase_pairs_between: the two-manager form (:734-743).
 * Host code, no GPU needed. */
typedef struct hfcl_pairlist hfcl_pairlist;
int hfcl_world_aabbs(const hfcl_shape* shapes, size_t n_shapes, const double* vertices, const uint32_t* object_shape,
                     const double* object_tf, size_t n_objects, double* aabbs_out, int n_threads);
hfcl_pairlist* hfcl_broadphase_self_pairs(const double* aabbs, size_t n_objects, int n_threads);
hfcl_pairlist* hfcl_broadphase_pairs_between(const double* aabbs_a, size_t n_a, const double* aabbs_b, size_t n_b,
                                             int n_threads);
size_t hfcl_pairlist_size(const hfcl_pairlist* pl);
const uint32_t* hfcl_pairlist_data(const hfcl_pairlist* pl);   /* 2 x size uint32: (i, j) */
void hfcl_pairlist_free(hfcl_pairlist* pl);

/* ---- batched queries, host buffers (H2D + kernels + D2H inside the call) ------------
 * shape1/shape2: n indices into the library; tf1/tf2: n poses (12 doubles each).
 * guess_in / guess_out: NULL or n records (used when q.gjk_initial_guess == CachedGuess).
 * Replaces: collide() src/collision.cpp:69-130 ; distance() src/distance.cpp:60-109.
 * The batch flows through a chunked three-stage pipeline on three internal streams (H2D of chunk k+1 | kernels of
 * chunk k | D2H of chunk k-1, straight from / to the caller's arrays); the call returns when every record is in `out`.
 * It synchronises its own streams only, never the device. */
int hfcl_collide_batch(hfcl_lib* lib, const uint32_t* shape1, const uint32_t* shape2,
                       const double* tf1, const double* tf2, size_t n,
                       const hfcl_collision_request* req, hfcl_result* out,
                       const hfcl_guess* guess_in, hfcl_guess* guess_out);

int hfcl_distance_batch(hfcl_lib* lib, const uint32_t* shape1, const uint32_t* shape2,
                        const double* tf1, const double* tf2, size_t n,
                        const hfcl_distance_request* req, hfcl_result* out,
                        const hfcl_guess* guess_in, hfcl_guess* guess_out);

/* Same calls with compact poses: 7 doubles per pose = unit quaternion (w, x, y, z: the order of the fp32 device path
 * below; Transform3f::getQuatRotation(), math/transform.h:108) followed by the translation.  112 instead of 192
 * bytes of pose per pair over the host link, which bounds the host-buffer entry points; the rotation matrix is rebuilt
 * on the device exactly as Eigen's Quaternion::toRotationMatrix does. */
int hfcl_collide_batch_qt(hfcl_lib* lib, const uint32_t* shape1, const uint32_t* shape2,
                          const double* pose1, const double* pose2, size_t n,
                          const hfcl_collision_request* req, hfcl_result* out,
                          const hfcl_guess* guess_in, hfcl_guess* guess_out);
int hfcl_distance_batch_qt(hfcl_lib* lib, const uint32_t* shape1, const uint32_t* shape2,
                           const double* pose1, const double* pose2, size_t n,
                           const hfcl_distance_request* req, hfcl_result* out,
                           const hfcl_guess* guess_in, hfcl_guess* guess_out);
/* pairs per chunk of the host-buffer pipeline; 0 = automatic (n/8 clamped to 32k .. 256k) */
void hfcl_lib_set_host_chunk(hfcl_lib* lib, size_t pairs);

/* ---- batched queries, device-resident buffers (no copies; asynchronous on `stream`,
 * a hipStream_t passed as void*; NULL = the null stream).  All pointers are device
 * pointers on the library's device.  Same semantics as above. */
int hfcl_collide_batch_device(hfcl_lib* lib, const uint32_t* d_shape1, const uint32_t* d_shape2,
                              const double* d_tf1, const double* d_tf2, size_t n,
                              const hfcl_collision_request* req, hfcl_result* d_out,
                              const hfcl_guess* d_guess_in, hfcl_guess* d_guess_out,
                              void* stream);

int hfcl_distance_batch_device(hfcl_lib* lib, const uint32_t* d_shape1, const uint32_t* d_shape2,
                               const double* d_tf1, const double* d_tf2, size_t n,
                               const hfcl_distance_request* req, hfcl_result* d_out,
                               const hfcl_guess* d_guess_in, hfcl_guess* d_guess_out,
                               void* stream);

/* fp32 compute path (the reference has no fp32; parity = fp32 result vs fp64 oracle within
 * the tolerance stated in tests/).  Poses are 7-float (quat wxyz + translation) records,
 * results are 44-byte hfcl_result_f32 records (no primitive ids: see hfcl_result_f32).  Device-resident arrays, or -- hfcl_*_batch_f32
 * below -- host arrays that go through the same chunked copy / compute / copy pipeline as the fp64 host entry points.
 * EPA statuses and iteration counts of this path are not the reference's step for step: the fp32 convex x convex fast tier finds
 * an expansion's horizon without the reference's walk (every face the new vertex is above is removed, connected to the
 * closest face or not; csrc/hfcl_epa.hpp: silhouette_parallel), the depth converges to the same value
 * (tests/test_epa_ground_truth.py). */
int hfcl_distance_batch_device_f32(hfcl_lib* lib, const uint32_t* d_shape1, const uint32_t* d_shape2,
                                   const float* d_pose1, const float* d_pose2, size_t n,
                                   const hfcl_distance_request* req, hfcl_result_f32* d_out,
                                   void* stream);
int hfcl_collide_batch_device_f32(hfcl_lib* lib, const uint32_t* d_shape1, const uint32_t* d_shape2,
                                  const float* d_pose1, const float* d_pose2, size_t n,
                                  const hfcl_collision_request* req, hfcl_result_f32* d_out,
                                  void* stream);
/* The same path from HOST arrays (blocking; records identical to the device-resident calls'): what a caller that holds hpp::fcl objects in host
 * memory and accepts the fp32 envelope uses -- 64 B in and 44 B out per pair over the link instead of 200 B and 96 B.  No cached guesses in this
 * format (the fp32 records carry none).  Replaces the same loop as hfcl_collide_batch / hfcl_distance_batch
 * (src/broadphase/default_broadphase_callbacks.cpp:43-91 over the pairs of a broadphase pass). */
int hfcl_collide_batch_f32(hfcl_lib* lib, const uint32_t* shape1, const uint32_t* shape2, const float* pose1, const float* pose2, size_t n,
                           const hfcl_collision_request* req, hfcl_result_f32* out);
int hfcl_distance_batch_f32(hfcl_lib* lib, const uint32_t* shape1, const uint32_t* shape2, const float* pose1, const float* pose2, size_t n,
                            const hfcl_distance_request* req, hfcl_result_f32* out);

/* Device-resident full records -> compact records (see hfcl_result_compact); asynchronous on `stream`. */
int hfcl_compact_results_device(hfcl_lib* lib, const hfcl_result* d_records, size_t n,
                                hfcl_result_compact* d_out, void* stream);
int hfcl_compact_results_device_f32(hfcl_lib* lib, const hfcl_result_f32* d_records, size_t n,
                                    hfcl_result_compact_f32* d_out, void* stream);

/* Mesh-mesh collide with more than one contact: per-pair records as above plus a
 * compacted contact list (capacity max_contacts_total; *n_contacts_out receives the
 * number produced; contacts beyond capacity are counted but dropped). Host buffers. */
int hfcl_collide_batch_contacts(hfcl_lib* lib, const uint32_t* shape1, const uint32_t* shape2,
                                const double* tf1, const double* tf2, size_t n,
                                const hfcl_collision_request* req, hfcl_result* out,
                                hfcl_contact* contacts, size_t max_contacts_total,
                                size_t* n_contacts_out);

/* ---- instrumentation ---------------------------------------------------------------
 * Milliseconds the GPU spent in the kernels of the most recent *_device call on this
 * library (HIP events on the launch stream); synchronises that stream. */
double hfcl_last_kernel_ms(hfcl_lib* lib);
/* Name of the dominant kernel of the last call (for matching rocprofv3 output). */
const char* hfcl_last_kernel_name(hfcl_lib* lib);
/* (name, milliseconds) of the kernels of the last call, up to `cap` entries; returns the number written.
 * Synchronises on the recorded events. */
int hfcl_last_kernel_breakdown(hfcl_lib* lib, const char** names, double* ms, int cap);
/* Populations of the last call: the 10 kernel buckets (closed, prim, cc, pc, cp, bvh, unsupported, large,
 * bvh_shape, tri) followed by the two EPA queue lengths.  Waits for the device (the counters come back with an
 * asynchronous copy into pinned memory at the end of the batch). */
void hfcl_last_bucket_counts(hfcl_lib* lib, uint32_t* out12);
/* distance() on meshes: the walks of the last call that a wave continued (several walks per wave, their tests pooled) and, of those,
 * the ones walked again in the reference's order because the pooled result could have hung on a bound that exceeds a distance beneath it
 * by a rounding error (see "Mesh distance(): which triangle pair is reported" below).  out4 = {mesh x mesh continued, re-run,
 * mesh x solid continued, re-run}.  Diagnostic; waits for the device. */
void hfcl_last_ordered_reruns(hfcl_lib* lib, uint32_t* out4);
/* Batches of >= 128k pairs (library without meshes) can run as two halves on two streams -- the caller's and an
 * internal one, forked and joined with events, so the call stays asynchronous and ordered on the caller's stream.
 * Pays when the halves run different kernels side by side (mixed scenes: -6 % per batch) and costs ~3 % when the
 * batch has one or two kernels.  parts: 0 = automatic (on when the library's shape kinds spread the pairs over three or
 * more iterative buckets; the default), 1 = never, 2 = always.  Results are bit-identical either way. */
void hfcl_lib_set_split(hfcl_lib* lib, int parts);
int  hfcl_lib_get_split(const hfcl_lib* lib);
int  hfcl_lib_last_split_parts(const hfcl_lib* lib);  /* 1 or 2: how the last batch ran */
/* Tuning options by name.  Every option chooses between forms of the same computation or sizes a budget / a table; none changes a
 * record (the forms are held against each other byte for byte in tests/).  Keys are case-insensitive; hfcl_lib_option_key(0), (1), ...
 * enumerates them (NULL ends the list); INTEGRATION.md describes each.  An option holds from the next batch on.  Returns
 * HFCL_ERR_INVALID_ARGUMENT for an unknown key or a value outside the option's range (nothing is changed then).
 * The environment is a FALLBACK, read once by hfcl_lib_create: HFCL_<KEY IN UPPER CASE>=value sets the same option for a process that
 * cannot be changed to call this function (A/B runs of a built binary); a later hfcl_lib_set_option wins. */
int hfcl_lib_set_option(hfcl_lib* lib, const char* key, const char* value);
/* 1 when this build carries the forms that lost their A/B and are kept as identity references for the tests only (options `bvh_filter`,
 * `epa_general_staged`: compiled with -DHFCL_KEEP_AB_FORMS=1, tools/build_variant.sh); the product build returns 0 and refuses those options. */
int hfcl_has_ab_forms(void);
const char* hfcl_lib_option_key(int index);
/* Per-kernel HIP events are recorded by default; a caller that does not read them can switch
 * them off (on = 0) and save two stream markers per kernel launch. */
void hfcl_lib_set_kernel_timing(hfcl_lib* lib, int on);

/* ---- several devices in one process ("one process, G streams") ---------------------------------------
 * The queries of a batch are independent: a batch over G devices is G contiguous shards of the pair list, one per replica
 * of the library, with no exchange between them -- replica g owns the pairs [lo_g, hi_g) of hfcl_shard_range (ceil(n / G)
 * pairs each, the last ones possibly short or empty).  What the reference offers for this is the collector that hands a
 * caller the broadphase's pair list (CollisionCallBackCollect, src/broadphase/default_broadphase_callbacks.cpp:91-123);
 * the narrow phase over that list is what is sharded here.
 * hfcl_multi_create: one replica of the library per entry of `devices` (a device may be listed more than once: two
 * replicas then share it); every hfcl_multi_* registration call is the hfcl_lib_* call of the same name on every replica
 * (same shape ids and bvh indices everywhere).  NULL + hfcl_last_error() on failure. */
typedef struct hfcl_multi hfcl_multi;
void hfcl_shard_range(size_t n, int rank, int world, size_t* lo, size_t* hi);
hfcl_multi* hfcl_multi_create(const int* devices, int n_devices, const hfcl_shape* shapes, size_t n_shapes,
                              const double* vertices, size_t n_vertices);
void      hfcl_multi_destroy(hfcl_multi* m);
int       hfcl_multi_size(const hfcl_multi* m);
hfcl_lib* hfcl_multi_replica(hfcl_multi* m, int i);
int hfcl_multi_set_shapes(hfcl_multi* m, const hfcl_shape* shapes, size_t n_shapes, const double* vertices, size_t n_vertices);
int hfcl_multi_set_convex_neighbors(hfcl_multi* m, uint32_t shape_id, const uint32_t* offsets, const uint32_t* neighbors);
int hfcl_multi_add_bvh(hfcl_multi* m, const hfcl_bvh_node* nodes, size_t n_nodes, const double* vertices, size_t n_vertices,
                       const uint32_t* triangles, size_t n_tris);
/* A registration that succeeds on some replicas and fails on another (a device out of memory) leaves the replicas different for good:
 * the call returns that replica's error and every later hfcl_multi_* call on this object fails with HFCL_ERR_INVALID_ARGUMENT and a
 * message that says so -- destroy it and create it again.  The caller's current HIP device is the same after every hfcl_multi_* call
 * as before it. */
/* hfcl_lib_set_option on every replica. */
int hfcl_multi_set_option(hfcl_multi* m, const char* key, const char* value);
/* Host buffers: hfcl_collide_batch / hfcl_distance_batch with the pair list cut into the replicas' shards, every shard
 * through its replica's own pipeline (a host thread each), the records straight into the caller's `out` (the host form
 * needs no collective).  The records equal the single-library call's byte for byte. */
int hfcl_collide_batch_multi(hfcl_multi* m, const uint32_t* shape1, const uint32_t* shape2, const double* tf1,
                             const double* tf2, size_t n, const hfcl_collision_request* req, hfcl_result* out,
                             const hfcl_guess* guess_in, hfcl_guess* guess_out);
int hfcl_distance_batch_multi(hfcl_multi* m, const uint32_t* shape1, const uint32_t* shape2, const double* tf1,
                              const double* tf2, size_t n, const hfcl_distance_request* req, hfcl_result* out,
                              const hfcl_guess* guess_in, hfcl_guess* guess_out);
/* ... and through the fp32 path (hfcl_collide_batch_f32 / hfcl_distance_batch_f32 per shard) */
int hfcl_collide_batch_multi_f32(hfcl_multi* m, const uint32_t* shape1, const uint32_t* shape2, const float* pose1, const float* pose2, size_t n,
                                 const hfcl_collision_request* req, hfcl_result_f32* out);
int hfcl_distance_batch_multi_f32(hfcl_multi* m, const uint32_t* shape1, const uint32_t* shape2, const float* pose1, const float* pose2, size_t n,
                                  const hfcl_distance_request* req, hfcl_result_f32* out);
/* Device-resident buffers: replica g finds the inputs of ITS shard on its device (d_shape1[g] ... : hi_g - lo_g entries)
 * and writes into d_gathered[g], a buffer of G * ceil(n / G) records on the same device: its own shard at slot g, then the
 * other replicas' shards arrive by an in-place all-gather of the fixed-size records (ncclAllGather of librccl.so -- RCCL
 * over xGMI --, loaded on first use; every device then holds the whole batch's records, north_star's exchange).
 * Asynchronous on streams[g] (hipStream_t as void*; NULL array: the null streams).  With one replica there is nothing to
 * gather.  HFCL_ERR_INVALID_ARGUMENT when a device is listed twice (one communicator rank per device) ; HFCL_ERR_HIP when librccl.so cannot be loaded. */
int hfcl_collide_batch_multi_device(hfcl_multi* m, const uint32_t* const* d_shape1, const uint32_t* const* d_shape2,
                                    const double* const* d_tf1, const double* const* d_tf2, size_t n,
                                    const hfcl_collision_request* req, hfcl_result* const* d_gathered, void* const* streams);
int hfcl_distance_batch_multi_device(hfcl_multi* m, const uint32_t* const* d_shape1, const uint32_t* const* d_shape2,
                                     const double* const* d_tf1, const double* const* d_tf2, size_t n,
                                     const hfcl_distance_request* req, hfcl_result* const* d_gathered, void* const* streams);
/* The last device-resident batch, for a caller that wants to see what the exchange did: the ranks the communicator reports
 * (ncclCommCount; 1 when there was no collective), the duration of the grouped all-gather on replica 0's stream in milliseconds (HIP
 * events around it; waits for it; < 0: no collective), the bytes each rank contributed.  Bus bandwidth of the all-gather =
 * (ranks - 1) * bytes_per_rank / ms (what each rank receives; DESIGN.md section 5 holds it against 153 GB/s per xGMI link). */
int hfcl_multi_last_gather(hfcl_multi* m, int* ranks, double* ms, size_t* bytes_per_rank);

#ifdef __cplusplus
}
#endif
#endif /* HPPFCL_AMD_H */
