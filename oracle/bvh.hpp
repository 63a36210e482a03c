// ORACLE -- TEST INFRASTRUCTURE ONLY.  Not part of the product path (see vec3.hpp).
//
// fp64 CPU restatement of BVHModel<OBBRSS> x BVHModel<OBBRSS> collide():
//   OBB overlap test      src/BV/OBB.cpp:290-393 (obbDisjointAndLowerBoundDistance), :475-483 (overlap)
//   traversal             src/traversal/traversal_recurse.cpp:44-85 (collisionRecurse)
//   traversal node        include/hpp/fcl/internal/traversal_node_bvhs.h:89-98 (firstOverSecond),
//                         :152-168 (BVDisjoints), :184-233 (leafCollides)
//   setup                 include/hpp/fcl/internal/traversal_node_setup.h:532-566, src/collision_func_matrix.cpp:187-204
//   lower-bound updates   include/hpp/fcl/collision_data.h:1177-1197
#pragma once
#include <vector>
#include "narrowphase.hpp"

namespace orc {

struct MeshView {
  const hfcl_bvh_node* nodes = nullptr;
  size_t n_nodes = 0;
  const double* verts = nullptr;
  const uint32_t* tris = nullptr;
};

struct BvhStats {
  unsigned num_bv_tests = 0, num_leaf_tests = 0;
};

// Returns HFCL_OK.  `out` = per-pair record (see DESIGN.md "BVH records"); contacts (may be null)
// receives every Contact added (up to num_max_contacts), in the reference's DFS order.
int bvh_collide_pair(const MeshView& m1, const Tf& tf1, const MeshView& m2, const Tf& tf2,
                     const hfcl_collision_request& req, hfcl_result& out, std::vector<hfcl_contact>* contacts,
                     uint32_t pair_index, BvhStats* stats);

// BVHModel<OBBRSS> x convex shape collide() (oracle/bvh_shape.cpp).  `swapped`: the caller's order was
// (shape, BVH); the record is returned in the caller's order (src/collision.cpp:93-108).
int bvh_shape_collide_pair(const MeshView& m1, const Tf& tf1, const Shape& s2, const Tf& tf2, const hfcl_collision_request& req,
                           bool swapped, hfcl_result& out, std::vector<hfcl_contact>* contacts, uint32_t pair_index,
                           hfcl_guess* guess_out, BvhStats* stats);
double bvh_shape_leaf_distance(const MeshView& m1, const Tf& tf1, const Shape& s2, const Tf& tf2, const hfcl_distance_request& req, int pid);
size_t bvh_shape_distance_trace(const MeshView& m1, const Tf& tf1, const Shape& s2, const Tf& tf2, const hfcl_distance_request& req, double* out, size_t cap);
int bvh_shape_distance_pair(const MeshView& m1, const Tf& tf1, const Shape& s2, const Tf& tf2, const hfcl_distance_request& req,
                            bool swapped, hfcl_result& out, hfcl_guess* guess_out);
// computeBV<OBBRSS,S>(shape, tf): fit of the shape's bound vertices (geometric_shapes_utility.h:73-82)
int shape_obbrss(const Shape& s, const Tf& tf, hfcl_bvh_node& bv);

// OBB overlap (exposed for unit tests): returns true when NOT disjoint.
bool obb_overlap(const M3& R0, const V3& T0, const hfcl_bvh_node& b1, const hfcl_bvh_node& b2, double security_margin,
                 double break_distance, double& sqrDistLowerBound);

}  // namespace orc

namespace orc {
// ---- BVHModel<OBBRSS> distance() ------------------------------------------------------------
//   rectDistance / segCoords / inVoronoi   src/BV/RSS.cpp:49-713
//   distance(R0,T0,rss1,rss2)              src/BV/RSS.cpp:995-1005 (OBBRSS forwards to RSS, OBBRSS.h:151-154)
//   segPoints / sqrTriDistance             src/intersect.cpp:60-384
//   distanceRecurse                        src/traversal/traversal_recurse.cpp:153-203
//   MeshDistanceTraversalNode<OBBRSS,0>    include/hpp/fcl/internal/traversal_node_bvhs.h:386-531
double rect_distance(const M3& Rab, const V3& Tab, const double a[2], const double b[2]);
double rss_distance(const M3& R0, const V3& T0, const hfcl_bvh_node& b1, const hfcl_bvh_node& b2);
double sqr_tri_distance(const V3 S[3], const V3 T[3], V3& P, V3& Q);
int bvh_distance_pair(const MeshView& m1, const Tf& tf1, const MeshView& m2, const Tf& tf2, hfcl_result& out,
                      BvhStats* stats);
double bvh_leaf_distance(const MeshView& m1, const Tf& tf1, const MeshView& m2, const Tf& tf2, int pid1, int pid2);
}  // namespace orc
