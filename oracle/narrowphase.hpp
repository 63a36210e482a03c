// ORACLE -- TEST INFRASTRUCTURE ONLY.  Not part of the product path (see vec3.hpp).
//
// fp64 CPU restatement of the reference's query layer for shape-shape pairs:
//   GJKSolver            include/hpp/fcl/narrowphase/narrowphase.h:58-724
//   ShapeShapeCollider / ShapeShapeDistancer   include/hpp/fcl/internal/shape_shape_func.h:51-175
//   closed forms         src/narrowphase/details.h:52-101,215-232,435-495,
//                        src/distance/{sphere_sphere,sphere_capsule,capsule_capsule,box_sphere}.cpp
//   collide()/distance() src/collision.cpp:69-130, src/distance.cpp:60-109
#pragma once
#include "../include/hppfcl_amd.h"
#include "gjk.hpp"

namespace orc {

struct SolverStats {
  int gjk_status = 0;
  int epa_status = -1;  // EPA::DidNotRun
  unsigned gjk_iterations = 0;
  unsigned epa_iterations = 0;
};

struct GJKSolver {
  // narrowphase.h:63-108
  size_t gjk_max_iterations = 128;
  double gjk_tolerance = 1e-6;
  int gjk_initial_guess = HFCL_GUESS_DEFAULT;
  V3 cached_guess = V3(1, 0, 0);
  int support_func_cached_guess[2] = {0, 0};
  double distance_upper_bound = std::numeric_limits<double>::max();
  int gjk_variant = DefaultGJK;
  int gjk_convergence_criterion = CritDefault;
  int gjk_convergence_criterion_type = Relative;
  size_t epa_max_iterations = 64;
  double epa_tolerance = 1e-6;
  mutable SolverStats stats;

  void set_query(const hfcl_query_request& q);
  void set(const hfcl_distance_request& r);   // narrowphase.h:162-190
  void set(const hfcl_collision_request& r);  // narrowphase.h:214-244

  // runGJKAndEPA, narrowphase.h:420-587.  relative_precomputed = the TriangleP overloads.
  double run_gjk_epa(const Shape& s1, const Tf& tf1, const Shape& s2, const Tf& tf2, bool compute_penetration,
                     V3& p1, V3& p2, V3& normal, bool relative_precomputed = false) const;

  mutable V3 out_cached_guess;  // solver.cached_guess after the call
  mutable int out_support_guess[2];
};

// internal::ShapeShapeDistance<S1,S2> incl. the closed-form specialisations that exist for
// the shape kinds in scope.  Returns false if the pair is not supported.
bool shape_shape_distance(const Shape& s1, const Tf& tf1, const Shape& s2, const Tf& tf2, const GJKSolver& solver,
                          bool compute_signed_distance, double& dist, V3& p1, V3& p2, V3& normal);

// One pair of hpp::fcl::distance() / collide() on primitives; fills one hfcl_result.
int distance_pair(const Shape& s1, const Tf& tf1, const Shape& s2, const Tf& tf2, const hfcl_distance_request& req,
                  const hfcl_guess* guess_in, hfcl_result& out, hfcl_guess* guess_out);
int collide_pair(const Shape& s1, const Tf& tf1, const Shape& s2, const Tf& tf2, const hfcl_collision_request& req,
                 const hfcl_guess* guess_in, hfcl_result& out, hfcl_guess* guess_out);

void distance_request_defaults(hfcl_distance_request* r);
void collision_request_defaults(hfcl_collision_request* r);

}  // namespace orc
