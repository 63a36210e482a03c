// ORACLE -- TEST INFRASTRUCTURE ONLY.  Not part of the product path (see vec3.hpp).
//
// fp64 CPU restatement of hpp-fcl's GJK / EPA / MinkowskiDiff / support functions,
// following the reference's control flow branch-for-branch so that statuses, iteration
// counts and numbers agree with it.  Reference files restated here:
//   src/narrowphase/support_functions.cpp:110-222, 400-421
//   src/narrowphase/minkowski_difference.cpp:47-63, 78-285
//   src/narrowphase/gjk.cpp:94-186 (closest points, inflate), 188-435 (GJK::evaluate),
//       437-492 (encloseOrigin), 494-1010 (simplex projections), 1012-1466 (EPA)
//   src/intersect.cpp:570-705 (Project::project*Origin)
// Differences in representation only: simplex vertices are held by value instead of
// by pointer into store_v (the free_v bookkeeping has no numerical effect); EPA faces
// are addressed by index instead of by pointer.
#pragma once
#include <cstdint>
#include <vector>
#include "vec3.hpp"

namespace orc {

enum ShapeKind { K_BOX = 9, K_SPHERE = 10, K_CAPSULE = 11, K_CONE = 12, K_CYLINDER = 13, K_CONVEX = 14, K_PLANE = 15, K_HALFSPACE = 16, K_TRIANGLE = 17, K_ELLIPSOID = 19 };

struct Shape {
  int kind = 0;
  double p[4] = {0, 0, 0, 0};   // Box halfSide / Sphere r / Capsule r,halfLength / Ellipsoid radii / Plane, Halfspace n,d
  double ssr = 0;            // swept sphere radius
  const double* verts = nullptr;  // CONVEX / TRIANGLE vertices (xyz)
  int nverts = 0;
  // ConvexBase::neighbors (CSR, ids ascending: fillNeighbors, details/convex.hxx:231-280) and
  // ConvexBase::support_warm_starts (buildSupportWarmStart, gjk.cpp:1470-1534); only read for
  // num_points > 32 (minkowski_difference.cpp:136-151)
  const uint32_t* nbr_off = nullptr;
  const uint32_t* nbr = nullptr;
  int n_warm = 0;
  double warm_pts[14][3];
  int warm_idx[14];
};

// details::ShapeSupportData (support_functions.h:79-85)
struct SupportData {
  V3 last_dir = V3(0, 0, 0);
  std::vector<int8_t> visited;
};

// getShapeSupportLog (support_functions.cpp:323-397, NoSweptSphere): neighbour hill climbing
inline V3 convex_support_log(const Shape& s, const V3& dir, int& hint, SupportData& sd) {
  const double use_warm_start_threshold = 0.9;
  const V3 dir_normalized = normalized(dir);
  const bool last_zero = std::abs(sd.last_dir.x) <= 1e-12 && std::abs(sd.last_dir.y) <= 1e-12 && std::abs(sd.last_dir.z) <= 1e-12;
  if (!last_zero && s.n_warm > 0 && dot(sd.last_dir, dir_normalized) < use_warm_start_threshold) {
    double maxdot = s.warm_pts[0][0] * dir.x + s.warm_pts[0][1] * dir.y + s.warm_pts[0][2] * dir.z;
    hint = s.warm_idx[0];
    for (int i = 1; i < s.n_warm; ++i) {
      const double d = s.warm_pts[i][0] * dir.x + s.warm_pts[i][1] * dir.y + s.warm_pts[i][2] * dir.z;
      if (d > maxdot) {
        maxdot = d;
        hint = s.warm_idx[i];
      }
    }
  }
  sd.last_dir = dir_normalized;
  if (hint < 0 || hint >= s.nverts) hint = 0;
  auto pt = [&](int i) { return V3(s.verts[3 * i], s.verts[3 * i + 1], s.verts[3 * i + 2]); };
  double maxdot = dot(pt(hint), dir);
  sd.visited.assign(size_t(s.nverts), 0);
  sd.visited[size_t(hint)] = 1;
  bool found = true, loose_check = true;
  while (found) {
    const uint32_t b = s.nbr_off[hint], e = s.nbr_off[hint + 1];
    found = false;
    for (uint32_t in = b; in < e; ++in) {
      const uint32_t ip = s.nbr[in];
      if (sd.visited[ip]) continue;
      sd.visited[ip] = 1;
      const double d = dot(pt(int(ip)), dir);
      bool better = false;
      if (d > maxdot) {
        better = true;
        loose_check = false;
      } else if (loose_check && d == maxdot)
        better = true;
      if (better) {
        maxdot = d;
        hint = int(ip);
        found = true;
      }
    }
  }
  return pt(hint);
}

enum GJKVariant { DefaultGJK = 0, PolyakAcceleration = 1, NesterovAcceleration = 2 };
enum GJKCrit { CritDefault = 0, CritDualityGap = 1, CritHybrid = 2 };
enum GJKCritType { Relative = 0, Absolute = 1 };

// Box support's `inflate` is a function-local static evaluated on the FIRST call in the
// process (support_functions.cpp:146): 1+1e-10 if that first direction had a zero
// component, else 1.  In a process whose first Box support is GJK's default first
// direction (-1,0,0) applied to shape 0 it is 1+1e-10; we pin that value.
constexpr double kBoxInflate = 1 + 1e-10;
constexpr double kDummyPrecision = 1e-12;  // Eigen::NumTraits<double>::dummy_precision()

// support_functions.cpp:110-222, 400-421 (NoSweptSphere option only: that is what
// GJKSolver::runGJKAndEPA instantiates, narrowphase.h:420-421)
inline V3 shape_support(const Shape& s, const V3& dir, int& hint, SupportData* sd = nullptr) {
  if (s.kind == K_CONVEX && s.nverts > 32 && s.nbr_off && sd) return convex_support_log(s, dir, hint, *sd);
  switch (s.kind) {
    case K_TRIANGLE: {  // :110-134
      V3 a(s.verts[0], s.verts[1], s.verts[2]), b(s.verts[3], s.verts[4], s.verts[5]),
          c(s.verts[6], s.verts[7], s.verts[8]);
      double da = dot(dir, a), db = dot(dir, b), dc = dot(dir, c);
      if (da > db) return (dc > da) ? c : a;
      return (dc > db) ? c : b;
    }
    case K_BOX: {  // :140-157
      V3 r;
      for (int i = 0; i < 3; ++i) {
        double s1 = (dir[i] > kDummyPrecision) ? s.p[i] : 0.0;
        double s2 = (dir[i] < -kDummyPrecision) ? (-kBoxInflate * s.p[i]) : 0.0;
        r[i] = s1 + s2;
      }
      return r;
    }
    case K_SPHERE:  // :163-176
      return V3(0, 0, 0);
    case K_ELLIPSOID: {  // :182-199
      double a2 = s.p[0] * s.p[0], b2 = s.p[1] * s.p[1], c2 = s.p[2] * s.p[2];
      V3 v(a2 * dir[0], b2 * dir[1], c2 * dir[2]);
      double d = std::sqrt(dot(v, dir));
      return v / d;
    }
    case K_CAPSULE: {  // :205-222
      V3 r(0, 0, 0);
      if (dir[2] > kDummyPrecision)
        r[2] = s.p[1];
      else if (dir[2] < -kDummyPrecision)
        r[2] = -s.p[1];
      return r;
    }
    case K_CONE: {  // :228-274  p[0] = radius, p[1] = halfLength
      const double inflate = 1 + 1e-10;
      const double h = s.p[1], r = s.p[0];
      V3 sup(0, 0, 0);
      if (std::abs(dir[0]) <= kDummyPrecision && std::abs(dir[1]) <= kDummyPrecision) {  // head<2>().isZero()
        sup[2] = (dir[2] > kDummyPrecision) ? h : -inflate * h;
      } else {
        double zdist = dir[0] * dir[0] + dir[1] * dir[1];
        double len = zdist + dir[2] * dir[2];
        zdist = std::sqrt(zdist);
        if (dir[2] <= 0) {
          const double rad = r / zdist;
          sup[0] = rad * dir[0];
          sup[1] = rad * dir[1];
          sup[2] = -h;
        } else {
          len = std::sqrt(len);
          const double sin_a = r / std::sqrt(r * r + 4 * h * h);
          if (dir[2] > len * sin_a) {
            sup = V3(0, 0, h);
          } else {
            const double rad = r / zdist;
            sup[0] = rad * dir[0];
            sup[1] = rad * dir[1];
            sup[2] = -h;
          }
        }
      }
      return sup;
    }
    case K_CYLINDER: {  // :280-317
      const double inflate = 1 + 1e-10;
      double half_h = s.p[1], r = s.p[0];
      const bool aligned = std::abs(dir[0]) <= kDummyPrecision && std::abs(dir[1]) <= kDummyPrecision;
      if (aligned) half_h *= inflate;
      V3 sup(0, 0, 0);
      if (dir[2] > kDummyPrecision) {
        sup[2] = half_h;
      } else if (dir[2] < -kDummyPrecision) {
        sup[2] = -half_h;
      } else {
        sup[2] = 0;
        r *= inflate;
      }
      if (!aligned) {  // dir.head<2>().normalized() * r
        const double n2 = dir[0] * dir[0] + dir[1] * dir[1];
        double nx = dir[0], ny = dir[1];
        if (n2 > 0) {
          const double n = std::sqrt(n2);
          nx = dir[0] / n;
          ny = dir[1] / n;
        }
        sup[0] = nx * r;
        sup[1] = ny * r;
      }
      return sup;
    }
    case K_CONVEX: {  // getShapeSupportLinear :400-421 (num_points <= 32 path)
      hint = 0;
      double maxdot = s.verts[0] * dir.x + s.verts[1] * dir.y + s.verts[2] * dir.z;
      for (int i = 1; i < s.nverts; ++i) {
        const double* p = s.verts + 3 * i;
        double d = p[0] * dir.x + p[1] * dir.y + p[2] * dir.z;
        if (d > maxdot) {
          maxdot = d;
          hint = i;
        }
      }
      const double* p = s.verts + 3 * hint;
      return V3(p[0], p[1], p[2]);
    }
  }
  return V3(0, 0, 0);
}

// minkowski_difference.cpp:78-285, include/hpp/fcl/narrowphase/minkowski_difference.h
struct MinkowskiDiff {
  const Shape* shapes[2] = {nullptr, nullptr};
  M3 oR1 = M3::identity();
  V3 ot1;
  bool identity = true;
  double swept_sphere_radius[2] = {0, 0};
  bool normalize_support_direction = false;

  static double radius_of(const Shape& s) {  // :89,103-125 / :170,183-201
    double r = s.ssr;
    if (s.kind == K_SPHERE || s.kind == K_CAPSULE) r += s.p[0];
    return r;
  }
  mutable SupportData data[2];  // minkowski_difference.h:76-77, reset by set() (:136-151 / :196-211)
  void set_common(const Shape* s0, const Shape* s1) {
    shapes[0] = s0;
    shapes[1] = s1;
    data[0] = SupportData();
    data[1] = SupportData();
    // geometric_shapes_traits.h:135-144 : only ConvexBase needs the heuristic; :261-266
    normalize_support_direction = (s0->kind == K_CONVEX) && (s1->kind == K_CONVEX);
    swept_sphere_radius[0] = radius_of(*s0);
    swept_sphere_radius[1] = radius_of(*s1);
  }
  void set(const Shape* s0, const Shape* s1, const Tf& tf0, const Tf& tf1) {  // :269-285
    set_common(s0, s1);
    oR1 = tmul(tf0.R, tf1.R);
    ot1 = tmul(tf0.R, tf1.T - tf0.T);
    identity = is_identity(oR1) && is_zero(ot1);
  }
  void set(const Shape* s0, const Shape* s1) {  // :293-305 (relative transform precomputed)
    set_common(s0, s1);
    oR1 = M3::identity();
    ot1 = V3(0, 0, 0);
    identity = true;
  }
  // getSupportTpl :47-63
  void support(const V3& dir, V3& s0, V3& s1, int hint[2]) const {
    s0 = shape_support(*shapes[0], dir, hint[0], &data[0]);
    if (identity) {
      s1 = shape_support(*shapes[1], -dir, hint[1], &data[1]);
    } else {
      s1 = shape_support(*shapes[1], -tmul(oR1, dir), hint[1], &data[1]);
      s1 = oR1 * s1 + ot1;
    }
  }
};

struct SimplexV {
  V3 w0, w1, w;
};
struct Simplex {
  SimplexV v[4];
  int rank = 0;
};

// src/intersect.cpp:570-705 -------------------------------------------------------------
struct ProjectResult {
  double param[4] = {0, 0, 0, 0};  // (uninitialised in the reference; zeroed here)
  double sqr_distance = -1;
  unsigned encode = 0;
};
ProjectResult project_line_origin(const V3& a, const V3& b);
ProjectResult project_triangle_origin(const V3& a, const V3& b, const V3& c);
ProjectResult project_tetrahedra_origin(const V3& a, const V3& b, const V3& c, const V3& d);

// gjk.cpp:94-151
void get_closest_points(const Simplex& simplex, V3& w0, V3& w1);

struct GJK {
  enum Status { DidNotRun = 0, Failed, NoCollisionEarlyStopped, NoCollision,
                CollisionWithPenetrationInformation, Collision };
  // parameters (gjk.h:105-109,125-126)
  double distance_upper_bound = std::numeric_limits<double>::max();
  int gjk_variant = DefaultGJK;
  int convergence_criterion = CritDefault;
  int convergence_criterion_type = Relative;
  size_t max_iterations = 128;
  double tolerance = 1e-6;
  // state
  Status status = DidNotRun;
  const MinkowskiDiff* shape = nullptr;
  V3 ray;
  int support_hint[2] = {0, 0};
  double distance = 0;
  Simplex simplex;  // result of the last run
  size_t iterations = 0, iterations_momentum_stop = 0;

  GJK(size_t max_it, double tol) : max_iterations(max_it), tolerance(tol) {}

  Status evaluate(const MinkowskiDiff& shape, const V3& guess, const int hint_in[2]);
  void get_support(const V3& d, SimplexV& sv, int hint[2]) const {  // gjk.h:163-167
    shape->support(d, sv.w0, sv.w1, hint);
    sv.w = sv.w0 - sv.w1;
  }
  bool enclose_origin();  // gjk.cpp:437-492 (operates on this->simplex)
  void get_witness_points_and_normal(const MinkowskiDiff& shape, V3& w0, V3& w1, V3& normal) const;

 private:
  bool check_convergence(const V3& w, double rl, double& alpha, double omega) const;
  bool project_line(const Simplex& cur, Simplex& next);
  bool project_triangle(const Simplex& cur, Simplex& next);
  bool project_tetra(const Simplex& cur, Simplex& next);
};

struct EPA {
  enum Status { DidNotRun = -1, Failed = 0, Valid = 1, AccuracyReached = (1 << 1) | 1,
                Degenerated = (1 << 1) | 0, NonConvex = (2 << 1) | 0, InvalidHull = (3 << 1) | 0,
                OutOfFaces = (4 << 1) | 0, OutOfVertices = (5 << 1) | 0, FallBack = (6 << 1) | 0 };
  struct Face {
    V3 n;
    double d = 0;
    bool ignore = false;
    size_t vertex_id[3] = {0, 0, 0};
    int adjacent_faces[3] = {-1, -1, -1};
    int prev = -1, next = -1;
    size_t adjacent_edge[3] = {0, 0, 0};
    size_t pass = 0;
  };
  struct FaceList {  // gjk.h:282-308
    int root = -1;
    size_t count = 0;
  };
  Status status = DidNotRun;
  Simplex result;
  V3 normal;
  int support_hint[2] = {0, 0};
  double depth = 0;
  size_t max_iterations;
  double tolerance;
  size_t iterations = 0;
  size_t num_vertices = 0;

  EPA(size_t max_it, double tol) : max_iterations(max_it), tolerance(tol) { reset(max_it, tol); }
  void reset(size_t max_it, double tol);
  Status evaluate(GJK& gjk, const V3& guess);
  void get_witness_points_and_normal(const MinkowskiDiff& shape, V3& w0, V3& w1, V3& normal) const;
  size_t num_faces() const { return hull.count; }

 private:
  std::vector<SimplexV> sv_store;
  std::vector<Face> fc_store;
  FaceList hull, stock;
  int closest_face = -1;
  void list_append(FaceList& l, int f);
  void list_remove(FaceList& l, int f);
  void bind(int fa, size_t ea, int fb, size_t eb);
  int new_face(size_t a, size_t b, size_t c, bool force = false);
  int find_closest_face();
  struct Horizon {
    int current_face = -1, first_face = -1;
    size_t num_faces = 0;
  };
  bool expand(size_t pass, const SimplexV& w, int f, size_t e, Horizon& horizon);
};

}  // namespace orc
