// hppfcl_amd_compat.hpp -- header-only C++ shim exposing the C ABI (hppfcl_amd.h) under the
// reference's own names so hpp-fcl user code for the narrow-phase path recompiles against it.
//
// Mirrors (same names, argument meaning, defaults and error behaviour):
//   Transform3f                 include/hpp/fcl/math/transform.h:56-218
//   CollisionGeometry/ShapeBase include/hpp/fcl/collision_object.h:94-190, shape/geometric_shapes.h:59-102
//   Box, Sphere, Capsule, Ellipsoid, ConvexBase   shape/geometric_shapes.h:164-187,238-251,381-400,303-320,638-872
//   QueryRequest, CollisionRequest, DistanceRequest, Contact, CollisionResult, DistanceResult
//                               include/hpp/fcl/collision_data.h:59-166,171-273,312-383,391-494,987-1174
//   collide(), distance()       include/hpp/fcl/collision.h:58-70, distance.h:53-65 (src/collision.cpp:69-130,
//                               src/distance.cpp:60-109); std::invalid_argument for num_max_contacts == 0 and for
//                               unsupported node-type pairs, exactly where the reference throws.
//   CollisionCallBackCollect-style batching: amd::BatchQueries (default_broadphase_callbacks.h:224-252)
//
// No Eigen: Vec3f / Matrix3f are minimal PODs with the same memory image as the Eigen types the
// reference uses (Matrix3f column-major), so `Transform3f` is bit-compatible with the ABI pose.
// There is no CPU fallback: every query runs on the GPU through libhppfcl_amd.so.
#pragma once
#include <array>
#include <cmath>
#include <cstddef>
#include <limits>
#include <map>
#include <memory>
#include <stdexcept>
#include <string>
#include <utility>
#include <typeinfo>
#include <vector>

#include <algorithm>
#include <cstring>

#include "hppfcl_amd.h"

namespace hpp {
namespace fcl {

typedef double FCL_REAL;

struct Vec3f {
  FCL_REAL v[3];
  Vec3f() : v{0, 0, 0} {}
  Vec3f(FCL_REAL x, FCL_REAL y, FCL_REAL z) : v{x, y, z} {}
  FCL_REAL& operator[](int i) { return v[i]; }
  const FCL_REAL& operator[](int i) const { return v[i]; }
  const FCL_REAL* data() const { return v; }
  static Vec3f Constant(FCL_REAL c) { return Vec3f(c, c, c); }
  static Vec3f Zero() { return Vec3f(0, 0, 0); }
  Vec3f operator+(const Vec3f& o) const { return Vec3f(v[0] + o.v[0], v[1] + o.v[1], v[2] + o.v[2]); }
  Vec3f operator-(const Vec3f& o) const { return Vec3f(v[0] - o.v[0], v[1] - o.v[1], v[2] - o.v[2]); }
  Vec3f operator*(FCL_REAL s) const { return Vec3f(v[0] * s, v[1] * s, v[2] * s); }
  Vec3f operator/(FCL_REAL s) const { return Vec3f(v[0] / s, v[1] / s, v[2] / s); }
  Vec3f operator-() const { return Vec3f(-v[0], -v[1], -v[2]); }
  FCL_REAL dot(const Vec3f& o) const { return v[0] * o.v[0] + v[1] * o.v[1] + v[2] * o.v[2]; }
  FCL_REAL norm() const { return std::sqrt(dot(*this)); }
};

struct Matrix3f {  // column-major, like Eigen::Matrix<double,3,3>
  FCL_REAL m[9];
  Matrix3f() { setIdentity(); }
  void setIdentity() {
    for (int i = 0; i < 9; ++i) m[i] = (i % 4 == 0) ? 1.0 : 0.0;
  }
  static Matrix3f Identity() { return Matrix3f(); }
  FCL_REAL& operator()(int r, int c) { return m[c * 3 + r]; }
  const FCL_REAL& operator()(int r, int c) const { return m[c * 3 + r]; }
  Vec3f operator*(const Vec3f& x) const {
    return Vec3f((*this)(0, 0) * x[0] + (*this)(0, 1) * x[1] + (*this)(0, 2) * x[2],
                 (*this)(1, 0) * x[0] + (*this)(1, 1) * x[1] + (*this)(1, 2) * x[2],
                 (*this)(2, 0) * x[0] + (*this)(2, 1) * x[1] + (*this)(2, 2) * x[2]);
  }
  Matrix3f operator*(const Matrix3f& o) const {
    Matrix3f r;
    for (int i = 0; i < 3; ++i)
      for (int j = 0; j < 3; ++j) r(i, j) = (*this)(i, 0) * o(0, j) + (*this)(i, 1) * o(1, j) + (*this)(i, 2) * o(2, j);
    return r;
  }
};

struct Quatf {  // makeQuat(w, x, y, z)
  FCL_REAL w, x, y, z;
  Matrix3f toRotationMatrix() const {  // Eigen::Quaternion::toRotationMatrix
    Matrix3f R;
    const FCL_REAL tx = 2 * x, ty = 2 * y, tz = 2 * z, twx = tx * w, twy = ty * w, twz = tz * w, txx = tx * x,
                   txy = ty * x, txz = tz * x, tyy = ty * y, tyz = tz * y, tzz = tz * z;
    R(0, 0) = 1 - (tyy + tzz); R(0, 1) = txy - twz; R(0, 2) = txz + twy;
    R(1, 0) = txy + twz; R(1, 1) = 1 - (txx + tzz); R(1, 2) = tyz - twx;
    R(2, 0) = txz - twy; R(2, 1) = tyz + twx; R(2, 2) = 1 - (txx + tyy);
    return R;
  }
};
inline Quatf makeQuat(FCL_REAL w, FCL_REAL x, FCL_REAL y, FCL_REAL z) { return Quatf{w, x, y, z}; }

class Transform3f {
  Matrix3f R;
  Vec3f T;

 public:
  Transform3f() {}
  Transform3f(const Matrix3f& R_, const Vec3f& T_) : R(R_), T(T_) {}
  Transform3f(const Quatf& q, const Vec3f& T_) : R(q.toRotationMatrix()), T(T_) {}
  explicit Transform3f(const Matrix3f& R_) : R(R_) {}
  explicit Transform3f(const Quatf& q) : R(q.toRotationMatrix()) {}
  explicit Transform3f(const Vec3f& T_) : T(T_) {}
  static Transform3f Identity() { return Transform3f(); }
  const Vec3f& getTranslation() const { return T; }
  const Matrix3f& getRotation() const { return R; }
  void setTranslation(const Vec3f& t) { T = t; }
  void setRotation(const Matrix3f& r) { R = r; }
  void setQuatRotation(const Quatf& q) { R = q.toRotationMatrix(); }
  Vec3f transform(const Vec3f& x) const { return R * x + T; }
  Transform3f operator*(const Transform3f& o) const { return Transform3f(R * o.R, R * o.T + T); }
};
static_assert(sizeof(Transform3f) == HFCL_POSE_DOUBLES * sizeof(double), "Transform3f must be the ABI pose image");

enum NODE_TYPE {  // include/hpp/fcl/collision_object.h:65-89 (subset in scope)
  BV_OBBRSS = HFCL_BV_OBBRSS, GEOM_BOX = HFCL_GEOM_BOX, GEOM_SPHERE = HFCL_GEOM_SPHERE, GEOM_CAPSULE = HFCL_GEOM_CAPSULE,
  GEOM_CONE = HFCL_GEOM_CONE, GEOM_CYLINDER = HFCL_GEOM_CYLINDER, GEOM_PLANE = HFCL_GEOM_PLANE,
  GEOM_HALFSPACE = HFCL_GEOM_HALFSPACE,
  GEOM_CONVEX = HFCL_GEOM_CONVEX, GEOM_TRIANGLE = HFCL_GEOM_TRIANGLE, GEOM_ELLIPSOID = HFCL_GEOM_ELLIPSOID
};
enum GJKInitialGuess { DefaultGuess, CachedGuess, BoundingVolumeGuess };
enum GJKVariant { DefaultGJK, PolyakAcceleration, NesterovAcceleration };
enum GJKConvergenceCriterion { Default, DualityGap, Hybrid };
enum GJKConvergenceCriterionType { Relative, Absolute };
typedef std::array<int, 2> support_func_guess_t;

class CollisionGeometry {
 public:
  virtual ~CollisionGeometry() {}
  virtual NODE_TYPE getNodeType() const = 0;
};
class ShapeBase : public CollisionGeometry {
 public:
  void setSweptSphereRadius(FCL_REAL r) {
    if (r < 0) throw std::invalid_argument("Swept-sphere radius must be positive.");
    m_swept_sphere_radius = r;
  }
  FCL_REAL getSweptSphereRadius() const { return m_swept_sphere_radius; }

 protected:
  FCL_REAL m_swept_sphere_radius = 0;
};
class Box : public ShapeBase {
 public:
  Box(FCL_REAL x, FCL_REAL y, FCL_REAL z) : halfSide(x / 2, y / 2, z / 2) {}
  explicit Box(const Vec3f& side) : halfSide(side * 0.5) {}
  Vec3f halfSide;
  NODE_TYPE getNodeType() const override { return GEOM_BOX; }
};
class Sphere : public ShapeBase {
 public:
  explicit Sphere(FCL_REAL r) : radius(r) {}
  FCL_REAL radius;
  NODE_TYPE getNodeType() const override { return GEOM_SPHERE; }
};
class Capsule : public ShapeBase {
 public:
  Capsule(FCL_REAL r, FCL_REAL lz) : radius(r), halfLength(lz / 2) {}
  FCL_REAL radius, halfLength;
  NODE_TYPE getNodeType() const override { return GEOM_CAPSULE; }
};
class Cone : public ShapeBase {
 public:
  Cone(FCL_REAL r, FCL_REAL lz) : radius(r), halfLength(lz / 2) {}
  FCL_REAL radius, halfLength;
  NODE_TYPE getNodeType() const override { return GEOM_CONE; }
};
class Cylinder : public ShapeBase {
 public:
  Cylinder(FCL_REAL r, FCL_REAL lz) : radius(r), halfLength(lz / 2) {}
  FCL_REAL radius, halfLength;
  NODE_TYPE getNodeType() const override { return GEOM_CYLINDER; }
};
class Halfspace : public ShapeBase {  // {x : n.x <= d}, geometric_shapes.h:873-962
 public:
  Halfspace(const Vec3f& n_, FCL_REAL d_) : n(n_), d(d_) { unitNormalTest(); }
  Halfspace(FCL_REAL a, FCL_REAL b, FCL_REAL c, FCL_REAL d_) : n(a, b, c), d(d_) { unitNormalTest(); }
  Halfspace() : n(1, 0, 0), d(0) {}
  Vec3f n;
  FCL_REAL d;
  FCL_REAL signedDistance(const Vec3f& p) const { return n.dot(p) - (d + getSweptSphereRadius()); }
  NODE_TYPE getNodeType() const override { return GEOM_HALFSPACE; }

 private:
  void unitNormalTest() {  // geometric_shapes.cpp:121-131
    const FCL_REAL l = n.norm();
    if (l > 0) { n = n * (1.0 / l); d *= 1.0 / l; } else { n = Vec3f(1, 0, 0); d = 0; }
  }
};
class Plane : public ShapeBase {  // {x : n.x = d}, geometric_shapes.h:968-1049
 public:
  Plane(const Vec3f& n_, FCL_REAL d_) : n(n_), d(d_) { unitNormalTest(); }
  Plane(FCL_REAL a, FCL_REAL b, FCL_REAL c, FCL_REAL d_) : n(a, b, c), d(d_) { unitNormalTest(); }
  Plane() : n(1, 0, 0), d(0) {}
  Vec3f n;
  FCL_REAL d;
  NODE_TYPE getNodeType() const override { return GEOM_PLANE; }

 private:
  void unitNormalTest() {  // geometric_shapes.cpp:133-143
    const FCL_REAL l = n.norm();
    if (l > 0) { n = n * (1.0 / l); d *= 1.0 / l; } else { n = Vec3f(1, 0, 0); d = 0; }
  }
};
class Ellipsoid : public ShapeBase {
 public:
  Ellipsoid(FCL_REAL rx, FCL_REAL ry, FCL_REAL rz) : radii(rx, ry, rz) {}
  Vec3f radii;
  NODE_TYPE getNodeType() const override { return GEOM_ELLIPSOID; }
};
class ConvexBase : public ShapeBase {
 public:
  explicit ConvexBase(std::shared_ptr<std::vector<Vec3f>> pts) : points(std::move(pts)) {
    num_points = static_cast<unsigned int>(points->size());
  }
  std::shared_ptr<std::vector<Vec3f>> points;
  unsigned int num_points;
  /// ConvexBase::neighbors (geometric_shapes.h: per vertex the vertices it shares a facet edge with), kept as CSR:
  /// neighbor_offsets[num_points + 1] into neighbor_ids.  Empty: none known (Convex<PolygonT> fills them).  The engine
  /// climbs them for hulls of HFCL_CLIMB_MIN vertices and more (hfcl_lib_set_convex_neighbors).
  std::vector<uint32_t> neighbor_offsets, neighbor_ids;
  NODE_TYPE getNodeType() const override { return GEOM_CONVEX; }
};

class TriangleP : public ShapeBase {  // geometric_shapes.h:98-134
 public:
  TriangleP(const Vec3f& a_, const Vec3f& b_, const Vec3f& c_) : a(a_), b(b_), c(c_) {}
  Vec3f a, b, c;
  NODE_TYPE getNodeType() const override { return GEOM_TRIANGLE; }
};

struct Triangle {  // include/hpp/fcl/data_types.h:101-144
  typedef std::size_t index_type;
  Triangle() : vids{0, 0, 0} {}
  Triangle(index_type a, index_type b, index_type c) : vids{a, b, c} {}
  index_type operator[](index_type i) const { return vids[i]; }
  index_type& operator[](index_type i) { return vids[i]; }
  index_type vids[3];
};
/// Convex<PolygonT> (include/hpp/fcl/shape/convex.h:49-105): a convex polytope given by its vertices and facets; the
/// constructor derives the vertex adjacency from the facets (fillNeighbors, shape/details/convex.hxx:231-280: the
/// ascending set of the two facet-cycle neighbours of every vertex over all facets).
template <typename PolygonT>
class Convex : public ConvexBase {
 public:
  Convex(std::shared_ptr<std::vector<Vec3f>> points_, unsigned int num_points_, std::shared_ptr<std::vector<PolygonT>> polygons_,
         unsigned int num_polygons_)
      : ConvexBase(std::move(points_)), polygons(std::move(polygons_)), num_polygons(num_polygons_) {
    if (num_points_ < num_points) num_points = num_points_;
    fillNeighbors();
  }
  std::shared_ptr<std::vector<PolygonT>> polygons;
  unsigned int num_polygons;

 protected:
  void fillNeighbors() {
    std::vector<std::vector<uint32_t>> nb(num_points);
    if (!polygons) return;
    for (unsigned int l = 0; l < num_polygons && l < polygons->size(); ++l) {
      const PolygonT& poly = (*polygons)[l];
      const std::size_t n = poly_size(poly);
      for (std::size_t j = 0; j < n; ++j) {
        const std::size_t pi = poly[static_cast<typename PolygonT::index_type>(j == 0 ? n - 1 : j - 1)], pj = poly[static_cast<typename PolygonT::index_type>(j)],
                          pk = poly[static_cast<typename PolygonT::index_type>(j == n - 1 ? 0 : j + 1)];
        if (pj >= num_points || pi >= num_points || pk >= num_points) throw std::invalid_argument("Convex: polygon vertex index out of range");
        nb[pj].push_back(static_cast<uint32_t>(pi));
        nb[pj].push_back(static_cast<uint32_t>(pk));
      }
    }
    neighbor_offsets.assign(num_points + 1, 0);
    neighbor_ids.clear();
    for (unsigned int i = 0; i < num_points; ++i) {
      std::sort(nb[i].begin(), nb[i].end());
      nb[i].erase(std::unique(nb[i].begin(), nb[i].end()), nb[i].end());
      neighbor_ids.insert(neighbor_ids.end(), nb[i].begin(), nb[i].end());
      neighbor_offsets[i + 1] = static_cast<uint32_t>(neighbor_ids.size());
    }
  }
  static std::size_t poly_size(const Triangle&) { return 3; }
  template <typename P>
  static std::size_t poly_size(const P& p) { return p.size(); }
};

struct OBBRSS {};  // tag: the one BV type in scope (include/hpp/fcl/BV/OBBRSS.h)
enum BVHReturnCode { BVH_OK = 0, BVH_ERR_BUILD_OUT_OF_SEQUENCE = -2, BVH_ERR_BUILD_EMPTY_MODEL = -3 };

/// BVHModel<OBBRSS> (include/hpp/fcl/BVH/BVH_model.h:61-368): triangle soup assembled with
/// beginModel / addVertex / addTriangle / addSubModel / endModel; endModel() builds the tree on the
/// host (hfcl_bvh_build, the reference's SPLIT_METHOD_MEAN construction).
template <typename BV>
class BVHModel : public CollisionGeometry {
 public:
  std::vector<Vec3f> vertices;
  std::vector<Triangle> tri_indices;
  unsigned int num_tris = 0, num_vertices = 0;

  NODE_TYPE getNodeType() const override { return BV_OBBRSS; }
  int beginModel(unsigned int = 0, unsigned int = 0) {  // BVH_model.cpp:264-306
    vertices.clear();
    tri_indices.clear();
    nodes_.clear();
    building_ = true;
    return BVH_OK;
  }
  int addVertex(const Vec3f& p) {
    if (!building_) return BVH_ERR_BUILD_OUT_OF_SEQUENCE;
    vertices.push_back(p);
    return BVH_OK;
  }
  int addTriangle(const Vec3f& p1, const Vec3f& p2, const Vec3f& p3) {  // BVH_model.cpp:385-438
    if (!building_) return BVH_ERR_BUILD_OUT_OF_SEQUENCE;
    const std::size_t o = vertices.size();
    vertices.push_back(p1);
    vertices.push_back(p2);
    vertices.push_back(p3);
    tri_indices.emplace_back(o, o + 1, o + 2);
    return BVH_OK;
  }
  int addSubModel(const std::vector<Vec3f>& ps, const std::vector<Triangle>& ts) {  // BVH_model.cpp:440-506
    if (!building_) return BVH_ERR_BUILD_OUT_OF_SEQUENCE;
    const std::size_t o = vertices.size();
    vertices.insert(vertices.end(), ps.begin(), ps.end());
    for (const Triangle& t : ts) tri_indices.emplace_back(t[0] + o, t[1] + o, t[2] + o);
    return BVH_OK;
  }
  int endModel() {  // BVH_model.cpp:508-576
    if (!building_) return BVH_ERR_BUILD_OUT_OF_SEQUENCE;
    if (tri_indices.empty() || vertices.empty()) return BVH_ERR_BUILD_EMPTY_MODEL;
    num_tris = static_cast<unsigned int>(tri_indices.size());
    num_vertices = static_cast<unsigned int>(vertices.size());
    flat_tris_.resize(3 * tri_indices.size());
    for (std::size_t i = 0; i < tri_indices.size(); ++i)
      for (int k = 0; k < 3; ++k) flat_tris_[3 * i + k] = static_cast<uint32_t>(tri_indices[i][k]);
    nodes_.resize(2 * tri_indices.size() - 1);
    primitive_indices_.resize(tri_indices.size());
    const int rc = hfcl_bvh_build(reinterpret_cast<const double*>(vertices.data()), vertices.size(), flat_tris_.data(),
                                  tri_indices.size(), nodes_.data(), primitive_indices_.data(), 0);
    if (rc) throw std::invalid_argument(hfcl_last_error());
    building_ = false;
    return BVH_OK;
  }
  unsigned int getNumBVs() const { return static_cast<unsigned int>(nodes_.size()); }
  const hfcl_bvh_node& getBV(unsigned int i) const { return nodes_[i]; }
  const std::vector<hfcl_bvh_node>& nodes() const { return nodes_; }
  const std::vector<uint32_t>& flatTriangles() const { return flat_tris_; }

 private:
  bool building_ = false;
  std::vector<hfcl_bvh_node> nodes_;
  std::vector<uint32_t> flat_tris_, primitive_indices_;
};

struct QueryRequest {
  GJKInitialGuess gjk_initial_guess = DefaultGuess;
  mutable Vec3f cached_gjk_guess = Vec3f(1, 0, 0);
  mutable support_func_guess_t cached_support_func_guess = {{0, 0}};
  size_t gjk_max_iterations = 128;
  FCL_REAL gjk_tolerance = 1e-6;
  GJKVariant gjk_variant = DefaultGJK;
  GJKConvergenceCriterion gjk_convergence_criterion = Default;
  GJKConvergenceCriterionType gjk_convergence_criterion_type = Relative;
  size_t epa_max_iterations = 64;
  FCL_REAL epa_tolerance = 1e-6;
  bool enable_timings = false;
  FCL_REAL collision_distance_threshold = 1e-12;
};
struct CollisionRequest : QueryRequest {
  size_t num_max_contacts = 1;
  bool enable_contact = true;
  FCL_REAL security_margin = 0;
  FCL_REAL break_distance = 1e-3;
  FCL_REAL distance_upper_bound = (std::numeric_limits<FCL_REAL>::max)();
};
struct DistanceRequest : QueryRequest {
  bool enable_nearest_points = true;
  bool enable_signed_distance = true;
  FCL_REAL rel_err = 0, abs_err = 0;
  DistanceRequest(bool enable_nearest_points_ = true, bool enable_signed_distance_ = true, FCL_REAL rel_err_ = 0,
                  FCL_REAL abs_err_ = 0)
      : enable_nearest_points(enable_nearest_points_), enable_signed_distance(enable_signed_distance_),
        rel_err(rel_err_), abs_err(abs_err_) {}
};

struct Contact {
  const CollisionGeometry* o1 = nullptr;
  const CollisionGeometry* o2 = nullptr;
  int b1 = -1, b2 = -1;
  Vec3f normal;
  std::array<Vec3f, 2> nearest_points;
  Vec3f pos;
  FCL_REAL penetration_depth = (std::numeric_limits<FCL_REAL>::max)();
  static const int NONE = -1;
};

struct QueryResult {
  Vec3f cached_gjk_guess;
  support_func_guess_t cached_support_func_guess = {{-1, -1}};
};
struct CollisionResult : QueryResult {
  FCL_REAL distance_lower_bound = (std::numeric_limits<FCL_REAL>::max)();
  Vec3f normal = Vec3f::Constant(std::numeric_limits<FCL_REAL>::quiet_NaN());
  std::array<Vec3f, 2> nearest_points = {{normal, normal}};
  bool isCollision() const { return !contacts.empty(); }
  size_t numContacts() const { return contacts.size(); }
  const Contact& getContact(size_t i) const {
    if (contacts.empty()) throw std::invalid_argument("The number of contacts is zero. No Contact can be returned.");
    return contacts[i < contacts.size() ? i : contacts.size() - 1];
  }
  void addContact(const Contact& c) { contacts.push_back(c); }
  void clear() { *this = CollisionResult(); }

 private:
  std::vector<Contact> contacts;
};
struct DistanceResult : QueryResult {
  FCL_REAL min_distance = (std::numeric_limits<FCL_REAL>::max)();
  Vec3f normal = Vec3f::Constant(std::numeric_limits<FCL_REAL>::quiet_NaN());
  std::array<Vec3f, 2> nearest_points = {{normal, normal}};
  const CollisionGeometry* o1 = nullptr;
  const CollisionGeometry* o2 = nullptr;
  int b1 = -1, b2 = -1;
  static const int NONE = -1;
  void clear() { *this = DistanceResult(); }
};

namespace amd {

inline void fill_query(hfcl_query_request& q, const QueryRequest& r) {
  q.gjk_initial_guess = r.gjk_initial_guess;
  q.gjk_variant = r.gjk_variant;
  q.gjk_convergence_criterion = r.gjk_convergence_criterion;
  q.gjk_convergence_criterion_type = r.gjk_convergence_criterion_type;
  q.gjk_max_iterations = static_cast<uint32_t>(r.gjk_max_iterations);
  q.epa_max_iterations = static_cast<uint32_t>(r.epa_max_iterations);
  q.gjk_tolerance = r.gjk_tolerance;
  q.epa_tolerance = r.epa_tolerance;
  q.collision_distance_threshold = r.collision_distance_threshold;
  for (int k = 0; k < 3; ++k) q.cached_gjk_guess[k] = r.cached_gjk_guess[k];
  q.cached_support_func_guess[0] = r.cached_support_func_guess[0];
  q.cached_support_func_guess[1] = r.cached_support_func_guess[1];
}
inline hfcl_collision_request to_abi(const CollisionRequest& r) {
  hfcl_collision_request a;
  hfcl_collision_request_init(&a);
  fill_query(a.q, r);
  a.num_max_contacts = static_cast<uint32_t>(r.num_max_contacts);
  a.enable_contact = r.enable_contact;
  a.security_margin = r.security_margin;
  a.break_distance = r.break_distance;
  a.distance_upper_bound = r.distance_upper_bound;
  return a;
}
inline hfcl_distance_request to_abi(const DistanceRequest& r) {
  hfcl_distance_request a;
  hfcl_distance_request_init(&a);
  fill_query(a.q, r);
  a.enable_nearest_points = r.enable_nearest_points;
  a.enable_signed_distance = r.enable_signed_distance;
  a.rel_err = r.rel_err;
  a.abs_err = r.abs_err;
  return a;
}
[[noreturn]] inline void throw_for(int rc) {
  const std::string msg = hfcl_last_error();
  if (rc == HFCL_ERR_INVALID_ARGUMENT || rc == HFCL_ERR_UNSUPPORTED_PAIR) throw std::invalid_argument(msg);
  throw std::runtime_error(msg);
}

// A set of geometries registered once (the device shape library) + batched queries on it:
// the CollisionCallBackCollect hand-off (collect pairs on the host, evaluate them in one call).
class BatchQueries {
 public:
  explicit BatchQueries(int device = 0) : devices_(1, device) {}
  /// Several devices of this process: the library is replicated on each and every batch is cut into contiguous shards, one per
  /// device (hfcl_multi_*, include/hppfcl_amd.h); the results are the single-device ones.  A device may be listed more than once.
  explicit BatchQueries(const std::vector<int>& devices) : devices_(devices.empty() ? std::vector<int>(1, 0) : devices) {}
  ~BatchQueries() { hfcl_multi_destroy(lib_); }
  size_t numDevices() const { return devices_.size(); }
  BatchQueries(const BatchQueries&) = delete;
  BatchQueries& operator=(const BatchQueries&) = delete;

  /// Forget every registered geometry (and free the device library): a long-lived context that has seen many temporary
  /// geometries -- the thread's default one behind collide() / distance() -- is pruned this way.
  void reset() {
    hfcl_multi_destroy(lib_);
    lib_ = nullptr;
    shapes_.clear();
    verts_.clear();
    geoms_.clear();
    ids_.clear();
    meshes_.clear();
    rec_.clear();
    guess_.clear();
    contacts_.clear();
    shapes_dirty_ = false;
    meshes_uploaded_ = 0;
    adjacency_.clear();
    adjacency_pending_ = false;
  }
  size_t numGeometries() const { return shapes_.size(); }

  uint32_t add(const CollisionGeometry* g) {
    hfcl_shape s{};
    std::vector<double> v;
    s.type = g->getNodeType();
    const ShapeBase* sb = dynamic_cast<const ShapeBase*>(g);
    s.swept_sphere_radius = sb ? sb->getSweptSphereRadius() : 0.0;
    switch (g->getNodeType()) {
      case GEOM_BOX: { auto* b = static_cast<const Box*>(g); for (int i = 0; i < 3; ++i) s.params[i] = b->halfSide[i]; break; }
      case GEOM_SPHERE: s.params[0] = static_cast<const Sphere*>(g)->radius; break;
      case GEOM_CAPSULE: { auto* c = static_cast<const Capsule*>(g); s.params[0] = c->radius; s.params[1] = c->halfLength; break; }
      case GEOM_CONE: { auto* c = static_cast<const Cone*>(g); s.params[0] = c->radius; s.params[1] = c->halfLength; break; }
      case GEOM_CYLINDER: { auto* c = static_cast<const Cylinder*>(g); s.params[0] = c->radius; s.params[1] = c->halfLength; break; }
      case GEOM_ELLIPSOID: { auto* e = static_cast<const Ellipsoid*>(g); for (int i = 0; i < 3; ++i) s.params[i] = e->radii[i]; break; }
      case GEOM_HALFSPACE: { auto* h = static_cast<const Halfspace*>(g); for (int i = 0; i < 3; ++i) s.params[i] = h->n[i]; s.params[3] = h->d; break; }
      case GEOM_PLANE: { auto* h = static_cast<const Plane*>(g); for (int i = 0; i < 3; ++i) s.params[i] = h->n[i]; s.params[3] = h->d; break; }
      case GEOM_CONVEX: {
        auto* c = static_cast<const ConvexBase*>(g);
        s.num_points = c->num_points;
        for (unsigned int k = 0; k < c->num_points; ++k) v.insert(v.end(), (*c->points)[k].data(), (*c->points)[k].data() + 3);
        break;
      }
      case GEOM_TRIANGLE: {
        auto* t = static_cast<const TriangleP*>(g);
        s.num_points = 3;
        for (const Vec3f* p : {&t->a, &t->b, &t->c}) v.insert(v.end(), p->data(), p->data() + 3);
        break;
      }
      case BV_OBBRSS: {
        auto* m = dynamic_cast<const BVHModel<OBBRSS>*>(g);
        if (!m || m->getNumBVs() == 0) throw std::invalid_argument("BVHModel: endModel() has not been called");
        auto it = ids_.find(g);
        // same object, same content?  (the address may have belonged to another geometry before)
        if (it != ids_.end() && shapes_[it->second].type == BV_OBBRSS &&
            static_cast<size_t>(shapes_[it->second].bvh_index) < meshes_.size()) {
          const MeshRef& r = meshes_[static_cast<size_t>(shapes_[it->second].bvh_index)];
          if (r.model == m && r.n_nodes == m->getNumBVs() && r.n_vertices == m->num_vertices &&
              r.checksum == checksum(*m))
            return it->second;
        }
        s.bvh_index = static_cast<int32_t>(meshes_.size());
        s.num_points = m->num_vertices;
        // the context keeps its own copy: the model may be destroyed before the library is (re)built
        meshes_.push_back(MeshRef{m, m->getNumBVs(), m->num_vertices, checksum(*m), m->nodes(),
                                  std::vector<double>(reinterpret_cast<const double*>(m->vertices.data()),
                                                      reinterpret_cast<const double*>(m->vertices.data()) + 3 * m->vertices.size()),
                                  m->flatTriangles()});
        break;
      }
      default: throw std::invalid_argument("unsupported node type");
    }
    // The cache is keyed by address; a hit is only trusted if the geometry's *content* is still
    // what was registered (addresses get reused once a geometry is destroyed).
    auto it = ids_.find(g);
    if (it != ids_.end() && s.type != BV_OBBRSS) {
      const hfcl_shape& o = shapes_[it->second];
      bool same = o.type == s.type && o.num_points == s.num_points && o.swept_sphere_radius == s.swept_sphere_radius &&
                  o.params[0] == s.params[0] && o.params[1] == s.params[1] && o.params[2] == s.params[2] &&
                  o.params[3] == s.params[3];
      if (same && (s.type == GEOM_CONVEX || s.type == GEOM_TRIANGLE))  // both carry their vertices in verts_
        for (size_t k = 0; k < v.size() && same; ++k) same = verts_[3 * size_t(o.vertex_offset) + k] == v[k];
      if (same) return it->second;
    }
    s.vertex_offset = static_cast<uint32_t>(verts_.size() / 3);
    verts_.insert(verts_.end(), v.begin(), v.end());
    shapes_.push_back(s);
    geoms_.push_back(g);
    shapes_dirty_ = true;  // the device tables follow at the next query (hfcl_lib_set_shapes: meshes and workspaces stay)
    const uint32_t id = static_cast<uint32_t>(shapes_.size() - 1);
    ids_[g] = id;
    if (s.type == GEOM_CONVEX) {  // a copy, like the vertices: the geometry may be gone by the next query
      auto* c = static_cast<const ConvexBase*>(g);
      if (c->neighbor_offsets.size() == size_t(c->num_points) + 1) {
        adjacency_[id] = std::make_pair(c->neighbor_offsets, c->neighbor_ids);
        adjacency_pending_ = true;
      }
    }
    return id;
  }

  void collide(const std::vector<std::pair<uint32_t, uint32_t>>& pairs, const std::vector<Transform3f>& tf1,
               const std::vector<Transform3f>& tf2, const CollisionRequest& request, std::vector<CollisionResult>& results) {
    run(pairs, tf1, tf2, &request, nullptr);
    results.assign(pairs.size(), CollisionResult());
    for (size_t i = 0; i < pairs.size(); ++i) fill(results[i], pairs[i], request, rec_[i], guess_[i]);
  }
  void distance(const std::vector<std::pair<uint32_t, uint32_t>>& pairs, const std::vector<Transform3f>& tf1,
                const std::vector<Transform3f>& tf2, const DistanceRequest& request, std::vector<DistanceResult>& results) {
    run(pairs, tf1, tf2, nullptr, &request);
    results.assign(pairs.size(), DistanceResult());
    for (size_t i = 0; i < pairs.size(); ++i) fill(results[i], pairs[i], rec_[i], guess_[i]);
  }

  void fill(CollisionResult& res, const std::pair<uint32_t, uint32_t>& p, const CollisionRequest& request,
            const hfcl_result& r, const hfcl_guess& g) const {
    if (!HFCL_STATUS_SKIPPED(r.status)) {
      const Vec3f n(r.normal[0], r.normal[1], r.normal[2]), p1(r.p1[0], r.p1[1], r.p1[2]), p2(r.p2[0], r.p2[1], r.p2[2]);
      const FCL_REAL dtc = r.distance - request.security_margin;  // updateDistanceLowerBoundFromLeaf
      if (dtc < res.distance_lower_bound) {
        res.distance_lower_bound = dtc;
        res.nearest_points = {{p1, p2}};
        res.normal = n;
      }
      if (r.num_contacts > 1 && !contacts_.empty()) {
        // contacts_ is ordered by query (run() sorts it once): this query's contacts are one contiguous range
        const uint32_t qi = static_cast<uint32_t>(&r - rec_.data());
        auto lo = std::lower_bound(contacts_.begin(), contacts_.end(), qi,
                                   [](const hfcl_contact& c, uint32_t q) { return c.pair < q; });
        for (; lo != contacts_.end() && lo->pair == qi; ++lo) {
          const hfcl_contact& k = *lo;
          if (res.numContacts() >= request.num_max_contacts) break;
          Contact c;
          c.o1 = geoms_[p.first];
          c.o2 = geoms_[p.second];
          c.b1 = k.b1;
          c.b2 = k.b2;
          c.normal = Vec3f(k.normal[0], k.normal[1], k.normal[2]);
          c.nearest_points = {{Vec3f(k.p1[0], k.p1[1], k.p1[2]), Vec3f(k.p2[0], k.p2[1], k.p2[2])}};
          c.pos = (c.nearest_points[0] + c.nearest_points[1]) / 2;
          c.penetration_depth = k.penetration_depth;
          res.addContact(c);
        }
      } else if (r.num_contacts > 0 && res.numContacts() < request.num_max_contacts) {
        Contact c;
        c.o1 = geoms_[p.first];
        c.o2 = geoms_[p.second];
        c.b1 = r.b1;
        c.b2 = r.b2;
        c.normal = n;
        c.nearest_points = {{p1, p2}};
        c.pos = (p1 + p2) / 2;
        c.penetration_depth = r.distance;
        res.addContact(c);
      }
    }
    res.cached_gjk_guess = Vec3f(g.gjk_guess[0], g.gjk_guess[1], g.gjk_guess[2]);
    res.cached_support_func_guess = {{g.support_guess[0], g.support_guess[1]}};
  }
  void fill(DistanceResult& res, const std::pair<uint32_t, uint32_t>& p, const hfcl_result& r, const hfcl_guess& g) const {
    if (res.min_distance > r.distance) {  // DistanceResult::update
      res.min_distance = r.distance;
      res.o1 = geoms_[p.first];
      res.o2 = geoms_[p.second];
      res.b1 = r.b1;
      res.b2 = r.b2;
      res.nearest_points = {{Vec3f(r.p1[0], r.p1[1], r.p1[2]), Vec3f(r.p2[0], r.p2[1], r.p2[2])}};
      res.normal = Vec3f(r.normal[0], r.normal[1], r.normal[2]);
    }
    res.cached_gjk_guess = Vec3f(g.gjk_guess[0], g.gjk_guess[1], g.gjk_guess[2]);
    res.cached_support_func_guess = {{g.support_guess[0], g.support_guess[1]}};
  }
  const std::vector<hfcl_result>& records() const { return rec_; }
  const std::vector<hfcl_contact>& contacts() const { return contacts_; }  // filled when num_max_contacts > 1 on meshes
  const std::vector<hfcl_guess>& guesses() const { return guess_; }

  void run(const std::vector<std::pair<uint32_t, uint32_t>>& pairs, const std::vector<Transform3f>& tf1,
           const std::vector<Transform3f>& tf2, const CollisionRequest* creq, const DistanceRequest* dreq) {
    if (tf1.size() != pairs.size() || tf2.size() != pairs.size()) throw std::invalid_argument("pairs/poses size mismatch");
    if (!lib_) {
      lib_ = hfcl_multi_create(devices_.data(), static_cast<int>(devices_.size()), shapes_.data(), shapes_.size(), verts_.data(), verts_.size() / 3);
      if (!lib_) throw std::runtime_error(hfcl_last_error());
      meshes_uploaded_ = 0;
    } else if (shapes_dirty_) {  // geometries were added since the last query: new shape tables, everything else stays
      const int rc = hfcl_multi_set_shapes(lib_, shapes_.data(), shapes_.size(), verts_.data(), verts_.size() / 3);
      if (rc) throw_for(rc);
    }
    if (shapes_dirty_ || adjacency_pending_) {  // (hfcl_lib_set_shapes dropped the adjacencies of the old table)
      for (const auto& kv : adjacency_) {
        const int rc = hfcl_multi_set_convex_neighbors(lib_, kv.first, kv.second.first.data(), kv.second.second.data());
        if (rc) throw_for(rc);
      }
      adjacency_pending_ = false;
    }
    shapes_dirty_ = false;
    for (; meshes_uploaded_ < meshes_.size(); ++meshes_uploaded_) {  // bvh_index = registration order
      const MeshRef& r = meshes_[meshes_uploaded_];
      if (hfcl_multi_add_bvh(lib_, r.nodes.data(), r.nodes.size(), r.verts.data(), r.verts.size() / 3, r.tris.data(),
                             r.tris.size() / 3) < 0)
        throw std::runtime_error(hfcl_last_error());
    }
    std::vector<uint32_t> s1(pairs.size()), s2(pairs.size());
    for (size_t i = 0; i < pairs.size(); ++i) {
      s1[i] = pairs[i].first;
      s2[i] = pairs[i].second;
    }
    rec_.resize(pairs.size());
    guess_.resize(pairs.size());
    int rc;
    contacts_.clear();
    if (creq && creq->num_max_contacts > 1 && !meshes_.empty()) {
      // mesh-mesh queries can produce several contacts each: use the contact-list entry point
      const hfcl_collision_request a = to_abi(*creq);
      size_t cap = std::max<size_t>(1024, 64 * pairs.size()), produced = 0;
      for (int attempt = 0; attempt < 2; ++attempt) {
        contacts_.resize(cap);
        // (contact lists are appended through one device counter: the first replica takes the whole batch)
        rc = hfcl_collide_batch_contacts(hfcl_multi_replica(lib_, 0), s1.data(), s2.data(), reinterpret_cast<const double*>(tf1.data()),
                                         reinterpret_cast<const double*>(tf2.data()), pairs.size(), &a, rec_.data(),
                                         contacts_.data(), cap, &produced);
        if (rc || produced <= cap) break;
        cap = produced;
      }
      contacts_.resize(rc ? 0 : std::min(produced, cap));
      // the device appends contacts in completion order: group them by query once (the order of one query's
      // contacts -- its traversal order -- is kept), so that fill() finds a query's range by binary search
      std::stable_sort(contacts_.begin(), contacts_.end(),
                       [](const hfcl_contact& x, const hfcl_contact& y) { return x.pair < y.pair; });
      for (auto& g : guess_) g = hfcl_guess{{1, 0, 0}, {0, 0}};
    } else if (creq) {
      const hfcl_collision_request a = to_abi(*creq);
      rc = hfcl_collide_batch_multi(lib_, s1.data(), s2.data(), reinterpret_cast<const double*>(tf1.data()),
                                    reinterpret_cast<const double*>(tf2.data()), pairs.size(), &a, rec_.data(), nullptr, guess_.data());
    } else {
      const hfcl_distance_request a = to_abi(*dreq);
      rc = hfcl_distance_batch_multi(lib_, s1.data(), s2.data(), reinterpret_cast<const double*>(tf1.data()),
                                     reinterpret_cast<const double*>(tf2.data()), pairs.size(), &a, rec_.data(), nullptr, guess_.data());
    }
    if (rc) throw_for(rc);
  }

 private:
  std::vector<int> devices_;
  hfcl_multi* lib_ = nullptr;  // one replica of the library per device (one device: a single library behind the same calls)
  std::vector<hfcl_shape> shapes_;
  std::vector<double> verts_;
  std::vector<const CollisionGeometry*> geoms_;
  std::map<const CollisionGeometry*, uint32_t> ids_;
  std::vector<hfcl_result> rec_;
  std::vector<hfcl_guess> guess_;
  std::vector<hfcl_contact> contacts_;
  bool shapes_dirty_ = false;
  std::map<uint32_t, std::pair<std::vector<uint32_t>, std::vector<uint32_t>>> adjacency_;  // shape id -> ConvexBase::neighbors (CSR)
  bool adjacency_pending_ = false;
  size_t meshes_uploaded_ = 0;
  struct MeshRef {
    const BVHModel<OBBRSS>* model;
    unsigned int n_nodes, n_vertices;
    double checksum;
    std::vector<hfcl_bvh_node> nodes;
    std::vector<double> verts;
    std::vector<uint32_t> tris;
  };
  std::vector<MeshRef> meshes_;
  static double checksum(const BVHModel<OBBRSS>& m) {  // cheap content fingerprint for the address-keyed cache
    double c = 0;
    for (size_t i = 0; i < m.vertices.size(); i += 1 + m.vertices.size() / 64) c += m.vertices[i][0] + 2 * m.vertices[i][1] + 3 * m.vertices[i][2];
    return c + m.getBV(0).rss_radius;
  }
};

inline BatchQueries& default_context() {
  static thread_local BatchQueries ctx(0);
  return ctx;
}

}  // namespace amd

/// hpp::fcl::collide (src/collision.cpp:69-130) as a batch of one.  Results accumulate in
/// `result` like in the reference (the caller clears between queries).
inline std::size_t collide(const CollisionGeometry* o1, const Transform3f& tf1, const CollisionGeometry* o2,
                           const Transform3f& tf2, const CollisionRequest& request, CollisionResult& result) {
  if (request.security_margin == -std::numeric_limits<FCL_REAL>::infinity()) {
    result.clear();
    return 0;
  }
  if (request.num_max_contacts == 0)
    throw std::invalid_argument("Invalid number of max contacts (current value is 0).");
  if (result.isCollision() && request.num_max_contacts <= result.numContacts()) return result.numContacts();
  amd::BatchQueries& ctx = amd::default_context();
  const std::pair<uint32_t, uint32_t> p(ctx.add(o1), ctx.add(o2));
  ctx.run({p}, {tf1}, {tf2}, &request, nullptr);
  ctx.fill(result, p, request, ctx.records()[0], ctx.guesses()[0]);
  if (request.gjk_initial_guess == CachedGuess) {  // QueryRequest::updateGuess
    request.cached_gjk_guess = result.cached_gjk_guess;
    request.cached_support_func_guess = result.cached_support_func_guess;
  }
  return result.numContacts();
}

/// hpp::fcl::distance (src/distance.cpp:60-109) as a batch of one.
inline FCL_REAL distance(const CollisionGeometry* o1, const Transform3f& tf1, const CollisionGeometry* o2,
                         const Transform3f& tf2, const DistanceRequest& request, DistanceResult& result) {
  if (result.min_distance <= 0) return result.min_distance;  // DistanceRequest::isSatisfied
  amd::BatchQueries& ctx = amd::default_context();
  const std::pair<uint32_t, uint32_t> p(ctx.add(o1), ctx.add(o2));
  ctx.run({p}, {tf1}, {tf2}, nullptr, &request);
  ctx.fill(result, p, ctx.records()[0], ctx.guesses()[0]);
  if (request.gjk_initial_guess == CachedGuess) {
    request.cached_gjk_guess = result.cached_gjk_guess;
    request.cached_support_func_guess = result.cached_support_func_guess;
  }
  return ctx.records()[0].distance;
}

/// ComputeCollision / ComputeDistance (include/hpp/fcl/collision.h:79-117, distance.h:74-112): the geometry pair is
/// looked up once at construction (std::invalid_argument for a pair without an evaluator, src/collision.cpp:132-158,
/// src/distance.cpp:111-137); every call is the batch-of-one path of collide() / distance().
class ComputeCollision {
 public:
  ComputeCollision(const CollisionGeometry* o1_, const CollisionGeometry* o2_) : o1(o1_), o2(o2_) {
    if (!hfcl_pair_supported(o1->getNodeType(), o2->getNodeType(), 0))
      throw std::invalid_argument("Collision function between the two node types is not yet supported.");
  }
  virtual ~ComputeCollision() {}
  std::size_t operator()(const Transform3f& tf1, const Transform3f& tf2, const CollisionRequest& request,
                         CollisionResult& result) const {
    return collide(o1, tf1, o2, tf2, request, result);
  }
  bool operator==(const ComputeCollision& other) const { return o1 == other.o1 && o2 == other.o2; }
  bool operator!=(const ComputeCollision& other) const { return !(*this == other); }

 protected:
  const CollisionGeometry* o1;
  const CollisionGeometry* o2;
};
class ComputeDistance {
 public:
  ComputeDistance(const CollisionGeometry* o1_, const CollisionGeometry* o2_) : o1(o1_), o2(o2_) {
    if (!hfcl_pair_supported(o1->getNodeType(), o2->getNodeType(), 1))
      throw std::invalid_argument("Distance function between the two node types is not yet supported.");
  }
  virtual ~ComputeDistance() {}
  FCL_REAL operator()(const Transform3f& tf1, const Transform3f& tf2, const DistanceRequest& request,
                      DistanceResult& result) const {
    return distance(o1, tf1, o2, tf2, request, result);
  }
  bool operator==(const ComputeDistance& other) const { return o1 == other.o1 && o2 == other.o2; }
  bool operator!=(const ComputeDistance& other) const { return !(*this == other); }

 protected:
  const CollisionGeometry* o1;
  const CollisionGeometry* o2;
};

// ---------------------------------------------------------------------------------------------
// Broadphase hand-off (include/hpp/fcl/collision_object.h:215-357, broadphase/broadphase_callbacks.h,
// broadphase/default_broadphase_callbacks.h:200-230, broadphase/broadphase_dynamic_AABB_tree.h).
// ---------------------------------------------------------------------------------------------
struct AABB {
  Vec3f min_, max_;
  bool overlap(const AABB& o) const {  // BV/AABB.h:112-122
    for (int k = 0; k < 3; ++k)
      if (min_[k] > o.max_[k] || max_[k] < o.min_[k]) return false;
    return true;
  }
};

class CollisionObject {
 public:
  CollisionObject(const std::shared_ptr<CollisionGeometry>& g, const Transform3f& tf = Transform3f()) : cgeom(g), t(tf) {}
  const std::shared_ptr<CollisionGeometry>& collisionGeometry() const { return cgeom; }
  const CollisionGeometry* collisionGeometryPtr() const { return cgeom.get(); }
  const Transform3f& getTransform() const { return t; }
  void setTransform(const Transform3f& tf) { t = tf; }
  const AABB& getAABB() const { return aabb; }
  AABB& getAABB() { return aabb; }

 private:
  std::shared_ptr<CollisionGeometry> cgeom;
  Transform3f t;
  AABB aabb;
};

// collide() / distance() on CollisionObjects (src/collision.cpp:60-67, src/distance.cpp:51-58)
inline std::size_t collide(const CollisionObject* o1, const CollisionObject* o2, const CollisionRequest& request,
                           CollisionResult& result) {
  return collide(o1->collisionGeometryPtr(), o1->getTransform(), o2->collisionGeometryPtr(), o2->getTransform(), request, result);
}
inline FCL_REAL distance(const CollisionObject* o1, const CollisionObject* o2, const DistanceRequest& request,
                         DistanceResult& result) {
  return distance(o1->collisionGeometryPtr(), o1->getTransform(), o2->collisionGeometryPtr(), o2->getTransform(), request, result);
}

struct CollisionCallBackBase {  // broadphase/broadphase_callbacks.h:52-75
  virtual ~CollisionCallBackBase() {}
  virtual void init() {}
  virtual bool collide(CollisionObject* o1, CollisionObject* o2) = 0;
  bool operator()(CollisionObject* o1, CollisionObject* o2) { return collide(o1, o2); }
};

struct CollisionCallBackCollect : CollisionCallBackBase {  // default_broadphase_callbacks.h:200-230
  typedef std::pair<CollisionObject*, CollisionObject*> CollisionPair;
  explicit CollisionCallBackCollect(size_t max_size_) : max_size(max_size_) { collision_pairs.reserve(max_size_); }
  bool collide(CollisionObject* o1, CollisionObject* o2) override {
    collision_pairs.push_back(std::make_pair(o1, o2));
    return false;
  }
  void init() override { collision_pairs.clear(); }
  size_t numCollisionPairs() const { return collision_pairs.size(); }
  const std::vector<CollisionPair>& getCollisionPairs() const { return collision_pairs; }
  bool exist(const CollisionPair& p) const {
    for (const auto& q : collision_pairs)
      if ((q.first == p.first && q.second == p.second) || (q.first == p.second && q.second == p.first)) return true;
    return false;
  }

 protected:
  std::vector<CollisionPair> collision_pairs;
  size_t max_size;
};

struct DistanceCallBackBase {  // broadphase/broadphase_callbacks.h:77-103
  virtual ~DistanceCallBackBase() {}
  virtual void init() {}
  virtual bool distance(CollisionObject* o1, CollisionObject* o2, FCL_REAL& dist) = 0;
  bool operator()(CollisionObject* o1, CollisionObject* o2, FCL_REAL& dist) { return distance(o1, o2, dist); }
};

/// CollisionData / DistanceData and the default callbacks (broadphase/default_broadphase_callbacks.h:55-98,118,192;
/// src/broadphase/default_broadphase_callbacks.cpp:43-91).  Called directly, a default callback evaluates its pair with
/// collide() / distance(): one query = one batch of one, tens of microseconds of launch latency where the reference
/// spends two.  Handed to DynamicAABBTreeCollisionManager::collide / distance, the manager recognises the default
/// callbacks and evaluates the culled pairs in device batches (geometrically growing: 256, 1024, ... pairs), then folds
/// the records into the callback's CollisionData / DistanceData in the order, and with the stop rule, of the sequential
/// calls: same result object, a fraction of a microsecond per pair (tests/cpp/test_compat.cpp measures both).
struct CollisionData {
  CollisionData() : done(false) {}
  CollisionRequest request;
  CollisionResult result;
  bool done;  // the broadphase evaluation stops when set
  void clear() {
    result.clear();
    done = false;
  }
};
struct DistanceData {
  DistanceData() : done(false) {}
  DistanceRequest request;
  DistanceResult result;
  bool done;
  void clear() {
    result.clear();
    done = false;
  }
};
inline bool defaultCollisionFunction(CollisionObject* o1, CollisionObject* o2, void* data) {
  CollisionData* cd = static_cast<CollisionData*>(data);
  if (cd->done) return true;
  collide(o1, o2, cd->request, cd->result);
  if (cd->result.isCollision() && cd->result.numContacts() >= cd->request.num_max_contacts) cd->done = true;
  return cd->done;
}
inline bool defaultDistanceFunction(CollisionObject* o1, CollisionObject* o2, void* data, FCL_REAL& dist) {
  DistanceData* cd = static_cast<DistanceData*>(data);
  if (cd->done) {
    dist = cd->result.min_distance;
    return true;
  }
  distance(o1, o2, cd->request, cd->result);
  dist = cd->result.min_distance;
  if (dist <= 0) return true;  // in collision or in touch
  return cd->done;
}
struct CollisionCallBackDefault : CollisionCallBackBase {
  void init() override { data.clear(); }
  bool collide(CollisionObject* o1, CollisionObject* o2) override { return defaultCollisionFunction(o1, o2, &data); }
  CollisionData data;
};
struct DistanceCallBackDefault : DistanceCallBackBase {
  void init() override { data.clear(); }
  bool distance(CollisionObject* o1, CollisionObject* o2, FCL_REAL& dist) override { return defaultDistanceFunction(o1, o2, &data, dist); }
  DistanceData data;
};

/// Same calls as the reference manager (registerObject(s) / setup / update / collide(callback));
/// the candidate set is identical (all AABB-overlapping pairs), produced by the host pair-list
/// builder behind hfcl_broadphase_self_pairs instead of an incremental tree.
class DynamicAABBTreeCollisionManager {
 public:
  void registerObject(CollisionObject* obj) { objs_.push_back(obj); dirty_ = true; }
  void registerObjects(const std::vector<CollisionObject*>& v) { objs_.insert(objs_.end(), v.begin(), v.end()); dirty_ = true; }
  void unregisterObject(CollisionObject* obj) {
    objs_.erase(std::remove(objs_.begin(), objs_.end(), obj), objs_.end());
    dirty_ = true;
  }
  void clear() { objs_.clear(); dirty_ = true; }
  size_t size() const { return objs_.size(); }
  bool empty() const { return objs_.empty(); }
  void getObjects(std::vector<CollisionObject*>& out) const { out = objs_; }
  void setup() { refresh(); }
  void update() { dirty_ = true; refresh(); }

  /// self collision: callback on every pair with overlapping AABBs, until it returns true
  void collide(CollisionCallBackBase* callback) {
    callback->init();
    refresh();
    if (objs_.size() < 2) return;
    hfcl_pairlist* pl = hfcl_broadphase_self_pairs(aabbs_.data(), objs_.size(), 0);
    const uint32_t* p = hfcl_pairlist_data(pl);
    const size_t n = hfcl_pairlist_size(pl);
    struct Free { hfcl_pairlist* l; ~Free() { hfcl_pairlist_free(l); } } guard{pl};
    auto at = [&](size_t k) { return std::make_pair(objs_[p[2 * k]], objs_[p[2 * k + 1]]); };
    if (collide_default_batched(callback, n, at)) return;
    for (size_t k = 0; k < n; ++k)
      if ((*callback)(objs_[p[2 * k]], objs_[p[2 * k + 1]])) break;
  }
  /// self distance (broadphase_dynamic_AABB_tree.cpp:745-751): the callback sees the pairs whose AABBs are closer than the
  /// smallest distance reported so far (the pruning rule of distanceRecurse, :350-420), nearest boxes first, until it
  /// returns true.  The reference visits them in tree order; the minimum a callback accumulates is the same.
  void distance(DistanceCallBackBase* callback) {
    callback->init();
    refresh();
    const size_t n = objs_.size();
    std::vector<std::pair<FCL_REAL, std::pair<uint32_t, uint32_t>>> cand;
    cand.reserve(n * (n - 1) / 2);
    for (size_t i = 0; i < n; ++i)
      for (size_t j = i + 1; j < n; ++j) cand.push_back({aabb_distance(i, j), {uint32_t(i), uint32_t(j)}});
    std::sort(cand.begin(), cand.end());
    FCL_REAL min_dist = std::numeric_limits<FCL_REAL>::max();
    if (auto* def = dynamic_cast<DistanceCallBackDefault*>(callback)) {
      if (typeid(*callback) == typeid(DistanceCallBackDefault) && def->data.request.gjk_initial_guess != CachedGuess) {
        // the default callback: the candidates in device batches, folded in the sequential order with the sequential rules
        DistanceData& d = def->data;
        amd::BatchQueries& ctx = amd::default_context();
        size_t k = 0, batch = 256;
        while (k < cand.size() && !d.done && cand[k].first < min_dist) {
          size_t m = 0;  // candidates of this batch: those the sequential walk could still look at
          std::vector<std::pair<uint32_t, uint32_t>> ids;
          std::vector<Transform3f> tf1, tf2;
          while (k + m < cand.size() && m < batch && cand[k + m].first < min_dist) {
            CollisionObject *a = objs_[cand[k + m].second.first], *b = objs_[cand[k + m].second.second];
            ids.push_back({ctx.add(a->collisionGeometryPtr()), ctx.add(b->collisionGeometryPtr())});
            tf1.push_back(a->getTransform());
            tf2.push_back(b->getTransform());
            ++m;
          }
          ctx.run(ids, tf1, tf2, nullptr, &d.request);
          bool stop = false;
          for (size_t j = 0; j < m && !stop; ++j) {
            if (cand[k + j].first >= min_dist) { stop = true; break; }  // no closer pair can be left
            if (!(d.result.min_distance <= 0)) ctx.fill(d.result, ids[j], ctx.records()[j], ctx.guesses()[j]);  // distance(): isSatisfied
            min_dist = d.result.min_distance;
            if (min_dist <= 0) stop = true;  // in collision or in touch
          }
          if (stop) return;
          k += m;
          batch = std::min<size_t>(batch * 4, size_t(1) << 20);
        }
        return;
      }
    }
    for (const auto& c : cand) {
      if (c.first >= min_dist) break;  // no closer pair can be left
      if ((*callback)(objs_[c.second.first], objs_[c.second.second], min_dist)) break;
    }
  }
  /// against another manager (broadphase_dynamic_AABB_tree.cpp:734-743)
  void collide(DynamicAABBTreeCollisionManager* other, CollisionCallBackBase* callback) {
    callback->init();
    refresh();
    other->refresh();
    if (objs_.empty() || other->objs_.empty()) return;
    hfcl_pairlist* pl = hfcl_broadphase_pairs_between(aabbs_.data(), objs_.size(), other->aabbs_.data(), other->objs_.size(), 0);
    const uint32_t* p = hfcl_pairlist_data(pl);
    const size_t n = hfcl_pairlist_size(pl);
    struct Free { hfcl_pairlist* l; ~Free() { hfcl_pairlist_free(l); } } guard{pl};
    auto at = [&](size_t k) { return std::make_pair(objs_[p[2 * k]], other->objs_[p[2 * k + 1]]); };
    if (collide_default_batched(callback, n, at)) return;
    for (size_t k = 0; k < n; ++k)
      if ((*callback)(objs_[p[2 * k]], other->objs_[p[2 * k + 1]])) break;
  }

 private:
  /// The default collision callback over a list of culled pairs, evaluated in device batches.  Returns false (nothing
  /// done) for any other callback -- a user callback sees its pairs one by one -- and for requests that chain the GJK guess
  /// from pair to pair (CachedGuess: QueryRequest::updateGuess makes pair k depend on pair k - 1).
  /// Sequential semantics kept: records are folded in pair order by the same routine collide() uses, and nothing is
  /// folded once data.done is set (default_broadphase_callbacks.cpp:43-57); pairs of a batch beyond that point were
  /// evaluated for nothing, which is why the batches start small and grow.
  template <class At>
  bool collide_default_batched(CollisionCallBackBase* callback, size_t n, At at) {
    auto* def = dynamic_cast<CollisionCallBackDefault*>(callback);
    if (!def || typeid(*callback) != typeid(CollisionCallBackDefault)) return false;
    CollisionData& d = def->data;
    if (d.request.gjk_initial_guess == CachedGuess) return false;
    if (d.request.security_margin == -std::numeric_limits<FCL_REAL>::infinity() || d.request.num_max_contacts == 0) return false;
    amd::BatchQueries& ctx = amd::default_context();
    size_t k = 0, batch = 256;
    while (k < n && !d.done) {
      const size_t m = std::min(batch, n - k);
      std::vector<std::pair<uint32_t, uint32_t>> ids(m);
      std::vector<Transform3f> tf1(m), tf2(m);
      for (size_t j = 0; j < m; ++j) {
        const auto pr = at(k + j);
        ids[j] = {ctx.add(pr.first->collisionGeometryPtr()), ctx.add(pr.second->collisionGeometryPtr())};
        tf1[j] = pr.first->getTransform();
        tf2[j] = pr.second->getTransform();
      }
      ctx.run(ids, tf1, tf2, &d.request, nullptr);
      for (size_t j = 0; j < m && !d.done; ++j) {
        if (!(d.result.isCollision() && d.request.num_max_contacts <= d.result.numContacts()))  // collide()'s early return
          ctx.fill(d.result, ids[j], d.request, ctx.records()[j], ctx.guesses()[j]);
        if (d.result.isCollision() && d.result.numContacts() >= d.request.num_max_contacts) d.done = true;
      }
      k += m;
      batch = std::min<size_t>(batch * 4, size_t(1) << 20);
    }
    return true;
  }
  FCL_REAL aabb_distance(size_t i, size_t j) const {  // AABB::distance (BV/AABB.cpp:53-110): 0 when the boxes overlap
    FCL_REAL s = 0;
    for (int k = 0; k < 3; ++k) {
      const FCL_REAL lo = aabbs_[6 * i + k] - aabbs_[6 * j + 3 + k], hi = aabbs_[6 * j + k] - aabbs_[6 * i + 3 + k];
      const FCL_REAL d = std::max(FCL_REAL(0), std::max(lo, hi));
      s += d * d;
    }
    return std::sqrt(s);
  }
  void refresh() {  // CollisionObject::computeAABB for every object (collision_object.h:259-276)
    if (!dirty_ && aabbs_.size() == 6 * objs_.size()) return;
    std::vector<hfcl_shape> shapes(objs_.size());
    std::vector<double> verts;
    std::vector<uint32_t> ids(objs_.size());
    std::vector<double> tf(12 * objs_.size());
    for (size_t i = 0; i < objs_.size(); ++i) {
      const CollisionGeometry* g = objs_[i]->collisionGeometryPtr();
      hfcl_shape s{};
      s.type = g->getNodeType();
      const ShapeBase* sb = dynamic_cast<const ShapeBase*>(g);
      s.swept_sphere_radius = sb ? sb->getSweptSphereRadius() : 0.0;
      switch (g->getNodeType()) {
        case GEOM_BOX: { auto* b = static_cast<const Box*>(g); for (int k = 0; k < 3; ++k) s.params[k] = b->halfSide[k]; break; }
        case GEOM_SPHERE: s.params[0] = static_cast<const Sphere*>(g)->radius; break;
        case GEOM_CAPSULE: { auto* c = static_cast<const Capsule*>(g); s.params[0] = c->radius; s.params[1] = c->halfLength; break; }
        case GEOM_CONE: { auto* c = static_cast<const Cone*>(g); s.params[0] = c->radius; s.params[1] = c->halfLength; break; }
        case GEOM_CYLINDER: { auto* c = static_cast<const Cylinder*>(g); s.params[0] = c->radius; s.params[1] = c->halfLength; break; }
        case GEOM_ELLIPSOID: { auto* e = static_cast<const Ellipsoid*>(g); for (int k = 0; k < 3; ++k) s.params[k] = e->radii[k]; break; }
        case GEOM_CONVEX: {
          auto* c = static_cast<const ConvexBase*>(g);
          s.num_points = c->num_points;
          s.vertex_offset = static_cast<uint32_t>(verts.size() / 3);
          for (const Vec3f& q : *c->points) verts.insert(verts.end(), q.data(), q.data() + 3);
          break;
        }
        case BV_OBBRSS: {  // a mesh enters the broadphase as the point set of its vertices (BVHModelBase::computeLocalAABB, BVH_model.cpp:782-800)
          auto* m = static_cast<const BVHModel<OBBRSS>*>(g);
          s.type = GEOM_CONVEX;
          s.num_points = m->num_vertices;
          s.vertex_offset = static_cast<uint32_t>(verts.size() / 3);
          for (const Vec3f& q : m->vertices) verts.insert(verts.end(), q.data(), q.data() + 3);
          break;
        }
        default: throw std::invalid_argument("unsupported node type");
      }
      shapes[i] = s;
      ids[i] = static_cast<uint32_t>(i);
      std::memcpy(&tf[12 * i], &objs_[i]->getTransform(), 12 * sizeof(double));
    }
    aabbs_.assign(6 * objs_.size(), 0.0);
    const int rc = hfcl_world_aabbs(shapes.data(), shapes.size(), verts.data(), ids.data(), tf.data(), objs_.size(), aabbs_.data(), 0);
    if (rc) amd::throw_for(rc);
    for (size_t i = 0; i < objs_.size(); ++i) {
      AABB& a = objs_[i]->getAABB();
      a.min_ = Vec3f(aabbs_[6 * i], aabbs_[6 * i + 1], aabbs_[6 * i + 2]);
      a.max_ = Vec3f(aabbs_[6 * i + 3], aabbs_[6 * i + 4], aabbs_[6 * i + 5]);
    }
    dirty_ = false;
  }
  std::vector<CollisionObject*> objs_;
  std::vector<double> aabbs_;
  bool dirty_ = true;
};

namespace amd {
/// Evaluate the pairs a CollisionCallBackCollect holds in ONE device batch: results[i] belongs to
/// pairs[i] (each with its own CollisionResult, see SURVEY.md 8e on accumulation).
inline void collide(const std::vector<CollisionCallBackCollect::CollisionPair>& pairs, const CollisionRequest& request,
                    std::vector<CollisionResult>& results, BatchQueries& ctx = default_context()) {
  std::vector<std::pair<uint32_t, uint32_t>> ids(pairs.size());
  std::vector<Transform3f> tf1(pairs.size()), tf2(pairs.size());
  for (size_t i = 0; i < pairs.size(); ++i) {
    ids[i] = {ctx.add(pairs[i].first->collisionGeometryPtr()), ctx.add(pairs[i].second->collisionGeometryPtr())};
    tf1[i] = pairs[i].first->getTransform();
    tf2[i] = pairs[i].second->getTransform();
  }
  ctx.collide(ids, tf1, tf2, request, results);
}
}  // namespace amd

}  // namespace fcl
}  // namespace hpp
