// ORACLE -- TEST INFRASTRUCTURE ONLY.  Not part of the product path (see vec3.hpp).
//
// What the reference's broadphase hands to the narrow phase, restated without any tree:
//   local AABBs     Shape::computeLocalAABB  src/shape/geometric_shapes.cpp:145-254 via
//                   computeBV<AABB,S>(s, Identity)  src/shape/geometric_shapes_utility.cpp:293-388
//   world AABB      CollisionObject::computeAABB  include/hpp/fcl/collision_object.h:259-276
//   candidate set   DynamicAABBTreeCollisionManager::collide(callback) calls the callback exactly for
//                   the leaf pairs whose AABBs overlap (leafCollide, src/broadphase/
//                   broadphase_dynamic_AABB_tree.cpp:252-293; AABB::overlap BV/AABB.h:112-122);
//                   the SET of pairs is tree-independent, so the oracle is the O(n^2) double loop.
// PARITY UNPINNED by reference numbers: test/broadphase*.cpp compare managers against each other
// (and against brute force) on random scenes, which is what tests/test_broadphase.py does.
#include <cmath>
#include <cstring>
#include <limits>
#include "../include/hppfcl_amd.h"

namespace {

// Plane / Halfspace: computeBV<AABB, Halfspace|Plane> with tf = identity (geometric_shapes_utility.cpp:391-455):
// unbounded (+-DBL_MAX) except along an axis the normal is aligned with.
bool flat_local_aabb(const hfcl_shape& s, double mn[3], double mx[3]) {
  if (s.type != HFCL_GEOM_HALFSPACE && s.type != HFCL_GEOM_PLANE) return false;
  const double big = std::numeric_limits<double>::max();
  const double* n = s.params;
  const double d = s.params[3];
  for (int k = 0; k < 3; ++k) {
    mn[k] = -big;
    mx[k] = big;
  }
  int axis = -1;
  if (n[1] == 0.0 && n[2] == 0.0) axis = 0;
  else if (n[0] == 0.0 && n[2] == 0.0) axis = 1;
  else if (n[0] == 0.0 && n[1] == 0.0) axis = 2;
  if (axis >= 0) {
    if (s.type == HFCL_GEOM_HALFSPACE) {
      if (n[axis] < 0) mn[axis] = -d;
      else if (n[axis] > 0) mx[axis] = d;
    } else {
      if (n[axis] < 0) mn[axis] = mx[axis] = -d;
      else if (n[axis] > 0) mn[axis] = mx[axis] = d;
    }
  }
  if (s.swept_sphere_radius > 0)  // geometric_shapes.cpp:222-243
    for (int k = 0; k < 3; ++k) {
      mn[k] -= s.swept_sphere_radius;
      mx[k] += s.swept_sphere_radius;
    }
  return true;
}

void local_aabb(const hfcl_shape& s, const double* verts, double mn[3], double mx[3]) {
  if (flat_local_aabb(s, mn, mx)) return;
  double d[3] = {0, 0, 0};
  bool have = true;
  switch (s.type) {
    case HFCL_GEOM_BOX: d[0] = s.params[0]; d[1] = s.params[1]; d[2] = s.params[2]; break;
    case HFCL_GEOM_SPHERE: d[0] = d[1] = d[2] = s.params[0]; break;
    case HFCL_GEOM_ELLIPSOID: d[0] = s.params[0]; d[1] = s.params[1]; d[2] = s.params[2]; break;
    case HFCL_GEOM_CAPSULE:  // |R.col(2)| * halfLength + radius with R = I
      d[0] = 0 * s.params[1] + s.params[0];
      d[1] = 0 * s.params[1] + s.params[0];
      d[2] = 1 * s.params[1] + s.params[0];
      break;
    case HFCL_GEOM_CONE:      // :334-366 with R = I: (|r| + 0 + 0, 0 + |r| + 0, 0 + 0 + |halfLength|)
    case HFCL_GEOM_CYLINDER:
      d[0] = std::fabs(1 * s.params[0]) + std::fabs(0 * s.params[0]) + std::fabs(0 * s.params[1]);
      d[1] = std::fabs(0 * s.params[0]) + std::fabs(1 * s.params[0]) + std::fabs(0 * s.params[1]);
      d[2] = std::fabs(0 * s.params[0]) + std::fabs(0 * s.params[0]) + std::fabs(1 * s.params[1]);
      break;
    default: have = false;
  }
  if (have) {
    for (int k = 0; k < 3; ++k) {
      mx[k] = 0 + d[k];
      mn[k] = 0 - d[k];
    }
  } else {  // ConvexBase / TriangleP: min / max over the points
    const double big = std::numeric_limits<double>::max();
    for (int k = 0; k < 3; ++k) {
      mn[k] = big;
      mx[k] = -big;
    }
    const double* p = verts + 3 * size_t(s.vertex_offset);
    for (uint32_t i = 0; i < s.num_points; ++i)
      for (int k = 0; k < 3; ++k) {
        if (p[3 * i + k] < mn[k]) mn[k] = p[3 * i + k];
        if (p[3 * i + k] > mx[k]) mx[k] = p[3 * i + k];
      }
  }
  if (s.swept_sphere_radius > 0)
    for (int k = 0; k < 3; ++k) {
      mn[k] -= s.swept_sphere_radius;
      mx[k] += s.swept_sphere_radius;
    }
}

bool is_identity(const double* R) {  // Eigen isIdentity(prec = 1e-12) on a column-major 3x3
  const double prec = 1e-12;
  for (int c = 0; c < 3; ++c)
    for (int r = 0; r < 3; ++r) {
      const double x = R[3 * c + r];
      if (r == c) {
        if (!(std::abs(x - 1.0) <= prec * std::fmin(std::abs(x), 1.0))) return false;
      } else if (!(std::abs(x) <= prec))
        return false;
    }
  return true;
}

}  // namespace

extern "C" int orc_world_aabbs(const hfcl_shape* shapes, const double* verts, const uint32_t* obj_shape,
                               const double* obj_tf, size_t n, double* out /* n x 6: min, max */) {
  for (size_t i = 0; i < n; ++i) {
    double lmn[3], lmx[3];
    local_aabb(shapes[obj_shape[i]], verts, lmn, lmx);
    const double* R = obj_tf + 12 * i;  // column-major
    const double* T = R + 9;
    double* o = out + 6 * i;
    if (is_identity(R)) {
      for (int k = 0; k < 3; ++k) {
        o[k] = lmn[k] + T[k];
        o[3 + k] = lmx[k] + T[k];
      }
      continue;
    }
    for (int k = 0; k < 3; ++k) {
      double a[3], b[3];
      for (int j = 0; j < 3; ++j) {
        a[j] = R[3 * j + k] * lmn[j];  // row k
        b[j] = R[3 * j + k] * lmx[j];
      }
      o[k] = T[k] + ((std::fmin(a[0], b[0]) + std::fmin(a[1], b[1])) + std::fmin(a[2], b[2]));
      o[3 + k] = T[k] + ((std::fmax(a[0], b[0]) + std::fmax(a[1], b[1])) + std::fmax(a[2], b[2]));
    }
  }
  return 0;
}

// all i < j with overlapping boxes; returns the count, stores up to cap pairs (i ascending, j ascending)
extern "C" size_t orc_bruteforce_pairs(const double* aabbs, size_t n, uint32_t* pairs, size_t cap) {
  size_t cnt = 0;
  for (size_t i = 0; i < n; ++i)
    for (size_t j = i + 1; j < n; ++j) {
      const double *a = aabbs + 6 * i, *b = aabbs + 6 * j;
      if (a[0] > b[3] || a[1] > b[4] || a[2] > b[5] || a[3] < b[0] || a[4] < b[1] || a[5] < b[2]) continue;
      if (cnt < cap) {
        pairs[2 * cnt] = uint32_t(i);
        pairs[2 * cnt + 1] = uint32_t(j);
      }
      ++cnt;
    }
  return cnt;
}
