// hfcl_broadphase.cpp -- host broadphase: posed-object AABBs and candidate-pair lists.
//
// north_star keeps the broadphase on the host; this is the pair-list producer that feeds
// hfcl_collide_batch / hfcl_distance_batch.  It reports exactly the pairs for which
// hpp-fcl's DynamicAABBTreeCollisionManager would invoke the collision callback: leaf pairs whose
// world AABBs overlap (/root/reference/src/broadphase/broadphase_dynamic_AABB_tree.cpp:252-293,
// include/hpp/fcl/BV/AABB.h:112-122), with world AABBs computed as CollisionObject::computeAABB does
// (include/hpp/fcl/collision_object.h:259-276) from the shapes' local AABBs
// (src/shape/geometric_shapes.cpp:145-254).
//
// Not a restatement of the reference's incremental tree (hierarchy_tree.hxx): the batch use-case
// rebuilds per frame, so the structure is a static median-split box tree over object centres,
// queried by all objects in parallel host threads; the reported SET equals the reference's, the
// order is (i ascending, j ascending) instead of the reference's tree-walk order.
#include <algorithm>
#include <cmath>
#include <cstdint>
#include <cstring>
#include <limits>
#include <thread>
#include <vector>

#include "../../include/hppfcl_amd.h"

namespace {

struct Box3 {
  double lo[3], hi[3];
};

inline bool boxes_touch(const double* a, const double* b) {  // AABB::overlap: closed intervals
  return !(a[0] > b[3] || a[1] > b[4] || a[2] > b[5] || a[3] < b[0] || a[4] < b[1] || a[5] < b[2]);
}

Box3 shape_local_box(const hfcl_shape& s, const double* verts) {
  Box3 b;
  auto symmetric = [&b](double hx, double hy, double hz) {
    const double h[3] = {hx, hy, hz};
    for (int k = 0; k < 3; ++k) {
      b.lo[k] = -h[k];
      b.hi[k] = h[k];
    }
  };
  switch (s.type) {
    case HFCL_GEOM_HALFSPACE:
    case HFCL_GEOM_PLANE: {
      // computeBV<AABB, Halfspace|Plane> in the shape's own frame (geometric_shapes_utility.cpp:391-455): the volume
      // is unbounded (+-DBL_MAX) except along a coordinate axis the normal is aligned with
      const double big = std::numeric_limits<double>::max();
      for (int k = 0; k < 3; ++k) {
        b.lo[k] = -big;
        b.hi[k] = big;
      }
      const double* n = s.params;
      const int axis = (n[1] == 0.0 && n[2] == 0.0) ? 0 : (n[0] == 0.0 && n[2] == 0.0) ? 1 : (n[0] == 0.0 && n[1] == 0.0) ? 2 : -1;
      if (axis >= 0 && n[axis] != 0.0) {
        const double v = n[axis] < 0 ? -s.params[3] : s.params[3];
        if (s.type == HFCL_GEOM_PLANE) b.lo[axis] = b.hi[axis] = v;
        else if (n[axis] < 0) b.lo[axis] = v;
        else b.hi[axis] = v;
      }
      break;
    }
    case HFCL_GEOM_BOX:
    case HFCL_GEOM_ELLIPSOID: symmetric(s.params[0], s.params[1], s.params[2]); break;
    case HFCL_GEOM_SPHERE: symmetric(s.params[0], s.params[0], s.params[0]); break;
    case HFCL_GEOM_CAPSULE: symmetric(s.params[0], s.params[0], s.params[1] + s.params[0]); break;
    case HFCL_GEOM_CONE:
    case HFCL_GEOM_CYLINDER: symmetric(std::abs(s.params[0]), std::abs(s.params[0]), std::abs(s.params[1])); break;
    default: {  // point sets: Convex, Triangle
      const double big = std::numeric_limits<double>::max();
      for (int k = 0; k < 3; ++k) {
        b.lo[k] = big;
        b.hi[k] = -big;
      }
      const double* p = verts + 3 * size_t(s.vertex_offset);
      for (uint32_t i = 0; i < s.num_points; ++i, p += 3)
        for (int k = 0; k < 3; ++k) {
          b.lo[k] = std::min(b.lo[k], p[k]);
          b.hi[k] = std::max(b.hi[k], p[k]);
        }
    }
  }
  if (s.swept_sphere_radius > 0)
    for (int k = 0; k < 3; ++k) {
      b.lo[k] -= s.swept_sphere_radius;
      b.hi[k] += s.swept_sphere_radius;
    }
  return b;
}

inline bool rotation_is_identity(const double* R) {  // Eigen::isIdentity with its default precision
  const double eps = 1e-12;
  for (int c = 0; c < 3; ++c)
    for (int r = 0; r < 3; ++r) {
      const double x = R[3 * c + r];
      if (r == c ? !(std::abs(x - 1.0) <= eps * std::min(std::abs(x), 1.0)) : !(std::abs(x) <= eps)) return false;
    }
  return true;
}

template <class F>
void parallel_ranges(size_t n, int n_threads, F f) {
  if (n_threads <= 1 || n < 4096) {
    f(0, size_t(0), n);
    return;
  }
  std::vector<std::thread> pool;
  const size_t chunk = (n + size_t(n_threads) - 1) / size_t(n_threads);
  for (int t = 0; t < n_threads; ++t) {
    const size_t b = std::min(n, size_t(t) * chunk), e = std::min(n, b + chunk);
    if (b < e) pool.emplace_back([=] { f(t, b, e); });
  }
  for (auto& th : pool) th.join();
}

int pick_threads(int n_threads) {
  if (n_threads > 0) return n_threads;
  return int(std::min<unsigned>(std::max(1u, std::thread::hardware_concurrency()), 32));
}

// twice the centre of [lo, hi]; unbounded boxes (Plane / Halfspace objects: +-DBL_MAX, +-inf after a rotation) sort as 0
inline double centre2(double lo, double hi) {
  const double c = lo + hi;
  return std::isfinite(c) ? c : 0.0;
}

// Static box tree over object centres (implicit layout: node k covers ids_[first, first+count)).
struct BoxTree {
  struct Node {
    double box[6];
    uint32_t first, count;
    int32_t left;  // children at left, left+1; -1 for leaves
  };
  static constexpr uint32_t LEAF = 8;
  const double* boxes;
  std::vector<uint32_t> ids;
  std::vector<Node> nodes;

  void build(const double* b, size_t n) {
    boxes = b;
    ids.resize(n);
    for (size_t i = 0; i < n; ++i) ids[i] = uint32_t(i);
    nodes.clear();
    nodes.reserve(2 * (n / (LEAF / 2) + 1));
    nodes.push_back(Node{});
    std::vector<uint32_t> todo{0};
    nodes[0].first = 0;
    nodes[0].count = uint32_t(n);
    while (!todo.empty()) {
      const uint32_t k = todo.back();
      todo.pop_back();
      Node nd = nodes[k];
      double clo[3], chi[3];
      for (int a = 0; a < 3; ++a) {
        nd.box[a] = clo[a] = std::numeric_limits<double>::max();
        nd.box[3 + a] = chi[a] = -std::numeric_limits<double>::max();
      }
      for (uint32_t i = nd.first; i < nd.first + nd.count; ++i) {
        const double* q = boxes + 6 * size_t(ids[i]);
        for (int a = 0; a < 3; ++a) {
          nd.box[a] = std::min(nd.box[a], q[a]);
          nd.box[3 + a] = std::max(nd.box[3 + a], q[3 + a]);
          const double c = centre2(q[a], q[3 + a]);
          clo[a] = std::min(clo[a], c);
          chi[a] = std::max(chi[a], c);
        }
      }
      nd.left = -1;
      if (nd.count > LEAF) {
        int ax = 0;
        if (chi[1] - clo[1] > chi[ax] - clo[ax]) ax = 1;
        if (chi[2] - clo[2] > chi[ax] - clo[ax]) ax = 2;
        const uint32_t half = nd.count / 2;
        uint32_t* base = ids.data() + nd.first;
        std::nth_element(base, base + half, base + nd.count, [&](uint32_t x, uint32_t y) {
          const double cx = centre2(boxes[6 * size_t(x) + ax], boxes[6 * size_t(x) + 3 + ax]);
          const double cy = centre2(boxes[6 * size_t(y) + ax], boxes[6 * size_t(y) + 3 + ax]);
          return cx < cy || (cx == cy && x < y);
        });
        nd.left = int32_t(nodes.size());
        Node l{}, r{};
        l.first = nd.first;
        l.count = half;
        r.first = nd.first + half;
        r.count = nd.count - half;
        nodes.push_back(l);
        nodes.push_back(r);
        todo.push_back(uint32_t(nd.left));
        todo.push_back(uint32_t(nd.left + 1));
      }
      nodes[k] = nd;
    }
  }

  // every id whose box touches `q` (closed test), appended to out
  void query(const double* q, std::vector<uint32_t>& out) const {
    uint32_t stack[128];
    int sp = 0;
    stack[sp++] = 0;
    while (sp) {
      const Node& nd = nodes[stack[--sp]];
      if (!boxes_touch(q, nd.box)) continue;
      if (nd.left < 0) {
        for (uint32_t i = nd.first; i < nd.first + nd.count; ++i)
          if (boxes_touch(q, boxes + 6 * size_t(ids[i]))) out.push_back(ids[i]);
      } else {
        stack[sp++] = uint32_t(nd.left);
        stack[sp++] = uint32_t(nd.left + 1);
      }
    }
  }
};

}  // namespace

struct hfcl_pairlist {
  std::vector<uint32_t> pairs;  // 2 per pair
};

extern "C" {

int hfcl_world_aabbs(const hfcl_shape* shapes, size_t n_shapes, const double* vertices, const uint32_t* object_shape,
                     const double* object_tf, size_t n_objects, double* aabbs_out, int n_threads) {
  if (!shapes || !object_shape || !object_tf || !aabbs_out) return HFCL_ERR_INVALID_ARGUMENT;
  std::vector<Box3> local(n_shapes);
  for (size_t s = 0; s < n_shapes; ++s) {
    const int t = shapes[s].type;
    if (t != HFCL_GEOM_BOX && t != HFCL_GEOM_SPHERE && t != HFCL_GEOM_CAPSULE && t != HFCL_GEOM_ELLIPSOID &&
        t != HFCL_GEOM_CONVEX && t != HFCL_GEOM_TRIANGLE && t != HFCL_GEOM_CONE && t != HFCL_GEOM_CYLINDER &&
        t != HFCL_GEOM_PLANE && t != HFCL_GEOM_HALFSPACE)
      return HFCL_ERR_UNSUPPORTED_PAIR;
    if ((t == HFCL_GEOM_CONVEX || t == HFCL_GEOM_TRIANGLE) && (!vertices || shapes[s].num_points == 0))
      return HFCL_ERR_INVALID_ARGUMENT;
    local[s] = shape_local_box(shapes[s], vertices);
  }
  for (size_t i = 0; i < n_objects; ++i)
    if (object_shape[i] >= n_shapes) return HFCL_ERR_INVALID_ARGUMENT;
  parallel_ranges(n_objects, pick_threads(n_threads), [&](int, size_t b, size_t e) {
    for (size_t i = b; i < e; ++i) {
      const Box3& L = local[object_shape[i]];
      const double* R = object_tf + 12 * i;
      const double* T = R + 9;
      double* o = aabbs_out + 6 * i;
      if (rotation_is_identity(R)) {
        for (int k = 0; k < 3; ++k) {
          o[k] = L.lo[k] + T[k];
          o[3 + k] = L.hi[k] + T[k];
        }
        continue;
      }
      for (int k = 0; k < 3; ++k) {  // interval arithmetic on row k of R
        double lo = 0, hi = 0;
        for (int j = 0; j < 3; ++j) {
          const double a = R[3 * j + k] * L.lo[j], c = R[3 * j + k] * L.hi[j];
          const double mn = c < a ? c : a, mx = c > a ? c : a;
          lo = j ? lo + mn : mn;
          hi = j ? hi + mx : mx;
        }
        o[k] = T[k] + lo;
        o[3 + k] = T[k] + hi;
      }
    }
  });
  return HFCL_OK;
}

hfcl_pairlist* hfcl_broadphase_self_pairs(const double* aabbs, size_t n_objects, int n_threads) {
  hfcl_pairlist* pl = new hfcl_pairlist;
  if (!aabbs || n_objects < 2) return pl;
  BoxTree tree;
  tree.build(aabbs, n_objects);
  const int nt = pick_threads(n_threads);
  std::vector<std::vector<uint32_t>> part(size_t(nt) + 1);
  parallel_ranges(n_objects, nt, [&](int t, size_t b, size_t e) {
    std::vector<uint32_t>& out = part[size_t(t)];
    std::vector<uint32_t> hits;
    for (size_t i = b; i < e; ++i) {
      hits.clear();
      tree.query(aabbs + 6 * i, hits);
      std::sort(hits.begin(), hits.end());
      for (uint32_t j : hits)
        if (j > i) {
          out.push_back(uint32_t(i));
          out.push_back(j);
        }
    }
  });
  size_t total = 0;
  for (auto& p : part) total += p.size();
  pl->pairs.reserve(total);
  for (auto& p : part) pl->pairs.insert(pl->pairs.end(), p.begin(), p.end());
  return pl;
}

// manager-vs-manager: pairs (i in A, j in B) with touching boxes
hfcl_pairlist* hfcl_broadphase_pairs_between(const double* aabbs_a, size_t n_a, const double* aabbs_b, size_t n_b,
                                             int n_threads) {
  hfcl_pairlist* pl = new hfcl_pairlist;
  if (!aabbs_a || !aabbs_b || !n_a || !n_b) return pl;
  BoxTree tree;
  tree.build(aabbs_b, n_b);
  const int nt = pick_threads(n_threads);
  std::vector<std::vector<uint32_t>> part(size_t(nt) + 1);
  parallel_ranges(n_a, nt, [&](int t, size_t b, size_t e) {
    std::vector<uint32_t>& out = part[size_t(t)];
    std::vector<uint32_t> hits;
    for (size_t i = b; i < e; ++i) {
      hits.clear();
      tree.query(aabbs_a + 6 * i, hits);
      std::sort(hits.begin(), hits.end());
      for (uint32_t j : hits) {
        out.push_back(uint32_t(i));
        out.push_back(j);
      }
    }
  });
  for (auto& p : part) pl->pairs.insert(pl->pairs.end(), p.begin(), p.end());
  return pl;
}

size_t hfcl_pairlist_size(const hfcl_pairlist* pl) { return pl ? pl->pairs.size() / 2 : 0; }
const uint32_t* hfcl_pairlist_data(const hfcl_pairlist* pl) { return pl ? pl->pairs.data() : nullptr; }
void hfcl_pairlist_free(hfcl_pairlist* pl) { delete pl; }

}  // extern "C"
