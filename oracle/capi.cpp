// ORACLE -- TEST INFRASTRUCTURE ONLY.  Not part of the product path (see vec3.hpp).
// C entry points (ctypes-friendly) over the fp64 CPU restatement.  Used by tests/, by
// __graft_entry__.smoke() as the checker, and by bench.py's cpu_baseline leg.
#include <cstring>
#include <algorithm>
#include <thread>
#include <vector>
#include "bvh.hpp"
#include "narrowphase.hpp"

using namespace orc;

// Test-only registry of ConvexBase::neighbors for large hulls (the C ABI of the product does not carry
// them: the device scans all vertices).  Keyed by vertex_offset of the hull.
#include <map>
struct NbrEntry {
  std::vector<uint32_t> off, ids;
};
static std::map<uint32_t, NbrEntry>& nbr_registry() {
  static std::map<uint32_t, NbrEntry> r;
  return r;
}
extern "C" void orc_clear_neighbors() { nbr_registry().clear(); }
extern "C" void orc_register_neighbors(uint32_t vertex_offset, const uint32_t* off, uint32_t n_points, const uint32_t* ids) {
  NbrEntry e;
  e.off.assign(off, off + n_points + 1);
  e.ids.assign(ids, ids + off[n_points]);
  nbr_registry()[vertex_offset] = std::move(e);
}

static Shape make_shape(const hfcl_shape& s, const double* vertices) {
  Shape r;
  r.kind = s.type;
  r.p[0] = s.params[0];
  r.p[1] = s.params[1];
  r.p[2] = s.params[2];
  r.p[3] = s.params[3];
  r.ssr = s.swept_sphere_radius;
  if (s.type == HFCL_GEOM_CONVEX || s.type == HFCL_GEOM_TRIANGLE) {
    r.verts = vertices + 3 * size_t(s.vertex_offset);
    r.nverts = int(s.num_points);
  }
  if (s.type == HFCL_GEOM_CONVEX && s.num_points >= 32) {
    auto it = nbr_registry().find(s.vertex_offset);
    if (it != nbr_registry().end() && it->second.off.size() == size_t(s.num_points) + 1) {
      r.nbr_off = it->second.off.data();
      r.nbr = it->second.ids.data();
      // buildSupportWarmStart (gjk.cpp:1470-1534): supports along +-e_i and the four cube diagonals,
      // hint and support data carried from one call to the next
      static const double dirs[14][3] = {{1, 0, 0}, {-1, 0, 0}, {0, 1, 0}, {0, -1, 0}, {0, 0, 1}, {0, 0, -1}, {1, 1, 1},
                                         {-1, -1, -1}, {-1, 1, 1}, {1, -1, -1}, {-1, -1, 1}, {1, 1, -1}, {1, -1, 1}, {-1, 1, -1}};
      SupportData sd;
      int hint = 0;
      r.n_warm = 0;  // grows while it is being built: each call already sees the earlier entries
      for (int k = 0; k < 14; ++k) {
        const V3 d(dirs[k][0], dirs[k][1], dirs[k][2]);
        const V3 sp = shape_support(r, d, hint, &sd);
        r.warm_pts[k][0] = sp.x; r.warm_pts[k][1] = sp.y; r.warm_pts[k][2] = sp.z;
        r.warm_idx[k] = hint;
        r.n_warm = k + 1;
      }
    }
  }
  return r;
}

template <class F>
static void parallel_for(size_t n, int n_threads, F f) {
  if (n_threads <= 1 || n < 2) {
    f(size_t(0), n);
    return;
  }
  std::vector<std::thread> th;
  size_t chunk = (n + n_threads - 1) / n_threads;
  for (int t = 0; t < n_threads; ++t) {
    size_t b = t * chunk, e = std::min(n, b + chunk);
    if (b >= e) break;
    th.emplace_back([=] { f(b, e); });
  }
  for (auto& t : th) t.join();
}

extern "C" {

void orc_distance_request_init(hfcl_distance_request* r) { distance_request_defaults(r); }
void orc_collision_request_init(hfcl_collision_request* r) { collision_request_defaults(r); }

int orc_distance_batch(const hfcl_shape* shapes, size_t n_shapes, const double* vertices, const uint32_t* s1,
                       const uint32_t* s2, const double* tf1, const double* tf2, size_t n,
                       const hfcl_distance_request* req, hfcl_result* out, const hfcl_guess* gin, hfcl_guess* gout,
                       int n_threads) {
  std::vector<Shape> lib(n_shapes);
  for (size_t i = 0; i < n_shapes; ++i) lib[i] = make_shape(shapes[i], vertices);
  int err = 0;
  parallel_for(n, n_threads, [&](size_t b, size_t e) {
    for (size_t i = b; i < e; ++i) {
      int rc = distance_pair(lib[s1[i]], tf_from_abi(tf1 + 12 * i), lib[s2[i]], tf_from_abi(tf2 + 12 * i), *req,
                             gin ? gin + i : nullptr, out[i], gout ? gout + i : nullptr);
      if (rc) err = rc;
    }
  });
  return err;
}

int orc_collide_batch(const hfcl_shape* shapes, size_t n_shapes, const double* vertices, const uint32_t* s1,
                      const uint32_t* s2, const double* tf1, const double* tf2, size_t n,
                      const hfcl_collision_request* req, hfcl_result* out, const hfcl_guess* gin, hfcl_guess* gout,
                      int n_threads) {
  std::vector<Shape> lib(n_shapes);
  for (size_t i = 0; i < n_shapes; ++i) lib[i] = make_shape(shapes[i], vertices);
  int err = 0;
  parallel_for(n, n_threads, [&](size_t b, size_t e) {
    for (size_t i = b; i < e; ++i) {
      int rc = collide_pair(lib[s1[i]], tf_from_abi(tf1 + 12 * i), lib[s2[i]], tf_from_abi(tf2 + 12 * i), *req,
                            gin ? gin + i : nullptr, out[i], gout ? gout + i : nullptr);
      if (rc) err = rc;
    }
  });
  return err;
}

// Raw GJK (+ optional EPA) on a MinkowskiDiff, the level test/gjk.cpp:337-490 exercises.
//   out[0..2]=w0, [3..5]=w1, [6..8]=normal (shape-0 frame), [9]=gjk.distance, [10..12]=ray,
//   [13]=epa.depth ; istat[0]=gjk status, [1]=gjk iterations, [2]=epa status (-1 n/a), [3]=epa iterations,
//   [4]=final simplex rank, [5]=iterations_momentum_stop
int orc_gjk_raw(const hfcl_shape* s0, const double* v0, const hfcl_shape* s1, const double* v1, const double* tf0,
                const double* tf1, unsigned gjk_max_it, double gjk_tol, int variant, int criterion, int criterion_type,
                double distance_upper_bound, const double* guess, int run_epa, unsigned epa_max_it, double epa_tol,
                const double* epa_guess, double* out, int* istat) {
  Shape a = make_shape(*s0, v0), b = make_shape(*s1, v1);
  if (a.verts) a.verts = v0 + 3 * size_t(s0->vertex_offset);
  if (b.verts) b.verts = v1 + 3 * size_t(s1->vertex_offset);
  MinkowskiDiff md;
  md.set(&a, &b, tf_from_abi(tf0), tf_from_abi(tf1));
  GJK gjk(gjk_max_it, gjk_tol);
  gjk.gjk_variant = variant;
  gjk.convergence_criterion = criterion;
  gjk.convergence_criterion_type = criterion_type;
  gjk.distance_upper_bound = distance_upper_bound;
  int hint[2] = {0, 0};
  GJK::Status st = gjk.evaluate(md, V3(guess[0], guess[1], guess[2]), hint);
  istat[0] = st;
  istat[1] = int(gjk.iterations);
  istat[2] = -1;
  istat[3] = 0;
  istat[4] = gjk.simplex.rank;
  istat[5] = int(gjk.iterations_momentum_stop);
  V3 w0, w1, n;
  out[13] = 0;
  bool use_epa = run_epa && st == GJK::Collision;
  if (!use_epa) {
    gjk.get_witness_points_and_normal(md, w0, w1, n);
  } else {
    EPA epa(epa_max_it, epa_tol);
    EPA::Status es = epa.evaluate(gjk, V3(epa_guess[0], epa_guess[1], epa_guess[2]));
    istat[2] = es;
    istat[3] = int(epa.iterations);
    epa.get_witness_points_and_normal(md, w0, w1, n);
    out[13] = epa.depth;
  }
  for (int k = 0; k < 3; ++k) {
    out[k] = w0[k];
    out[3 + k] = w1[k];
    out[6 + k] = n[k];
    out[10 + k] = gjk.ray[k];
  }
  out[9] = gjk.distance;
  return 0;
}

// Project::project{Line,Triangle,Tetrahedra}Origin (test/simple.cpp KATs). pts: rank x 3.
// out: param[4], sqr_distance ; returns encode.
unsigned orc_project_origin(int rank, const double* pts, double* out) {
  V3 p[4];
  for (int i = 0; i < rank; ++i) p[i] = V3(pts[3 * i], pts[3 * i + 1], pts[3 * i + 2]);
  ProjectResult r;
  if (rank == 2) r = project_line_origin(p[0], p[1]);
  if (rank == 3) r = project_triangle_origin(p[0], p[1], p[2]);
  if (rank == 4) r = project_tetrahedra_origin(p[0], p[1], p[2], p[3]);
  for (int i = 0; i < 4; ++i) out[i] = r.param[i];
  out[4] = r.sqr_distance;
  return r.encode;
}

// Support function of one shape (shape frame), for supports KATs.
void orc_shape_support(const hfcl_shape* s, const double* verts, const double* dir, double* out, int* hint) {
  Shape a = make_shape(*s, verts);
  V3 r = shape_support(a, V3(dir[0], dir[1], dir[2]), *hint);
  out[0] = r.x;
  out[1] = r.y;
  out[2] = r.z;
}

// BVHModel<OBBRSS> x BVHModel<OBBRSS> collide().  Mesh table: per mesh (node_off, n_nodes, vert_off, tri_off)
// into the concatenated node / vertex (xyz doubles) / triangle (3 x uint32, local vertex ids) arrays.
// out_stats (nullable): 2 x uint32 per pair (num_bv_tests, num_leaf_tests).
// contacts (nullable): capacity max_contacts; *n_contacts receives the number stored.
int orc_bvh_collide_batch(const hfcl_bvh_node* nodes, const double* verts, const uint32_t* tris,
                          const uint64_t* mesh_table, size_t n_meshes, const uint32_t* m1, const uint32_t* m2,
                          const double* tf1, const double* tf2, size_t n, const hfcl_collision_request* req,
                          hfcl_result* out, uint32_t* out_stats, hfcl_contact* contacts, size_t max_contacts,
                          size_t* n_contacts, int n_threads) {
  std::vector<MeshView> meshes(n_meshes);
  for (size_t i = 0; i < n_meshes; ++i) {
    meshes[i].nodes = nodes + mesh_table[4 * i];
    meshes[i].n_nodes = mesh_table[4 * i + 1];
    meshes[i].verts = verts + 3 * mesh_table[4 * i + 2];
    meshes[i].tris = tris + 3 * mesh_table[4 * i + 3];
  }
  int err = 0;
  std::vector<std::vector<hfcl_contact>> per_thread(std::max(1, n_threads));
  size_t chunk = (n + std::max(1, n_threads) - 1) / std::max(1, n_threads);
  parallel_for(n, n_threads, [&](size_t b, size_t e) {
    std::vector<hfcl_contact>& cl = per_thread[chunk ? b / chunk : 0];
    for (size_t i = b; i < e; ++i) {
      BvhStats st;
      int rc = bvh_collide_pair(meshes[m1[i]], tf_from_abi(tf1 + 12 * i), meshes[m2[i]], tf_from_abi(tf2 + 12 * i), *req,
                                out[i], contacts ? &cl : nullptr, uint32_t(i), &st);
      if (rc) err = rc;
      if (out_stats) {
        out_stats[2 * i] = st.num_bv_tests;
        out_stats[2 * i + 1] = st.num_leaf_tests;
      }
    }
  });
  if (contacts) {
    size_t k = 0;
    for (auto& cl : per_thread)
      for (auto& c : cl)
        if (k < max_contacts) contacts[k++] = c;
    if (n_contacts) *n_contacts = k;
  }
  return err;
}

// collide() over one shape table that may hold BVHModel<OBBRSS> entries (type HFCL_BV_OBBRSS, bvh_index into
// the mesh table) next to convex shapes: mesh x mesh, mesh x shape, shape x mesh (operand swap of
// src/collision.cpp:93-108) and shape x shape, dispatched like the reference's collision matrix.
// out_stats (nullable): 2 x uint32 per pair -- num_bv_tests, num_leaf_tests of the traversal node
// (traversal_node_bvhs.h:126-128 / traversal_node_bvh_shape.h: the counters the reference keeps), 0 for shape x shape pairs.
int orc_mixed_collide_batch_stats(const hfcl_shape* shapes, size_t n_shapes, const double* shape_verts, const hfcl_bvh_node* nodes,
                                  const double* mesh_verts, const uint32_t* tris, const uint64_t* mesh_table, size_t n_meshes,
                                  const uint32_t* s1, const uint32_t* s2, const double* tf1, const double* tf2, size_t n,
                                  const hfcl_collision_request* req, hfcl_result* out, hfcl_guess* guess_out,
                                  hfcl_contact* contacts, size_t max_contacts, size_t* n_contacts, int n_threads,
                                  uint32_t* out_stats) {
  std::vector<MeshView> meshes(n_meshes);
  for (size_t i = 0; i < n_meshes; ++i) {
    meshes[i].nodes = nodes + mesh_table[4 * i];
    meshes[i].n_nodes = mesh_table[4 * i + 1];
    meshes[i].verts = mesh_verts + 3 * mesh_table[4 * i + 2];
    meshes[i].tris = tris + 3 * mesh_table[4 * i + 3];
  }
  std::vector<Shape> lib(n_shapes);
  for (size_t i = 0; i < n_shapes; ++i) lib[i] = make_shape(shapes[i], shape_verts);
  int err = 0;
  std::vector<std::vector<hfcl_contact>> per_thread(std::max(1, n_threads));
  size_t chunk = (n + std::max(1, n_threads) - 1) / std::max(1, n_threads);
  parallel_for(n, n_threads, [&](size_t b, size_t e) {
    std::vector<hfcl_contact>& cl = per_thread[chunk ? b / chunk : 0];
    for (size_t i = b; i < e; ++i) {
      const hfcl_shape &a = shapes[s1[i]], &c = shapes[s2[i]];
      const bool ma = a.type == HFCL_BV_OBBRSS, mc = c.type == HFCL_BV_OBBRSS;
      const Tf t1 = tf_from_abi(tf1 + 12 * i), t2 = tf_from_abi(tf2 + 12 * i);
      hfcl_guess* go = guess_out ? guess_out + i : nullptr;
      int rc;
      BvhStats st;
      if (ma && mc) {
        rc = bvh_collide_pair(meshes[a.bvh_index], t1, meshes[c.bvh_index], t2, *req, out[i], contacts ? &cl : nullptr,
                              uint32_t(i), &st);
        if (go) *go = hfcl_guess{{req->q.cached_gjk_guess[0], req->q.cached_gjk_guess[1], req->q.cached_gjk_guess[2]},
                                 {req->q.cached_support_func_guess[0], req->q.cached_support_func_guess[1]}};
      } else if (ma) {
        rc = bvh_shape_collide_pair(meshes[a.bvh_index], t1, lib[s2[i]], t2, *req, false, out[i], contacts ? &cl : nullptr,
                                    uint32_t(i), go, &st);
      } else if (mc) {
        rc = bvh_shape_collide_pair(meshes[c.bvh_index], t2, lib[s1[i]], t1, *req, true, out[i], contacts ? &cl : nullptr,
                                    uint32_t(i), go, &st);
      } else {
        rc = collide_pair(lib[s1[i]], t1, lib[s2[i]], t2, *req, nullptr, out[i], go);
      }
      if (out_stats) {
        out_stats[2 * i] = st.num_bv_tests;
        out_stats[2 * i + 1] = st.num_leaf_tests;
      }
      if (rc) err = rc;
    }
  });
  if (contacts) {
    size_t k = 0;
    for (auto& cl : per_thread)
      for (auto& c : cl)
        if (k < max_contacts) contacts[k++] = c;
    if (n_contacts) *n_contacts = k;
  }
  return err;
}

int orc_mixed_collide_batch(const hfcl_shape* shapes, size_t n_shapes, const double* shape_verts, const hfcl_bvh_node* nodes,
                            const double* mesh_verts, const uint32_t* tris, const uint64_t* mesh_table, size_t n_meshes,
                            const uint32_t* s1, const uint32_t* s2, const double* tf1, const double* tf2, size_t n,
                            const hfcl_collision_request* req, hfcl_result* out, hfcl_guess* guess_out,
                            hfcl_contact* contacts, size_t max_contacts, size_t* n_contacts, int n_threads) {
  return orc_mixed_collide_batch_stats(shapes, n_shapes, shape_verts, nodes, mesh_verts, tris, mesh_table, n_meshes, s1, s2, tf1, tf2, n,
                                       req, out, guess_out, contacts, max_contacts, n_contacts, n_threads, nullptr);
}

// distance() counterpart of orc_mixed_collide_batch.
int orc_mixed_distance_batch(const hfcl_shape* shapes, size_t n_shapes, const double* shape_verts, const hfcl_bvh_node* nodes,
                             const double* mesh_verts, const uint32_t* tris, const uint64_t* mesh_table, size_t n_meshes,
                             const uint32_t* s1, const uint32_t* s2, const double* tf1, const double* tf2, size_t n,
                             const hfcl_distance_request* req, hfcl_result* out, hfcl_guess* guess_out, int n_threads) {
  std::vector<MeshView> meshes(n_meshes);
  for (size_t i = 0; i < n_meshes; ++i) {
    meshes[i].nodes = nodes + mesh_table[4 * i];
    meshes[i].n_nodes = mesh_table[4 * i + 1];
    meshes[i].verts = mesh_verts + 3 * mesh_table[4 * i + 2];
    meshes[i].tris = tris + 3 * mesh_table[4 * i + 3];
  }
  std::vector<Shape> lib(n_shapes);
  for (size_t i = 0; i < n_shapes; ++i) lib[i] = make_shape(shapes[i], shape_verts);
  int err = 0;
  parallel_for(n, n_threads, [&](size_t b, size_t e) {
    for (size_t i = b; i < e; ++i) {
      const hfcl_shape &a = shapes[s1[i]], &c = shapes[s2[i]];
      const bool ma = a.type == HFCL_BV_OBBRSS, mc = c.type == HFCL_BV_OBBRSS;
      const Tf t1 = tf_from_abi(tf1 + 12 * i), t2 = tf_from_abi(tf2 + 12 * i);
      hfcl_guess* go = guess_out ? guess_out + i : nullptr;
      int rc;
      if (ma && mc)
        rc = bvh_distance_pair(meshes[a.bvh_index], t1, meshes[c.bvh_index], t2, out[i], nullptr);
      else if (ma)
        rc = bvh_shape_distance_pair(meshes[a.bvh_index], t1, lib[s2[i]], t2, *req, false, out[i], go);
      else if (mc)
        rc = bvh_shape_distance_pair(meshes[c.bvh_index], t2, lib[s1[i]], t1, *req, true, out[i], go);
      else
        rc = distance_pair(lib[s1[i]], t1, lib[s2[i]], t2, *req, nullptr, out[i], go);
      if (rc) err = rc;
    }
  });
  return err;
}

// distance of ONE triangle of a (mesh, solid) query of orc_mixed_distance_batch, as the traversal's leaf computes it
double orc_mixed_leaf_distance(const hfcl_shape* shapes, size_t n_shapes, const double* shape_verts, const hfcl_bvh_node* nodes,
                               const double* mesh_verts, const uint32_t* tris, const uint64_t* mesh_table, size_t n_meshes, uint32_t s1,
                               uint32_t s2, const double* tf1, const double* tf2, const hfcl_distance_request* req, int pid) {
  const hfcl_shape &a = shapes[s1], &c = shapes[s2];
  const bool ma = a.type == HFCL_BV_OBBRSS;
  const hfcl_shape& msh = ma ? a : c;
  MeshView mv;
  mv.nodes = nodes + mesh_table[4 * size_t(msh.bvh_index)];
  mv.n_nodes = mesh_table[4 * size_t(msh.bvh_index) + 1];
  mv.verts = mesh_verts + 3 * mesh_table[4 * size_t(msh.bvh_index) + 2];
  mv.tris = tris + 3 * mesh_table[4 * size_t(msh.bvh_index) + 3];
  (void)n_shapes;
  (void)n_meshes;
  const Shape solid = make_shape(ma ? c : a, shape_verts);
  const Tf t1 = tf_from_abi(tf1), t2 = tf_from_abi(tf2);
  return ma ? bvh_shape_leaf_distance(mv, t1, solid, t2, *req, pid) : bvh_shape_leaf_distance(mv, t2, solid, t1, *req, pid);
}

// (diagnostic) the walk of ONE (mesh, solid) distance() query, event by event (bvh_shape_distance_trace)
size_t orc_mixed_distance_trace(const hfcl_shape* shapes, const double* shape_verts, const hfcl_bvh_node* nodes, const double* mesh_verts,
                                const uint32_t* tris, const uint64_t* mesh_table, uint32_t s1, uint32_t s2, const double* tf1, const double* tf2,
                                const hfcl_distance_request* req, double* out, size_t cap) {
  const hfcl_shape &a = shapes[s1], &c = shapes[s2];
  const bool ma = a.type == HFCL_BV_OBBRSS;
  const hfcl_shape& msh = ma ? a : c;
  MeshView mv;
  mv.nodes = nodes + mesh_table[4 * size_t(msh.bvh_index)];
  mv.n_nodes = mesh_table[4 * size_t(msh.bvh_index) + 1];
  mv.verts = mesh_verts + 3 * mesh_table[4 * size_t(msh.bvh_index) + 2];
  mv.tris = tris + 3 * mesh_table[4 * size_t(msh.bvh_index) + 3];
  const Shape solid = make_shape(ma ? c : a, shape_verts);
  const Tf t1 = tf_from_abi(tf1), t2 = tf_from_abi(tf2);
  return ma ? bvh_shape_distance_trace(mv, t1, solid, t2, *req, out, cap) : bvh_shape_distance_trace(mv, t2, solid, t1, *req, out, cap);
}

// BVHModel<OBBRSS> x BVHModel<OBBRSS> distance(); same mesh table as orc_bvh_collide_batch.
int orc_bvh_distance_batch(const hfcl_bvh_node* nodes, const double* verts, const uint32_t* tris,
                           const uint64_t* mesh_table, size_t n_meshes, const uint32_t* m1, const uint32_t* m2,
                           const double* tf1, const double* tf2, size_t n, hfcl_result* out, uint32_t* out_stats,
                           int n_threads) {
  std::vector<MeshView> meshes(n_meshes);
  for (size_t i = 0; i < n_meshes; ++i) {
    meshes[i].nodes = nodes + mesh_table[4 * i];
    meshes[i].n_nodes = mesh_table[4 * i + 1];
    meshes[i].verts = verts + 3 * mesh_table[4 * i + 2];
    meshes[i].tris = tris + 3 * mesh_table[4 * i + 3];
  }
  parallel_for(n, n_threads, [&](size_t b, size_t e) {
    for (size_t i = b; i < e; ++i) {
      BvhStats st;
      bvh_distance_pair(meshes[m1[i]], tf_from_abi(tf1 + 12 * i), meshes[m2[i]], tf_from_abi(tf2 + 12 * i), out[i], &st);
      if (out_stats) {
        out_stats[2 * i] = st.num_bv_tests;
        out_stats[2 * i + 1] = st.num_leaf_tests;
      }
    }
  });
  return 0;
}

// distance of ONE triangle pair of a mesh x mesh query, as the traversal's leaf computes it (model-1 frame)
double orc_bvh_leaf_distance(const hfcl_bvh_node* nodes, const double* verts, const uint32_t* tris, const uint64_t* mesh_table,
                             size_t n_meshes, uint32_t m1, uint32_t m2, const double* tf1, const double* tf2, int pid1, int pid2) {
  std::vector<MeshView> meshes(n_meshes);
  for (size_t i = 0; i < n_meshes; ++i) {
    meshes[i].nodes = nodes + mesh_table[4 * i];
    meshes[i].n_nodes = mesh_table[4 * i + 1];
    meshes[i].verts = verts + 3 * mesh_table[4 * i + 2];
    meshes[i].tris = tris + 3 * mesh_table[4 * i + 3];
  }
  return bvh_leaf_distance(meshes[m1], tf_from_abi(tf1), meshes[m2], tf_from_abi(tf2), pid1, pid2);
}

// rectDistance on raw inputs (unit tests): Rab row-major 9, Tab 3, a 2, b 2
double orc_rect_distance(const double* Rab, const double* Tab, const double* a, const double* b) {
  M3 R;
  for (int i = 0; i < 3; ++i)
    for (int j = 0; j < 3; ++j) R.m[i][j] = Rab[3 * i + j];
  return rect_distance(R, V3(Tab[0], Tab[1], Tab[2]), a, b);
}
// sqrTriDistance on raw inputs: S, T = 3x3 each; out = P(3), Q(3); returns d^2
double orc_sqr_tri_distance(const double* S, const double* T, double* out) {
  V3 s[3], t[3], P, Q;
  for (int k = 0; k < 3; ++k) {
    s[k] = V3(S[3 * k], S[3 * k + 1], S[3 * k + 2]);
    t[k] = V3(T[3 * k], T[3 * k + 1], T[3 * k + 2]);
  }
  const double d2 = sqr_tri_distance(s, t, P, Q);
  for (int k = 0; k < 3; ++k) {
    out[k] = P[k];
    out[3 + k] = Q[k];
  }
  return d2;
}

}  // extern "C"
