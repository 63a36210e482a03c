// hfcl_bvh_build.cpp -- host construction of BVHModel<OBBRSS> node arrays (C-ABI hfcl_bvh_build).
//
// Produces the arrays hpp-fcl's BVHModel<OBBRSS>::endModel() produces for a triangle model with the
// default SPLIT_METHOD_MEAN (/root/reference/src/BVH/BVH_model.cpp:508-576,858-960): per node the
// OBB (covariance of the triangle corners -> cyclic Jacobi -> axes ordered max/mid/cross ->
// min/max projections; BV_fitter.cpp:50-76,501-531, BVH_utility.cpp:183-259,529-575,
// internal/tools.h:103-202) and the RSS on the same axes (slab radius + rectangle shrunk by the
// cap radius + corner growth; BVH_utility.cpp:264-482); split = mean of the corner projections on
// OBB axis 0, centroid test, stable left partition (BV_splitter.cpp:81-118,276-279).
//
// Host code by design: the reference builds its trees on the host too; the device
// consumes the arrays through hfcl_lib_add_bvh.  Not a device fallback.
//
// Organisation (differs from the reference's recursion): node ids are computed in closed form --
// a subtree of n triangles owns 2n-1 consecutive ids after its root's sibling pair -- so subtrees
// are independent jobs and large models are built by several host threads; the corner projections
// of a node are staged once in a per-thread buffer and shared by the OBB and RSS fits.
#include <algorithm>
#include <cmath>
#include <cstdint>
#include <cstring>
#include <limits>
#include <thread>
#include <vector>

#include "../../include/hppfcl_amd.h"

namespace {

struct Sym3 {
  double m[3][3];
};

// Cyclic Jacobi eigen-solver with the reference's thresholds and rotation order.
bool jacobi_eigen(const Sym3& in, double eval[3], double evec[3][3]) {
  double A[3][3];
  std::memcpy(A, in.m, sizeof(A));
  double acc[3] = {0, 0, 0}, base[3], d[3];
  double V[3][3] = {{1, 0, 0}, {0, 1, 0}, {0, 0, 1}};
  for (int k = 0; k < 3; ++k) base[k] = d[k] = A[k][k];
  for (int sweep = 0; sweep < 50; ++sweep) {
    const double off = std::abs(A[0][1]) + std::abs(A[0][2]) + std::abs(A[1][2]);
    if (off == 0.0) {
      std::memcpy(eval, d, sizeof(d));
      std::memcpy(evec, V, sizeof(V));
      return true;
    }
    const double thresh = sweep < 3 ? 0.2 * off / 9 : 0.0;
    static const int PQ[3][2] = {{0, 1}, {0, 2}, {1, 2}};
    for (const auto& pq : PQ) {
      const int p = pq[0], q = pq[1];
      const double apq = A[p][q];
      const double g = 100.0 * std::abs(apq);
      if (sweep > 3 && std::abs(d[p]) + g == std::abs(d[p]) && std::abs(d[q]) + g == std::abs(d[q])) {
        A[p][q] = 0.0;
        continue;
      }
      if (!(std::abs(apq) > thresh)) continue;
      double h = d[q] - d[p], t;
      if (std::abs(h) + g == std::abs(h)) {
        t = apq / h;
      } else {
        const double theta = 0.5 * h / apq;
        t = 1.0 / (std::abs(theta) + std::sqrt(1.0 + theta * theta));
        if (theta < 0.0) t = -t;
      }
      const double c = 1.0 / std::sqrt(1 + t * t), s = t * c, tau = s / (1.0 + c);
      h = t * apq;
      acc[p] -= h;
      acc[q] += h;
      d[p] -= h;
      d[q] += h;
      A[p][q] = 0.0;
      auto givens = [s, tau](double& x, double& y) {
        const double x0 = x, y0 = y;
        x = x0 - s * (y0 + x0 * tau);
        y = y0 + s * (x0 - y0 * tau);
      };
      // the one remaining off-diagonal pair of a 3x3: index r != p, q
      const int r = 3 - p - q;
      if (r < p)
        givens(A[r][p], A[r][q]);
      else if (r < q)
        givens(A[p][r], A[r][q]);
      else
        givens(A[p][r], A[q][r]);
      for (int j = 0; j < 3; ++j) givens(V[j][p], V[j][q]);
    }
    for (int k = 0; k < 3; ++k) {
      base[k] += acc[k];
      d[k] = base[k];
      acc[k] = 0.0;
    }
  }
  return false;
}

struct Job {
  int32_t node;
  uint32_t first, count;
};

class TreeBuilder {
 public:
  TreeBuilder(const double* v, const uint32_t* t, hfcl_bvh_node* nodes, uint32_t* prim)
      : v_(v), t_(t), nodes_(nodes), prim_(prim) {}

  // Builds the subtree rooted at `root` (its own id already assigned) serially.
  void build_subtree(const Job& root, std::vector<double>& proj) {
    std::vector<Job> todo{root};
    while (!todo.empty()) {
      const Job j = todo.back();
      todo.pop_back();
      Job kids[2];
      if (process(j, proj, kids)) {
        todo.push_back(kids[1]);
        todo.push_back(kids[0]);
      }
    }
  }

  // One node: fit, split rule, partition.  Returns true and the two child jobs for inner nodes.
  bool process(const Job& j, std::vector<double>& proj, Job kids[2]) {
    hfcl_bvh_node& nd = nodes_[j.node];
    uint32_t* ids = prim_ + j.first;
    const uint32_t n = j.count;
    fit(ids, n, nd, proj);
    nd.first_primitive = int32_t(j.first);
    nd.num_primitives = int32_t(n);
    nd._pad = 0;
    if (n == 1) {
      nd.first_child = -(int32_t(ids[0]) + 1);
      return false;
    }
    const double ax0[3] = {nd.obb_axes[0], nd.obb_axes[1], nd.obb_axes[2]};
    double sum[3] = {0, 0, 0};
    for (uint32_t i = 0; i < n; ++i) {
      const double *a = corner(ids[i], 0), *b = corner(ids[i], 1), *c = corner(ids[i], 2);
      for (int k = 0; k < 3; ++k) sum[k] += (a[k] + b[k]) + c[k];
    }
    const double split = (sum[0] * ax0[0] + sum[1] * ax0[1] + sum[2] * ax0[2]) / (3 * n);
    uint32_t left = 0;
    for (uint32_t i = 0; i < n; ++i) {
      const double *a = corner(ids[i], 0), *b = corner(ids[i], 1), *c = corner(ids[i], 2);
      const double gx = ((a[0] + b[0]) + c[0]) / 3., gy = ((a[1] + b[1]) + c[1]) / 3., gz = ((a[2] + b[2]) + c[2]) / 3.;
      if (!(ax0[0] * gx + ax0[1] * gy + ax0[2] * gz > split)) std::swap(ids[i], ids[left++]);
    }
    if (left == 0 || left == n) left = n / 2;
    return make_children(j, nd, left, kids);
  }

  // id bookkeeping: next_id_[node] = id of the child pair `node` allocates (the reference's running
  // num_bvs at the moment its recursion reaches `node`)
  bool make_children(const Job& j, hfcl_bvh_node& nd, uint32_t left, Job kids[2]) {
    const int32_t pair = next_id_[size_t(j.node)];
    nd.first_child = pair;
    kids[0] = Job{pair, j.first, left};
    kids[1] = Job{pair + 1, j.first + left, j.count - left};
    // the left subtree owns 2*left-2 ids after the pair, then the right subtree's
    next_id_[size_t(pair)] = pair + 2;
    next_id_[size_t(pair) + 1] = pair + 2 + (2 * int32_t(left) - 2);
    return true;
  }

  void init_ids(size_t n_nodes) {
    next_id_.assign(n_nodes, 0);
    next_id_[0] = 1;
  }

 private:
  const double* corner(uint32_t tri, int k) const { return v_ + 3 * size_t(t_[3 * size_t(tri) + k]); }

  void fit(const uint32_t* ids, uint32_t n, hfcl_bvh_node& nd, std::vector<double>& proj) const {
    // second moments of the 3n corners, accumulated triangle by triangle
    double s1[3] = {0, 0, 0}, sxx = 0, syy = 0, szz = 0, sxy = 0, sxz = 0, syz = 0;
    for (uint32_t i = 0; i < n; ++i) {
      const double *a = corner(ids[i], 0), *b = corner(ids[i], 1), *c = corner(ids[i], 2);
      s1[0] += (a[0] + b[0] + c[0]);
      s1[1] += (a[1] + b[1] + c[1]);
      s1[2] += (a[2] + b[2] + c[2]);
      sxx += (a[0] * a[0] + b[0] * b[0] + c[0] * c[0]);
      syy += (a[1] * a[1] + b[1] * b[1] + c[1] * c[1]);
      szz += (a[2] * a[2] + b[2] * b[2] + c[2] * c[2]);
      sxy += (a[0] * a[1] + b[0] * b[1] + c[0] * c[1]);
      sxz += (a[0] * a[2] + b[0] * b[2] + c[0] * c[2]);
      syz += (a[1] * a[2] + b[1] * b[2] + c[1] * c[2]);
    }
    const unsigned np = 3 * n;
    Sym3 C;
    C.m[0][0] = sxx - s1[0] * s1[0] / np;
    C.m[1][1] = syy - s1[1] * s1[1] / np;
    C.m[2][2] = szz - s1[2] * s1[2] / np;
    C.m[0][1] = C.m[1][0] = sxy - s1[0] * s1[1] / np;
    C.m[1][2] = C.m[2][1] = syz - s1[1] * s1[2] / np;
    C.m[0][2] = C.m[2][0] = sxz - s1[0] * s1[2] / np;
    double ev[3], E[3][3];
    if (!jacobi_eigen(C, ev, E)) {
      std::memset(ev, 0, sizeof(ev));
      std::memset(E, 0, sizeof(E));
    }
    // order: largest, middle; third axis = their cross product
    int lo = ev[0] > ev[1] ? 1 : 0, hi = 1 - lo, md;
    if (ev[2] < ev[lo]) {
      md = lo;
      lo = 2;
    } else if (ev[2] > ev[hi]) {
      md = hi;
      hi = 2;
    } else {
      md = 2;
    }
    (void)lo;
    double U[3][3];  // U[axis][component]
    for (int r = 0; r < 3; ++r) {
      U[0][r] = E[r][hi];
      U[1][r] = E[r][md];
    }
    U[2][0] = E[1][hi] * E[2][md] - E[1][md] * E[2][hi];
    U[2][1] = E[0][md] * E[2][hi] - E[0][hi] * E[2][md];
    U[2][2] = E[0][hi] * E[1][md] - E[0][md] * E[1][hi];
    for (int c = 0; c < 3; ++c)
      for (int r = 0; r < 3; ++r) nd.obb_axes[3 * c + r] = nd.rss_axes[3 * c + r] = U[c][r];

    // corner coordinates in the node frame, SoA (x[], y[], z[]) -- shared by both fits
    const size_t m = size_t(3) * n;
    if (proj.size() < 3 * m) proj.resize(3 * m);
    double *X = proj.data(), *Y = X + m, *Z = Y + m;
    const double big = std::numeric_limits<double>::max();
    double lo3[3] = {big, big, big}, hi3[3] = {-big, -big, -big};
    for (uint32_t i = 0; i < n; ++i)
      for (int k = 0; k < 3; ++k) {
        const double* p = corner(ids[i], k);
        const double q[3] = {U[0][0] * p[0] + U[0][1] * p[1] + U[0][2] * p[2], U[1][0] * p[0] + U[1][1] * p[1] + U[1][2] * p[2],
                             U[2][0] * p[0] + U[2][1] * p[1] + U[2][2] * p[2]};
        const size_t s = size_t(3) * i + k;
        X[s] = q[0];
        Y[s] = q[1];
        Z[s] = q[2];
        for (int a = 0; a < 3; ++a) {
          if (q[a] > hi3[a]) hi3[a] = q[a];
          if (q[a] < lo3[a]) lo3[a] = q[a];
        }
      }
    double mid[3];
    for (int a = 0; a < 3; ++a) {
      mid[a] = (hi3[a] + lo3[a]) / 2;
      nd.obb_extent[a] = (hi3[a] - lo3[a]) / 2;
    }
    for (int r = 0; r < 3; ++r) nd.obb_To[r] = U[0][r] * mid[0] + U[1][r] * mid[1] + U[2][r] * mid[2];

    // RSS: slab along axis 2 gives the radius; rectangle sides from the cap-corrected extremes
    double zlo = Z[0], zhi = Z[0];
    for (size_t s = 1; s < m; ++s) {
      if (Z[s] < zlo)
        zlo = Z[s];
      else if (Z[s] > zhi)
        zhi = Z[s];
    }
    const double rad = 0.5 * (zhi - zlo), rad2 = rad * rad, zc = 0.5 * (zhi + zlo);
    auto cap = [&](size_t s) {  // half-chord of the cap sphere at the corner's height
      const double dz = Z[s] - zc;
      return std::sqrt(std::max<double>(rad2 - dz * dz, 0));
    };
    double rlo[2], rhi[2];
    for (int a = 0; a < 2; ++a) {
      const double* W = a == 0 ? X : Y;
      size_t ilo = 0, ihi = 0;
      double wlo = W[0], whi = W[0];
      for (size_t s = 1; s < m; ++s) {
        if (W[s] < wlo) {
          ilo = s;
          wlo = W[s];
        } else if (W[s] > whi) {
          ihi = s;
          whi = W[s];
        }
      }
      double lo_a = W[ilo] + cap(ilo), hi_a = W[ihi] - cap(ihi);
      for (size_t s = 0; s < m; ++s) {
        if (W[s] < lo_a) {
          const double w = W[s] + cap(s);
          if (w < lo_a) lo_a = w;
        } else if (W[s] > hi_a) {
          const double w = W[s] - cap(s);
          if (w > hi_a) hi_a = w;
        }
      }
      rlo[a] = lo_a;
      rhi[a] = hi_a;
    }
    // corners: a point outside both side ranges may still poke out of the rounded corner
    const double h = std::sqrt(0.5);
    for (size_t s = 0; s < m; ++s) {
      const int sx = X[s] > rhi[0] ? 1 : (X[s] < rlo[0] ? -1 : 0);
      if (!sx) continue;
      const int sy = Y[s] > rhi[1] ? 1 : (Y[s] < rlo[1] ? -1 : 0);
      if (!sy) continue;
      const double dx = X[s] - (sx > 0 ? rhi[0] : rlo[0]), dy = Y[s] - (sy > 0 ? rhi[1] : rlo[1]);
      // diagonal coordinate u along (sx, sy)/sqrt2; the four reference branches differ only in these signs
      double u;
      if (sx > 0 && sy > 0)
        u = dx * h + dy * h;
      else if (sx > 0)
        u = dx * h - dy * h;
      else if (sy > 0)
        u = dy * h - dx * h;
      else
        u = -dx * h - dy * h;
      const double ex = (sx > 0 ? h * u : -h * u) - dx, ey = (sy > 0 ? h * u : -h * u) - dy;
      const double t = ex * ex + ey * ey + (zc - Z[s]) * (zc - Z[s]);
      u = u - std::sqrt(std::max<double>(rad2 - t, 0));
      if (u > 0) {
        if (sx > 0)
          rhi[0] += u * h;
        else
          rlo[0] -= u * h;
        if (sy > 0)
          rhi[1] += u * h;
        else
          rlo[1] -= u * h;
      }
    }
    for (int r = 0; r < 3; ++r) nd.rss_Tr[r] = U[0][r] * rlo[0] + U[1][r] * rlo[1] + U[2][r] * zc;
    nd.rss_length[0] = std::max<double>(rhi[0] - rlo[0], 0);
    nd.rss_length[1] = std::max<double>(rhi[1] - rlo[1], 0);
    nd.rss_radius = rad;
  }

  const double* v_;
  const uint32_t* t_;
  hfcl_bvh_node* nodes_;
  uint32_t* prim_;
  std::vector<int32_t> next_id_;  // per node: the id its first child pair will get
};

}  // namespace

extern "C" int hfcl_bvh_build(const double* vertices, size_t n_vertices, const uint32_t* triangles, size_t n_tris,
                              hfcl_bvh_node* nodes_out, uint32_t* primitive_indices_out, int n_threads) {
  if (!vertices || !triangles || !nodes_out || !primitive_indices_out || !n_vertices || !n_tris)
    return HFCL_ERR_INVALID_ARGUMENT;
  if (n_tris > size_t(1) << 30) return HFCL_ERR_INVALID_ARGUMENT;  // first_child is an int32 (BV_node.h:57)
  for (size_t i = 0; i < 3 * n_tris; ++i)
    if (triangles[i] >= n_vertices) return HFCL_ERR_INVALID_ARGUMENT;
  for (size_t i = 0; i < n_tris; ++i) primitive_indices_out[i] = uint32_t(i);
  TreeBuilder tb(vertices, triangles, nodes_out, primitive_indices_out);
  tb.init_ids(2 * n_tris - 1);
  if (n_threads <= 0) n_threads = int(std::min<unsigned>(std::max(1u, std::thread::hardware_concurrency()), 16));
  // frontier of independent subtrees: split the biggest job until there are enough of them
  std::vector<Job> frontier{Job{0, 0, uint32_t(n_tris)}};
  std::vector<double> scratch;
  const size_t want = n_tris < 20000 || n_threads == 1 ? 1 : size_t(4 * n_threads);
  while (frontier.size() < want) {
    size_t big = 0;
    for (size_t i = 1; i < frontier.size(); ++i)
      if (frontier[i].count > frontier[big].count) big = i;
    if (frontier[big].count < 2048) break;
    const Job j = frontier[big];
    Job kids[2];
    frontier.erase(frontier.begin() + long(big));
    if (tb.process(j, scratch, kids)) {
      frontier.push_back(kids[0]);
      frontier.push_back(kids[1]);
    }
  }
  if (frontier.size() == 1 || n_threads == 1) {
    for (const Job& j : frontier) tb.build_subtree(j, scratch);
    return HFCL_OK;
  }
  std::sort(frontier.begin(), frontier.end(), [](const Job& a, const Job& b) { return a.count > b.count; });
  std::vector<std::thread> pool;
  const size_t nt = size_t(n_threads);
  std::vector<std::vector<Job>> per_thread(nt);
  std::vector<size_t> load(nt, 0);
  for (const Job& j : frontier) {  // longest-processing-time-first assignment
    const size_t t = size_t(std::min_element(load.begin(), load.end()) - load.begin());
    per_thread[t].push_back(j);
    load[t] += j.count;
  }
  for (int t = 0; t < n_threads; ++t)
    pool.emplace_back([&tb, &per_thread, t] {
      std::vector<double> local;
      for (const Job& j : per_thread[size_t(t)]) tb.build_subtree(j, local);
    });
  for (auto& th : pool) th.join();
  return HFCL_OK;
}
