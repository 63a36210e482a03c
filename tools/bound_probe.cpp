// experiment (CPU, the oracle's arithmetic): how many of the RSS tests of a mesh x mesh distance() walk a cheap lower bound decides.
// A pair of children is tested when their parent is expanded; a child whose lower bound is already >= min_distance THEN is pruned
// whatever its exact RSS distance is (min_distance only falls, and the reference visits a child only if its distance is < min_distance), so its
// rectDistance need not be computed.  Bounds tried: bounding spheres; the gap along the centre line; the larger of that and the gaps
// along the two rectangles' normals.
#include <cmath>
#include <cstdint>
#include <limits>
#include <thread>
#include <vector>
#include "bvh.hpp"
using namespace orc;
namespace {
struct Rs { V3 c; V3 a0, a1, a2; double h0, h1, r; };  // centre, axes (frame of model 1), half sides, radius
struct Walk {
  const MeshView &m1, &m2; M3 RT_R; V3 RT_T;
  double mind; unsigned long long nbv = 0, pruned = 0, by_sphere = 0, by_line = 0, by_three = 0, by_sat = 0;
  Walk(const MeshView& a, const Tf& t1, const MeshView& b, const Tf& t2) : m1(a), m2(b) { RT_R = tmul(t1.R, t2.R); RT_T = tmul(t1.R, t2.T - t1.T); mind = std::numeric_limits<double>::max(); }
  void leaf(int p1, int p2) { V3 S[3], T[3]; for (int k = 0; k < 3; ++k) { const double* p = m1.verts + 3 * size_t(m1.tris[3 * p1 + k]); const double* q = m2.verts + 3 * size_t(m2.tris[3 * p2 + k]); S[k] = V3(p[0], p[1], p[2]); T[k] = RT_R * V3(q[0], q[1], q[2]) + RT_T; } V3 P, Q; double d = std::sqrt(sqr_tri_distance(S, T, P, Q)); if (mind > d) mind = d; }
  static V3 col(const double* m, int k) { return V3(m[3 * k], m[3 * k + 1], m[3 * k + 2]); }
  Rs rs1(const hfcl_bvh_node& n) const { Rs s; s.a0 = col(n.rss_axes, 0); s.a1 = col(n.rss_axes, 1); s.a2 = col(n.rss_axes, 2); s.h0 = 0.5 * n.rss_length[0]; s.h1 = 0.5 * n.rss_length[1]; s.r = n.rss_radius; s.c = V3(n.rss_Tr[0], n.rss_Tr[1], n.rss_Tr[2]) + s.a0 * s.h0 + s.a1 * s.h1; return s; }
  Rs rs2(const hfcl_bvh_node& n) const { Rs s = rs1(n); s.c = RT_R * s.c + RT_T; s.a0 = RT_R * s.a0; s.a1 = RT_R * s.a1; s.a2 = RT_R * s.a2; return s; }
  static double gap(const Rs& A, const Rs& B, const V3& u) {  // separation of the two swept rectangles along the unit direction u
    const double ea = A.h0 * std::fabs(dot(A.a0, u)) + A.h1 * std::fabs(dot(A.a1, u)) + A.r;
    const double eb = B.h0 * std::fabs(dot(B.a0, u)) + B.h1 * std::fabs(dot(B.a1, u)) + B.r;
    return std::fabs(dot(B.c - A.c, u)) - ea - eb;
  }
  // the bound as the kernel forms it: in the frame of rectangle 1 (R, Tv of rss_distance), three axes
  double kernel_bound(const hfcl_bvh_node& b1, const hfcl_bvh_node& b2) const {
    M3 A1, A2;
    for (int r = 0; r < 3; ++r) for (int c = 0; c < 3; ++c) { A1.m[r][c] = b1.rss_axes[3 * c + r]; A2.m[r][c] = b2.rss_axes[3 * c + r]; }
    const M3 R = tmul(A1, RT_R * A2);
    const V3 Tt = RT_R * V3(b2.rss_Tr[0], b2.rss_Tr[1], b2.rss_Tr[2]) + RT_T - V3(b1.rss_Tr[0], b1.rss_Tr[1], b1.rss_Tr[2]);
    const V3 Tv = tmul(A1, Tt);
    const double h0 = 0.5 * b1.rss_length[0], h1 = 0.5 * b1.rss_length[1], g0 = 0.5 * b2.rss_length[0], g1 = 0.5 * b2.rss_length[1];
    const double cx = Tv.x + R.m[0][0] * g0 + R.m[0][1] * g1 - h0, cy = Tv.y + R.m[1][0] * g0 + R.m[1][1] * g1 - h1, cz = Tv.z + R.m[2][0] * g0 + R.m[2][1] * g1;
    const double gap1 = std::fabs(cz) - (g0 * std::fabs(R.m[2][0]) + g1 * std::fabs(R.m[2][1]));
    const double gap2 = std::fabs(cx * R.m[0][2] + cy * R.m[1][2] + cz * R.m[2][2]) - (h0 * std::fabs(R.m[0][2]) + h1 * std::fabs(R.m[1][2]));
    const double L = std::sqrt(cx * cx + cy * cy + cz * cz);
    const double e = h0 * std::fabs(cx) + h1 * std::fabs(cy) + g0 * std::fabs(R.m[0][0] * cx + R.m[1][0] * cy + R.m[2][0] * cz) + g1 * std::fabs(R.m[0][1] * cx + R.m[1][1] * cy + R.m[2][1] * cz);
    const double gap3 = L > 0 ? L - e / L : -1.0;
    const double lb = std::max(gap1, std::max(gap2, gap3)) - (b1.rss_radius + b2.rss_radius);
    return lb - 1e-9 * (L + h0 + h1 + g0 + g1 + b1.rss_radius + b2.rss_radius);
  }
  unsigned long long by_kernel = 0, violations = 0; double worst = 0;
  void bounds(const hfcl_bvh_node& n1, const hfcl_bvh_node& n2, double d) {
    {
      const double kb = kernel_bound(n1, n2);
      if (kb > d) { ++violations; worst = std::max(worst, kb - d); }
      if (d >= mind && kb > mind) ++by_kernel;
    }
    ++nbv;
    if (!(d >= mind)) return;  // (the exact test does not prune it now)
    ++pruned;
    const Rs A = rs1(n1), B = rs2(n2);
    const V3 dc = B.c - A.c;
    const double L = std::sqrt(dot(dc, dc));
    const double ls = L - std::sqrt(A.h0 * A.h0 + A.h1 * A.h1) - A.r - std::sqrt(B.h0 * B.h0 + B.h1 * B.h1) - B.r;
    if (ls >= mind) ++by_sphere;
    double ll = L > 0 ? gap(A, B, dc * (1.0 / L)) : -1;
    if (ll >= mind) ++by_line;
    double l3 = std::max(ll, std::max(gap(A, B, A.a2), gap(A, B, B.a2)));
    if (l3 >= mind) ++by_three;
    double l7 = l3;
    for (const V3& u : {A.a0, A.a1, B.a0, B.a1}) l7 = std::max(l7, gap(A, B, u));
    if (l7 >= mind) ++by_sat;
  }
  void rec(unsigned i, unsigned j) {
    const hfcl_bvh_node &n1 = m1.nodes[i], &n2 = m2.nodes[j]; bool l1 = n1.first_child < 0, l2 = n2.first_child < 0;
    if (l1 && l2) { leaf(-(n1.first_child + 1), -(n2.first_child + 1)); return; }
    double s1 = n1.obb_extent[0] * n1.obb_extent[0] + n1.obb_extent[1] * n1.obb_extent[1] + n1.obb_extent[2] * n1.obb_extent[2]; double s2 = n2.obb_extent[0] * n2.obb_extent[0] + n2.obb_extent[1] * n2.obb_extent[1] + n2.obb_extent[2] * n2.obb_extent[2];
    unsigned a1, a2, c1, c2; if (l2 || (!l1 && (s1 > s2))) { a1 = n1.first_child; a2 = j; c1 = a1 + 1; c2 = j; } else { a1 = i; a2 = n2.first_child; c1 = i; c2 = a2 + 1; }
    double d1 = rss_distance(RT_R, RT_T, m1.nodes[a1], m2.nodes[a2]), d2 = rss_distance(RT_R, RT_T, m1.nodes[c1], m2.nodes[c2]);
    bounds(m1.nodes[a1], m2.nodes[a2], d1); bounds(m1.nodes[c1], m2.nodes[c2], d2);
    if (d2 < d1) { if (!(d2 >= mind)) rec(c1, c2); if (!(d1 >= mind)) rec(a1, a2); } else { if (!(d1 >= mind)) rec(a1, a2); if (!(d2 >= mind)) rec(c1, c2); }
  }
};
}
// out[6]: tests, pruned at expansion time by the exact distance, of those: decided by the sphere bound, the centre-line gap, three axes, seven axes
extern "C" int bound_probe(const hfcl_bvh_node* nodes, const double* verts, const uint32_t* tris, const uint64_t* mt, size_t nm, const uint32_t* q1, const uint32_t* q2, const double* tf1, const double* tf2, size_t n, double* out, int nthreads) {
  std::vector<MeshView> ms(nm); for (size_t i = 0; i < nm; ++i) { ms[i].nodes = nodes + mt[4 * i]; ms[i].n_nodes = mt[4 * i + 1]; ms[i].verts = verts + 3 * mt[4 * i + 2]; ms[i].tris = tris + 3 * mt[4 * i + 3]; }
  std::vector<std::vector<double>> acc(nthreads, std::vector<double>(9, 0.0));
  std::vector<std::thread> th; size_t chunk = (n + nthreads - 1) / nthreads;
  for (int t = 0; t < nthreads; ++t) th.emplace_back([&, t] { for (size_t i = t * chunk; i < std::min(n, (t + 1) * chunk); ++i) {
    Walk w(ms[q1[i]], tf_from_abi(tf1 + 12 * i), ms[q2[i]], tf_from_abi(tf2 + 12 * i)); w.leaf(0, 0); w.rec(0, 0);
    acc[t][0] += w.nbv; acc[t][1] += w.pruned; acc[t][2] += w.by_sphere; acc[t][3] += w.by_line; acc[t][4] += w.by_three; acc[t][5] += w.by_sat; acc[t][6] += w.by_kernel; acc[t][7] += w.violations; acc[t][8] = std::max(acc[t][8], w.worst); } });
  for (auto& x : th) x.join();
  for (int k = 0; k < 9; ++k) { out[k] = 0; for (int t = 0; t < nthreads; ++t) out[k] += acc[t][k]; }
  return 0;
}
