// ORACLE -- TEST INFRASTRUCTURE ONLY.  Not part of the product path (see vec3.hpp).
//
// fp64 CPU restatement of the BVHModel<OBBRSS> construction (triangles model, SPLIT_METHOD_MEAN):
//   BVHModel::buildTree / recursiveBuildTree   src/BVH/BVH_model.cpp:858-960
//   BVFitter<OBBRSS>::fit                      src/BVH/BV_fitter.cpp:501-531 (axisFromEigen :50-76)
//   getCovariance                              src/BVH/BVH_utility.cpp:183-259
//   eigen (cyclic Jacobi, <= 50 sweeps)        include/hpp/fcl/internal/tools.h:103-202
//   getExtentAndCenter (mesh)                  src/BVH/BVH_utility.cpp:529-575
//   getRadiusAndOriginAndRectangleSize         src/BVH/BVH_utility.cpp:264-482
//   BVSplitter<OBBRSS> mean rule / apply       src/BVH/BV_splitter.cpp:81-118,241-279
// PARITY UNPINNED by reference tests: no reference test fixes node values; checked by structural
// properties (tests/test_bvh_build.py) and by code review against the cited lines.
#include <algorithm>
#include <cmath>
#include <cstring>
#include <limits>
#include <vector>
#include "../include/hppfcl_amd.h"

namespace {

struct Builder {
  const double* vs;
  const uint32_t* ts;
  hfcl_bvh_node* nodes;
  uint32_t* prim;
  unsigned num_bvs;

  const double* vert(uint32_t tri, int j) const { return vs + 3 * size_t(ts[3 * size_t(tri) + j]); }

  static void jacobi(const double M[3][3], double dout[3], double vout[3][3]) {  // tools.h:103-202
    double R[3][3];
    std::memcpy(R, M, sizeof(R));
    const int n = 3;
    double b[3], z[3], d[3], v[3][3] = {{1, 0, 0}, {0, 1, 0}, {0, 0, 1}};
    for (int ip = 0; ip < n; ++ip) {
      b[ip] = d[ip] = R[ip][ip];
      z[ip] = 0;
    }
    for (int i = 0; i < 50; ++i) {
      double sm = 0;
      for (int ip = 0; ip < n; ++ip)
        for (int iq = ip + 1; iq < n; ++iq) sm += std::abs(R[ip][iq]);
      if (sm == 0.0) {
        for (int a = 0; a < 3; ++a) {
          dout[a] = d[a];
          for (int c = 0; c < 3; ++c) vout[a][c] = v[a][c];
        }
        return;
      }
      const double tresh = i < 3 ? 0.2 * sm / (n * n) : 0.0;
      for (int ip = 0; ip < n; ++ip) {
        for (int iq = ip + 1; iq < n; ++iq) {
          double g = 100.0 * std::abs(R[ip][iq]);
          if (i > 3 && std::abs(d[ip]) + g == std::abs(d[ip]) && std::abs(d[iq]) + g == std::abs(d[iq]))
            R[ip][iq] = 0.0;
          else if (std::abs(R[ip][iq]) > tresh) {
            double h = d[iq] - d[ip], t;
            if (std::abs(h) + g == std::abs(h))
              t = R[ip][iq] / h;
            else {
              const double theta = 0.5 * h / R[ip][iq];
              t = 1.0 / (std::abs(theta) + std::sqrt(1.0 + theta * theta));
              if (theta < 0.0) t = -t;
            }
            const double c = 1.0 / std::sqrt(1 + t * t);
            const double s = t * c;
            const double tau = s / (1.0 + c);
            h = t * R[ip][iq];
            z[ip] -= h;
            z[iq] += h;
            d[ip] -= h;
            d[iq] += h;
            R[ip][iq] = 0.0;
            auto rot = [&](double& x, double& y) {
              const double gg = x, hh = y;
              x = gg - s * (hh + gg * tau);
              y = hh + s * (gg - hh * tau);
            };
            for (int j = 0; j < ip; ++j) rot(R[j][ip], R[j][iq]);
            for (int j = ip + 1; j < iq; ++j) rot(R[ip][j], R[j][iq]);
            for (int j = iq + 1; j < n; ++j) rot(R[ip][j], R[iq][j]);
            for (int j = 0; j < n; ++j) rot(v[j][ip], v[j][iq]);
          }
        }
      }
      for (int ip = 0; ip < n; ++ip) {
        b[ip] += z[ip];
        d[ip] = b[ip];
        z[ip] = 0.0;
      }
    }
    // "too many iterations": the reference returns with dout/vout unwritten (tools.h:199-201)
    for (int a = 0; a < 3; ++a) {
      dout[a] = 0;
      for (int c = 0; c < 3; ++c) vout[a][c] = 0;
    }
  }

  void fit(const uint32_t* idx, unsigned n, hfcl_bvh_node& nd) const {
    // getCovariance
    double S1[3] = {0, 0, 0}, S2[3][3] = {{0, 0, 0}, {0, 0, 0}, {0, 0, 0}};
    for (unsigned i = 0; i < n; ++i) {
      const double *p1 = vert(idx[i], 0), *p2 = vert(idx[i], 1), *p3 = vert(idx[i], 2);
      S1[0] += (p1[0] + p2[0] + p3[0]);
      S1[1] += (p1[1] + p2[1] + p3[1]);
      S1[2] += (p1[2] + p2[2] + p3[2]);
      S2[0][0] += (p1[0] * p1[0] + p2[0] * p2[0] + p3[0] * p3[0]);
      S2[1][1] += (p1[1] * p1[1] + p2[1] * p2[1] + p3[1] * p3[1]);
      S2[2][2] += (p1[2] * p1[2] + p2[2] * p2[2] + p3[2] * p3[2]);
      S2[0][1] += (p1[0] * p1[1] + p2[0] * p2[1] + p3[0] * p3[1]);
      S2[0][2] += (p1[0] * p1[2] + p2[0] * p2[2] + p3[0] * p3[2]);
      S2[1][2] += (p1[1] * p1[2] + p2[1] * p2[2] + p3[1] * p3[2]);
    }
    const unsigned n_points = 3 * n;
    double M[3][3];
    M[0][0] = S2[0][0] - S1[0] * S1[0] / n_points;
    M[1][1] = S2[1][1] - S1[1] * S1[1] / n_points;
    M[2][2] = S2[2][2] - S1[2] * S1[2] / n_points;
    M[0][1] = S2[0][1] - S1[0] * S1[1] / n_points;
    M[1][2] = S2[1][2] - S1[1] * S1[2] / n_points;
    M[0][2] = S2[0][2] - S1[0] * S1[2] / n_points;
    M[1][0] = M[0][1];
    M[2][0] = M[0][2];
    M[2][1] = M[1][2];
    double s[3], E[3][3];
    jacobi(M, s, E);
    // axisFromEigen
    int mn, mid, mx;
    if (s[0] > s[1]) {
      mx = 0;
      mn = 1;
    } else {
      mn = 0;
      mx = 1;
    }
    if (s[2] < s[mn]) {
      mid = mn;
      mn = 2;
    } else if (s[2] > s[mx]) {
      mid = mx;
      mx = 2;
    } else {
      mid = 2;
    }
    double ax[3][3];  // ax[c] = column c
    for (int r = 0; r < 3; ++r) {
      ax[0][r] = E[r][mx];
      ax[1][r] = E[r][mid];
    }
    ax[2][0] = E[1][mx] * E[2][mid] - E[1][mid] * E[2][mx];
    ax[2][1] = E[0][mid] * E[2][mx] - E[0][mx] * E[2][mid];
    ax[2][2] = E[0][mx] * E[1][mid] - E[0][mid] * E[1][mx];
    for (int c = 0; c < 3; ++c)
      for (int r = 0; r < 3; ++r) nd.obb_axes[3 * c + r] = nd.rss_axes[3 * c + r] = ax[c][r];

    // getExtentAndCenter_mesh + projections P for the RSS fit
    const double real_max = std::numeric_limits<double>::max();
    double mnc[3] = {real_max, real_max, real_max}, mxc[3] = {-real_max, -real_max, -real_max};
    std::vector<double> P(size_t(9) * n);
    for (unsigned i = 0; i < n; ++i)
      for (int j = 0; j < 3; ++j) {
        const double* p = vert(idx[i], j);
        double* q = &P[3 * (size_t(3) * i + j)];
        for (int k = 0; k < 3; ++k) {
          q[k] = ax[k][0] * p[0] + ax[k][1] * p[1] + ax[k][2] * p[2];
          if (q[k] > mxc[k]) mxc[k] = q[k];
          if (q[k] < mnc[k]) mnc[k] = q[k];
        }
      }
    double o[3];
    for (int k = 0; k < 3; ++k) {
      o[k] = (mxc[k] + mnc[k]) / 2;
      nd.obb_extent[k] = (mxc[k] - mnc[k]) / 2;
    }
    for (int r = 0; r < 3; ++r) nd.obb_To[r] = ax[0][r] * o[0] + ax[1][r] * o[1] + ax[2][r] * o[2];

    // getRadiusAndOriginAndRectangleSize
    const size_t size_P = size_t(3) * n;
    auto Px = [&](size_t i, int k) { return P[3 * i + k]; };
    double minz = Px(0, 2), maxz = Px(0, 2);
    for (size_t i = 1; i < size_P; ++i) {
      const double zv = Px(i, 2);
      if (zv < minz)
        minz = zv;
      else if (zv > maxz)
        maxz = zv;
    }
    const double r = 0.5 * (maxz - minz), radsqr = r * r, cz = 0.5 * (maxz + minz);
    double lo[2], hi[2];
    for (int k = 0; k < 2; ++k) {  // the x pass, then the identical y pass
      size_t minindex = 0, maxindex = 0;
      double mintmp = Px(0, k), maxtmp = Px(0, k);
      for (size_t i = 1; i < size_P; ++i) {
        const double v = Px(i, k);
        if (v < mintmp) {
          minindex = i;
          mintmp = v;
        } else if (v > maxtmp) {
          maxindex = i;
          maxtmp = v;
        }
      }
      double dz = Px(minindex, 2) - cz;
      double mnv = Px(minindex, k) + std::sqrt(std::max<double>(radsqr - dz * dz, 0));
      dz = Px(maxindex, 2) - cz;
      double mxv = Px(maxindex, k) - std::sqrt(std::max<double>(radsqr - dz * dz, 0));
      for (size_t i = 0; i < size_P; ++i) {
        if (Px(i, k) < mnv) {
          dz = Px(i, 2) - cz;
          const double x = Px(i, k) + std::sqrt(std::max<double>(radsqr - dz * dz, 0));
          if (x < mnv) mnv = x;
        } else if (Px(i, k) > mxv) {
          dz = Px(i, 2) - cz;
          const double x = Px(i, k) - std::sqrt(std::max<double>(radsqr - dz * dz, 0));
          if (x > mxv) mxv = x;
        }
      }
      lo[k] = mnv;
      hi[k] = mxv;
    }
    double minx = lo[0], maxx = hi[0], miny = lo[1], maxy = hi[1];
    const double a = std::sqrt(0.5);
    for (size_t i = 0; i < size_P; ++i) {
      double dx, dy, u, t;
      const double px = Px(i, 0), py = Px(i, 1), pz = Px(i, 2);
      if (px > maxx) {
        if (py > maxy) {
          dx = px - maxx;
          dy = py - maxy;
          u = dx * a + dy * a;
          t = (a * u - dx) * (a * u - dx) + (a * u - dy) * (a * u - dy) + (cz - pz) * (cz - pz);
          u = u - std::sqrt(std::max<double>(radsqr - t, 0));
          if (u > 0) {
            maxx += u * a;
            maxy += u * a;
          }
        } else if (py < miny) {
          dx = px - maxx;
          dy = py - miny;
          u = dx * a - dy * a;
          t = (a * u - dx) * (a * u - dx) + (-a * u - dy) * (-a * u - dy) + (cz - pz) * (cz - pz);
          u = u - std::sqrt(std::max<double>(radsqr - t, 0));
          if (u > 0) {
            maxx += u * a;
            miny -= u * a;
          }
        }
      } else if (px < minx) {
        if (py > maxy) {
          dx = px - minx;
          dy = py - maxy;
          u = dy * a - dx * a;
          t = (-a * u - dx) * (-a * u - dx) + (a * u - dy) * (a * u - dy) + (cz - pz) * (cz - pz);
          u = u - std::sqrt(std::max<double>(radsqr - t, 0));
          if (u > 0) {
            minx -= u * a;
            maxy += u * a;
          }
        } else if (py < miny) {
          dx = px - minx;
          dy = py - miny;
          u = -dx * a - dy * a;
          t = (-a * u - dx) * (-a * u - dx) + (-a * u - dy) * (-a * u - dy) + (cz - pz) * (cz - pz);
          u = u - std::sqrt(std::max<double>(radsqr - t, 0));
          if (u > 0) {
            minx -= u * a;
            miny -= u * a;
          }
        }
      }
    }
    for (int rr = 0; rr < 3; ++rr) nd.rss_Tr[rr] = ax[0][rr] * minx + ax[1][rr] * miny + ax[2][rr] * cz;
    nd.rss_length[0] = std::max<double>(maxx - minx, 0);
    nd.rss_length[1] = std::max<double>(maxy - miny, 0);
    nd.rss_radius = r;
  }

  void recurse(int bv_id, unsigned first, unsigned num) {  // BVH_model.cpp:892-960
    uint32_t* cur = prim + first;
    hfcl_bvh_node& nd = nodes[bv_id];
    fit(cur, num, nd);
    // computeRule_mean
    const double sv[3] = {nd.obb_axes[0], nd.obb_axes[1], nd.obb_axes[2]};
    double c[3] = {0, 0, 0};
    for (unsigned i = 0; i < num; ++i) {
      const double *p1 = vert(cur[i], 0), *p2 = vert(cur[i], 1), *p3 = vert(cur[i], 2);
      for (int k = 0; k < 3; ++k) c[k] += (p1[k] + p2[k]) + p3[k];
    }
    const double split_value = (c[0] * sv[0] + c[1] * sv[1] + c[2] * sv[2]) / (3 * num);
    nd.first_primitive = int32_t(first);
    nd.num_primitives = int32_t(num);
    nd._pad = 0;
    if (num == 1) {
      nd.first_child = -(int32_t(cur[0]) + 1);
      return;
    }
    nd.first_child = int32_t(num_bvs);
    num_bvs += 2;
    unsigned c1 = 0;
    for (unsigned i = 0; i < num; ++i) {
      const double *p1 = vert(cur[i], 0), *p2 = vert(cur[i], 1), *p3 = vert(cur[i], 2);
      double p[3];
      for (int k = 0; k < 3; ++k) p[k] = ((p1[k] + p2[k]) + p3[k]) / 3.;
      if (sv[0] * p[0] + sv[1] * p[1] + sv[2] * p[2] > split_value) {
      } else {
        std::swap(cur[i], cur[c1]);
        c1++;
      }
    }
    if (c1 == 0 || c1 == num) c1 = num / 2;
    recurse(nd.first_child, first, c1);
    recurse(nd.first_child + 1, first + c1, num - c1);
  }
};

}  // namespace

namespace orc {
void jacobi_eigen3(const double M[3][3], double dout[3], double vout[3][3]) { Builder::jacobi(M, dout, vout); }
}  // namespace orc

// nodes: 2*n_tris-1 records; prim: n_tris indices (the model's primitive_indices permutation)
extern "C" int orc_bvh_build(const double* verts, size_t n_verts, const uint32_t* tris, size_t n_tris,
                             hfcl_bvh_node* nodes, uint32_t* prim) {
  if (!n_tris || !n_verts) return HFCL_ERR_INVALID_ARGUMENT;
  Builder b{verts, tris, nodes, prim, 1};
  for (size_t i = 0; i < n_tris; ++i) prim[i] = uint32_t(i);  // BVH_model.cpp:866-868
  b.recurse(0, 0, unsigned(n_tris));
  return HFCL_OK;
}
