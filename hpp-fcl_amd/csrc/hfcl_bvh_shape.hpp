// hfcl_bvh_shape.hpp -- BVHModel<OBBRSS> x convex shape collide(): one query.
//
// Behavioural contract (reference file:line):
//   BVHShapeCollider<OBBRSS,S>::oriented      src/collision_func_matrix.cpp:102-155
//   initialize(MeshShapeCollisionTraversalNode<BV,S,0>)   internal/traversal_node_setup.h:378-404
//   computeBV<OBBRSS,S> = fit of getBoundVertices   shape/geometric_shapes_utility.h:73-82,
//                              src/shape/geometric_shapes_utility.cpp:46-240, src/BVH/BV_fitter.cpp:131-143
//                              (only the OBB half is read by collide(): OBBRSS::overlap -> OBB, OBBRSS.h:88-92)
//   BVDisjoints / leafCollides                 internal/traversal_node_bvh_shape.h:121-186
//   collisionRecurse with a leaf second node   src/traversal/traversal_recurse.cpp:44-85
//   leaf = ShapeShapeDistance<TriangleP,S>     GJK/EPA (shape_shape_func.h table) or
//                                              sphereTriangleDistance (src/narrowphase/details.h:235-340)
//
// The traversal state is sequential by nature (first contact in DFS order, running lower bound), so a
// query is walked by ONE lane group: control flow is group-uniform, the group's lanes share the
// support scans and the EPA face work exactly as in k_epa; the DFS stack and the polytope live in LDS.
// On the host validation build the group is a single lane.
#pragma once
#include "hfcl_bvh.hpp"
#include "hfcl_epa.hpp"

namespace hfcl {

// cyclic Jacobi eigen-decomposition of a symmetric 3x3 (internal/tools.h:103-202); returns false after
// 50 sweeps without convergence (the reference then leaves its outputs unset; we zero them)
template <typename T>
HFCL_HD bool jacobi_eigen3(const T (&Min)[3][3], T (&dout)[3], T (&vout)[3][3]) {
  T R[3][3];
  for (int i = 0; i < 3; ++i)
    for (int j = 0; j < 3; ++j) R[i][j] = Min[i][j];
  T b[3], z[3], d[3], v[3][3] = {{T(1), T(0), T(0)}, {T(0), T(1), T(0)}, {T(0), T(0), T(1)}};
  for (int ip = 0; ip < 3; ++ip) {
    b[ip] = d[ip] = R[ip][ip];
    z[ip] = T(0);
  }
  for (int i = 0; i < 50; ++i) {
    const T sm = (habs(R[0][1]) + habs(R[0][2])) + habs(R[1][2]);
    if (sm == T(0)) {
      for (int a = 0; a < 3; ++a) {
        dout[a] = d[a];
        for (int c = 0; c < 3; ++c) vout[a][c] = v[a][c];
      }
      return true;
    }
    const T tresh = i < 3 ? T(0.2) * sm / T(9) : T(0);
    for (int ip = 0; ip < 3; ++ip)
      for (int iq = ip + 1; iq < 3; ++iq) {
        T g = T(100) * habs(R[ip][iq]);
        if (i > 3 && habs(d[ip]) + g == habs(d[ip]) && habs(d[iq]) + g == habs(d[iq])) {
          R[ip][iq] = T(0);
        } else if (habs(R[ip][iq]) > tresh) {
          T h = d[iq] - d[ip], t;
          if (habs(h) + g == habs(h)) {
            t = R[ip][iq] / h;
          } else {
            const T theta = T(0.5) * h / R[ip][iq];
            t = T(1) / (habs(theta) + hsqrt(T(1) + theta * theta));
            if (theta < T(0)) t = -t;
          }
          const T c = T(1) / hsqrt(T(1) + t * t), s = t * c, tau = s / (T(1) + c);
          h = t * R[ip][iq];
          z[ip] -= h;
          z[iq] += h;
          d[ip] -= h;
          d[iq] += h;
          R[ip][iq] = T(0);
          const int r = 3 - ip - iq;  // the third index of a 3x3
          T* x;
          T* y;
          if (r < ip) {
            x = &R[r][ip];
            y = &R[r][iq];
          } else if (r < iq) {
            x = &R[ip][r];
            y = &R[r][iq];
          } else {
            x = &R[ip][r];
            y = &R[iq][r];
          }
          {
            const T gg = *x, hh = *y;
            *x = gg - s * (hh + gg * tau);
            *y = hh + s * (gg - hh * tau);
          }
          for (int j = 0; j < 3; ++j) {
            const T gg = v[j][ip], hh = v[j][iq];
            v[j][ip] = gg - s * (hh + gg * tau);
            v[j][iq] = hh + s * (gg - hh * tau);
          }
        }
      }
    for (int ip = 0; ip < 3; ++ip) {
      b[ip] += z[ip];
      d[ip] = b[ip];
      z[ip] = T(0);
    }
  }
  for (int a = 0; a < 3; ++a) {
    dout[a] = T(0);
    for (int c = 0; c < 3; ++c) vout[a][c] = T(0);
  }
  return false;
}

// getBoundVertices (geometric_shapes_utility.cpp:46-240): calls f(world point) in the reference's order.
// Returns the number of points (0: kind without a polyhedral bound here).
template <typename T, class F>
HFCL_HD int for_each_bound_vertex(const DShape<T>& s, const T* verts, const Pose<T>& tf, F f) {
  auto add = [&](T x, T y, T z) { f(xform(tf, mk<T>(x, y, z))); };
  if (s.kind == K_BOX) {
    const T a = s.p0, b = s.p1, c = s.p2;
    add(a, b, c); add(a, b, -c); add(a, -b, c); add(a, -b, -c);
    add(-a, b, c); add(-a, b, -c); add(-a, -b, c); add(-a, -b, -c);
    return 8;
  }
  if (s.kind == K_SPHERE || s.kind == K_CAPSULE) {
    const T m = (T(1) + hsqrt(T(5))) / T(2);
    const T edge = s.p0 * T(6) / (hsqrt(T(27)) + hsqrt(T(15)));
    const T a = edge, b = m * edge;
    if (s.kind == K_SPHERE) {
      add(T(0), a, b); add(T(0), -a, b); add(T(0), a, -b); add(T(0), -a, -b);
      add(a, b, T(0)); add(-a, b, T(0)); add(a, -b, T(0)); add(-a, -b, T(0));
      add(b, T(0), a); add(b, T(0), -a); add(-b, T(0), a); add(-b, T(0), -a);
      return 12;
    }
    const T hl = s.p1, r2 = s.p0 * T(2) / hsqrt(T(3));
    add(T(0), a, b + hl); add(T(0), -a, b + hl); add(T(0), a, -b + hl); add(T(0), -a, -b + hl);
    add(a, b, hl); add(-a, b, hl); add(a, -b, hl); add(-a, -b, hl);
    add(b, T(0), a + hl); add(b, T(0), -a + hl); add(-b, T(0), a + hl); add(-b, T(0), -a + hl);
    add(T(0), a, b - hl); add(T(0), -a, b - hl); add(T(0), a, -b - hl); add(T(0), -a, -b - hl);
    add(a, b, -hl); add(-a, b, -hl); add(a, -b, -hl); add(-a, -b, -hl);
    add(b, T(0), a - hl); add(b, T(0), -a - hl); add(-b, T(0), a - hl); add(-b, T(0), -a - hl);
    const T c = T(0.5) * r2, d = s.p0;
    add(r2, T(0), hl); add(c, d, hl); add(-c, d, hl); add(-r2, T(0), hl); add(-c, -d, hl); add(c, -d, hl);
    add(r2, T(0), -hl); add(c, d, -hl); add(-c, d, -hl); add(-r2, T(0), -hl); add(-c, -d, -hl); add(c, -d, -hl);
    return 36;
  }
  if (s.kind == K_ELLIPSOID) {
    const T phi = (T(1) + hsqrt(T(5))) / T(2);
    const T a = hsqrt(T(3)) / (phi * phi), b = phi * a;
    const T Aa = s.p0 * a, Ab = s.p0 * b, Ba = s.p1 * a, Bb = s.p1 * b, Ca = s.p2 * a, Cb = s.p2 * b;
    add(T(0), Ba, Cb); add(T(0), -Ba, Cb); add(T(0), Ba, -Cb); add(T(0), -Ba, -Cb);
    add(Aa, Bb, T(0)); add(-Aa, Bb, T(0)); add(Aa, -Bb, T(0)); add(-Aa, -Bb, T(0));
    add(Ab, T(0), Ca); add(Ab, T(0), -Ca); add(-Ab, T(0), Ca); add(-Ab, T(0), -Ca);
    return 12;
  }
  if (s.kind == K_CONE || s.kind == K_CYLINDER) {
    const T hl = s.p1, r2 = s.p0 * T(2) / hsqrt(T(3)), a = T(0.5) * r2, b = s.p0;
    add(r2, T(0), -hl); add(a, b, -hl); add(-a, b, -hl); add(-r2, T(0), -hl); add(-a, -b, -hl); add(a, -b, -hl);
    if (s.kind == K_CONE) {
      add(T(0), T(0), hl);
      return 7;
    }
    add(r2, T(0), hl); add(a, b, hl); add(-a, b, hl); add(-r2, T(0), hl); add(-a, -b, hl); add(a, -b, hl);
    return 12;
  }
  if (s.kind == K_CONVEX) {
    const T* v = verts + 3 * size_t(s.vertex_offset);
    for (uint32_t i = 0; i < s.num_points; ++i) add(v[3 * i], v[3 * i + 1], v[3 * i + 2]);
    return int(s.num_points);
  }
  return 0;
}

// computeBV<OBBRSS,S>(shape, tf): OBB_fit_functions::fitn (+ RSS_fit_functions::fitn on the same axes when
// `rss` is given: getRadiusAndOriginAndRectangleSize, BVH_utility.cpp:264-482, point-cloud branch; the
// reference stores the projections in an array, here every pass re-projects the bound vertices).
// Returns false when the reference path is not restated (fewer than 4 points, swept-sphere radius).
template <typename T>
HFCL_HD bool shape_obbrss(const DShape<T>& s, const T* verts, const Pose<T>& tf, DNode<T>& bv, DRss<T>* rss) {
  if (s.ssr > T(0)) return false;  // "Swept-sphere radius not yet supported." (geometric_shapes_utility.h:75-78)
  if (s.kind == K_HALFSPACE || s.kind == K_PLANE) {  // unbounded "very rough" volumes, geometric_shapes_utility.cpp:545-581,803-850
    const T big = Lim<T>::max();
    bv.first_child = -1;
    bv.pad_ = 0;
    if (s.kind == K_HALFSPACE) {
      bv.axes.r0 = mk<T>(T(1), T(0), T(0));
      bv.axes.r1 = mk<T>(T(0), T(1), T(0));
      bv.axes.r2 = mk<T>(T(0), T(0), T(1));
      bv.To = mk<T>(T(0), T(0), T(0));
      bv.extent = mk<T>(big, big, big);
      if (rss) {
        rss->Tr = bv.To;
        rss->l0 = rss->l1 = rss->r = big;
      }
      return true;
    }
    const V3<T> n = mul(tf.R, mk<T>(s.p0, s.p1, s.p2));
    V3<T> u, v;  // generateCoordinateSystem, internal/tools.h:60-87
    if (habs(n.x) >= habs(n.y)) {
      const T inv = T(1) / hsqrt(n.x * n.x + n.z * n.z);
      u = mk<T>(-n.z * inv, T(0), n.x * inv);
      v = mk<T>(n.y * u.z, n.z * u.x - n.x * u.z, -n.y * u.x);
    } else {
      const T inv = T(1) / hsqrt(n.y * n.y + n.z * n.z);
      u = mk<T>(T(0), n.z * inv, -n.y * inv);
      v = mk<T>(n.y * u.z - n.z * u.y, -n.x * u.z, n.x * u.y);
    }
    bv.axes.r0 = mk<T>(n.x, u.x, v.x);
    bv.axes.r1 = mk<T>(n.y, u.y, v.y);
    bv.axes.r2 = mk<T>(n.z, u.z, v.z);
    bv.To = xform(tf, mk<T>(s.p0, s.p1, s.p2) * s.p3);
    bv.extent = mk<T>(T(0), big, big);
    if (rss) {
      rss->Tr = bv.To;
      rss->l0 = rss->l1 = big;
      rss->r = T(0);
    }
    return true;
  }
  V3<T> S1 = mk<T>(T(0), T(0), T(0));
  T sxx = T(0), syy = T(0), szz = T(0), sxy = T(0), sxz = T(0), syz = T(0);
  const int n = for_each_bound_vertex(s, verts, tf, [&](const V3<T>& p) {  // getCovariance, point-cloud branch
    S1 = S1 + p;
    sxx += (p.x * p.x);
    syy += (p.y * p.y);
    szz += (p.z * p.z);
    sxy += (p.x * p.y);
    sxz += (p.x * p.z);
    syz += (p.y * p.z);
  });
  if (n < 4) return false;
  const T np = T(n);
  T M[3][3];
  M[0][0] = sxx - S1.x * S1.x / np;
  M[1][1] = syy - S1.y * S1.y / np;
  M[2][2] = szz - S1.z * S1.z / np;
  M[0][1] = M[1][0] = sxy - S1.x * S1.y / np;
  M[1][2] = M[2][1] = syz - S1.y * S1.z / np;
  M[0][2] = M[2][0] = sxz - S1.x * S1.z / np;
  T ev[3], E[3][3];
  jacobi_eigen3(M, ev, E);
  int mn, mid, mx;  // axisFromEigen
  if (ev[0] > ev[1]) {
    mx = 0;
    mn = 1;
  } else {
    mn = 0;
    mx = 1;
  }
  if (ev[2] < ev[mn]) {
    mid = mn;
    mn = 2;
  } else if (ev[2] > ev[mx]) {
    mid = mx;
    mx = 2;
  } else {
    mid = 2;
  }
  (void)mn;
  const V3<T> a0 = mk<T>(E[0][mx], E[1][mx], E[2][mx]), a1 = mk<T>(E[0][mid], E[1][mid], E[2][mid]);
  const V3<T> a2 = mk<T>(E[1][mx] * E[2][mid] - E[1][mid] * E[2][mx], E[0][mid] * E[2][mx] - E[0][mx] * E[2][mid],
                         E[0][mx] * E[1][mid] - E[0][mid] * E[1][mx]);
  const T big = Lim<T>::max();
  V3<T> lo = mk<T>(big, big, big), hi = mk<T>(-big, -big, -big);
  for_each_bound_vertex(s, verts, tf, [&](const V3<T>& p) {  // getExtentAndCenter_pointcloud
    const V3<T> q = mk<T>(dot(a0, p), dot(a1, p), dot(a2, p));
    if (q.x > hi.x) hi.x = q.x;
    if (q.x < lo.x) lo.x = q.x;
    if (q.y > hi.y) hi.y = q.y;
    if (q.y < lo.y) lo.y = q.y;
    if (q.z > hi.z) hi.z = q.z;
    if (q.z < lo.z) lo.z = q.z;
  });
  const V3<T> sum = hi + lo;
  // axes matrix: columns a0 a1 a2 -> rows r_i = (a0[i], a1[i], a2[i])
  bv.axes.r0 = mk<T>(a0.x, a1.x, a2.x);
  bv.axes.r1 = mk<T>(a0.y, a1.y, a2.y);
  bv.axes.r2 = mk<T>(a0.z, a1.z, a2.z);
  bv.To = mul(bv.axes, sum) / T(2);
  bv.extent = (hi - lo) / T(2);
  bv.first_child = -1;
  bv.pad_ = 0;
  if (!rss) return true;
  // ---- RSS on the same axes.  Pass 1: slab along axis 2 (the first point seeds both ends)
  auto prj = [&](const V3<T>& p) { return mk<T>(dot(a0, p), dot(a1, p), dot(a2, p)); };
  T minz = T(0), maxz = T(0);
  int idx = 0;
  for_each_bound_vertex(s, verts, tf, [&](const V3<T>& p) {
    const T z = dot(a2, p);
    if (idx == 0) {
      minz = maxz = z;
    } else if (z < minz) {
      minz = z;
    } else if (z > maxz) {
      maxz = z;
    }
    ++idx;
  });
  const T r = T(0.5) * (maxz - minz), radsqr = r * r, cz = T(0.5) * (maxz + minz);
  auto cap = [&](T z) {
    const T dz = z - cz;
    return hsqrt(hmax(radsqr - dz * dz, T(0)));
  };
  T rlo[2], rhi[2];
  for (int k = 0; k < 2; ++k) {
    // extreme points along axis k (strictly smaller / larger than everything before, as the reference's scan)
    T wlo = T(0), whi = T(0), zlo = T(0), zhi = T(0);
    idx = 0;
    for_each_bound_vertex(s, verts, tf, [&](const V3<T>& p) {
      const V3<T> q = prj(p);
      const T w = k == 0 ? q.x : q.y;
      if (idx == 0) {
        wlo = whi = w;
        zlo = zhi = q.z;
      } else if (w < wlo) {
        wlo = w;
        zlo = q.z;
      } else if (w > whi) {
        whi = w;
        zhi = q.z;
      }
      ++idx;
    });
    T lo_k = wlo + cap(zlo), hi_k = whi - cap(zhi);
    for_each_bound_vertex(s, verts, tf, [&](const V3<T>& p) {
      const V3<T> q = prj(p);
      const T w = k == 0 ? q.x : q.y;
      if (w < lo_k) {
        const T x = w + cap(q.z);
        if (x < lo_k) lo_k = x;
      } else if (w > hi_k) {
        const T x = w - cap(q.z);
        if (x > hi_k) hi_k = x;
      }
    });
    rlo[k] = lo_k;
    rhi[k] = hi_k;
  }
  const T h = hsqrt(T(0.5));
  for_each_bound_vertex(s, verts, tf, [&](const V3<T>& p) {  // corner growth
    const V3<T> q = prj(p);
    const int sx = q.x > rhi[0] ? 1 : (q.x < rlo[0] ? -1 : 0);
    if (!sx) return;
    const int sy = q.y > rhi[1] ? 1 : (q.y < rlo[1] ? -1 : 0);
    if (!sy) return;
    const T dx = q.x - (sx > 0 ? rhi[0] : rlo[0]), dy = q.y - (sy > 0 ? rhi[1] : rlo[1]);
    T u;
    if (sx > 0 && sy > 0)
      u = dx * h + dy * h;
    else if (sx > 0)
      u = dx * h - dy * h;
    else if (sy > 0)
      u = dy * h - dx * h;
    else
      u = -dx * h - dy * h;
    const T ex = (sx > 0 ? h * u : -h * u) - dx, ey = (sy > 0 ? h * u : -h * u) - dy;
    const T t = ex * ex + ey * ey + (cz - q.z) * (cz - q.z);
    u = u - hsqrt(hmax(radsqr - t, T(0)));
    if (u > T(0)) {
      if (sx > 0)
        rhi[0] += u * h;
      else
        rlo[0] -= u * h;
      if (sy > 0)
        rhi[1] += u * h;
      else
        rlo[1] -= u * h;
    }
  });
  // origin = axes * (minx, miny, cz)
  rss->Tr = mk<T>(a0.x * rlo[0] + a1.x * rlo[1] + a2.x * cz, a0.y * rlo[0] + a1.y * rlo[1] + a2.y * cz,
                  a0.z * rlo[0] + a1.z * rlo[1] + a2.z * cz);
  rss->l0 = hmax(rhi[0] - rlo[0], T(0));
  rss->l1 = hmax(rhi[1] - rlo[1], T(0));
  rss->r = r;
  return true;
}
template <typename T>
HFCL_HD bool shape_obb(const DShape<T>& s, const T* verts, const Pose<T>& tf, DNode<T>& bv) {
  return shape_obbrss<T>(s, verts, tf, bv, nullptr);
}

#if defined(HFCL_LEAF_CONTRACT_OFF)
#pragma clang fp contract(off)  // (the leaves' arithmetic: hfcl_bvh.hpp)
#endif
// details::segmentSqrDistance :235-255, projectInTriangle :258-279, sphereTriangleDistance :286-340
template <typename T>
HFCL_HD T segment_sqr_distance(const V3<T>& from, const V3<T>& to, const V3<T>& p, V3<T>& nearest) {
  V3<T> diff = p - from;
  const V3<T> v = to - from;
  T t = dot(v, diff);
  if (t > T(0)) {
    const T vv = sqnorm(v);
    if (t < vv) {
      t /= vv;
      diff = diff - v * t;
    } else {
      t = T(1);
      diff = diff - v;
    }
  } else {
    t = T(0);
  }
  nearest = from + v * t;
  return sqnorm(diff);
}
template <typename T>
HFCL_HD T sphere_triangle(const DShape<T>& s, const Pose<T>& tf1, const V3<T>& P1, const V3<T>& P2, const V3<T>& P3, V3<T>& p1,
                          V3<T>& p2, V3<T>& normal) {
  V3<T> tn = normalized(cross(P2 - P1, P3 - P1));
  const V3<T> center = tf1.t;
  const T radius = s.p0 + s.ssr;  // + the triangle's swept-sphere radius (0 for mesh triangles)
  T dplane = dot(center - P1, tn);
  if (dplane < T(0)) {
    dplane = -dplane;
    tn = tn * T(-1);
  }
  const T r1 = dot(cross(P2 - P1, tn), center - P1), r2 = dot(cross(P3 - P2, tn), center - P2),
          r3 = dot(cross(P1 - P3, tn), center - P3);
  V3<T> closest;
  T mind;
  if ((r1 > T(0) && r2 > T(0) && r3 > T(0)) || (r1 <= T(0) && r2 <= T(0) && r3 <= T(0))) {
    closest = center - tn * dplane;
    mind = dplane * dplane;
  } else {
    V3<T> e;
    mind = segment_sqr_distance(P1, P2, center, closest);
    T d2 = segment_sqr_distance(P2, P3, center, e);
    if (d2 < mind) {
      mind = d2;
      closest = e;
    }
    d2 = segment_sqr_distance(P3, P1, center, e);
    if (d2 < mind) {
      mind = d2;
      closest = e;
    }
  }
  normal = normalized(closest - center);
  p1 = center + normal * (s.p0 + s.ssr);
  p2 = closest;
  return hsqrt(mind) - radius;
}

// MinkowskiDiff of (solid shape, TriangleP moved into the solid's frame): shape 0 = the solid, identity
// relative transform (MinkowskiDiff::set(&s1, &tri), narrowphase.h:320-336).
template <typename T, class Solid>
struct SolidTriSupport {
  V3<T> a, b, c;       // the triangle in the solid's frame
  const Solid* solid;  // V3<T> (*solid)(dir): support of the solid in its own frame (NoSweptSphere)
  HFCL_HD void operator()(const V3<T>& dir, V3<T>& w, V3<T>& w0) const {
    w0 = (*solid)(dir);
    w = w0 - tri_support(a, b, c, -dir);
  }
};

// ShapeShapeDistance<TriangleP, S>(tri, tf_tri, solid, tf_solid) for the solids that go through GJK/EPA:
// overload resolution in GJKSolver::shapeDistance (narrowphase.h:320-348) turns it into the swapped call
// shapeDistance(solid, tf_solid, tri, tf_tri) with the triangle moved into the solid's frame by
// tf_solid.inverseTimes(tf_tri), the relative transform precomputed (identity), and on return the points
// exchanged and the normal negated.  `o` keeps the statuses / cached guess of that run (solid = shape 0);
// p_tri / p_solid / n are the world-frame results in (triangle, solid) order.
template <typename T, class Grp, class Solid>
HFCL_HD T triangle_solid_distance(const V3<T>& ta, const V3<T>& tb, const V3<T>& tc, const Pose<T>& tft, const Pose<T>& tfs,
                                  const Solid& solid, T r_solid, const QParams<T>& q, const V3<T>& guess0,
                                  EpaScratch<T, EPA_MAX_ITER>* scratch, PairOut<T>& o, V3<T>& p_tri, V3<T>& p_solid, V3<T>& n) {
  const MDiff<T> sMt = make_mdiff(tfs, tft);  // Transform3f::inverseTimes
  SolidTriSupport<T, Solid> sup;
  sup.a = mul(sMt.oR1, ta) + sMt.ot1;
  sup.b = mul(sMt.oR1, tb) + sMt.ot1;
  sup.c = mul(sMt.oR1, tc) + sMt.ot1;
  sup.solid = &solid;
  Gjk<T, PW0<T>> g;
  gjk_run(g, q.gjk, guess0, r_solid, false, sup);
  EpaSeed<T> seed;
  if (gjk_finish(g, q, tfs, r_solid, T(0), guess0, o, seed)) {
    Grp::sync();
    epa_run<T, Grp, EPA_MAX_ITER>(scratch, seed, q, tfs, r_solid, T(0), sup, o);
    Grp::sync();
  }
  p_tri = o.p2;
  p_solid = o.p1;
  n = -o.normal;
  return o.distance;
}

// One top-level pair with a TriangleP on either side (not against Plane / Halfspace: closed forms):
// TriangleP x TriangleP (triangle_triangle.cpp:46-105), TriangleP x Sphere (triangle_sphere.cpp:45-68), and
// the GJK solids through triangle_solid_distance.  `solid` = support of the non-triangle shape.
template <typename T, class Grp, class Solid>
HFCL_HD void triangle_pair(const DShape<T>& a, const DShape<T>& b, const T* verts, const Pose<T>& tf1, const Pose<T>& tf2,
                           const Solid& solid, const QParams<T>& q, const V3<T>& guess_in,
                           EpaScratch<T, EPA_MAX_ITER>* scratch, PairOut<T>& o) {
  auto vtx = [&](const DShape<T>& s, uint32_t i) {
    const T* p = verts + 3 * (size_t(s.vertex_offset) + i);
    return mk<T>(p[0], p[1], p[2]);
  };
  o.gjk_status = GJK_DID_NOT_RUN;
  o.epa_status = EPA_DID_NOT_RUN;
  o.gjk_iters = o.epa_iters = 0;
  o.cached_guess = guess_in;
  const bool t1 = a.kind == K_TRIANGLE, t2 = b.kind == K_TRIANGLE;
  if (t1 && t2) {
    TriSupport<T> ts;
    ts.p1 = xform(tf1, vtx(a, 0)); ts.p2 = xform(tf1, vtx(a, 1)); ts.p3 = xform(tf1, vtx(a, 2));
    ts.q1 = xform(tf2, vtx(b, 0)); ts.q2 = xform(tf2, vtx(b, 1)); ts.q3 = xform(tf2, vtx(b, 2));
    o.distance = tri_tri_distance(ts, q.gjk, q.guess_mode == HFCL_GUESS_CACHED, guess_in, o.p1, o.p2, o.normal, o.gjk_status,
                                  o.gjk_iters, &o.cached_guess);
    return;
  }
  const DShape<T>& tri = t1 ? a : b;
  const DShape<T>& sol = t1 ? b : a;
  const Pose<T>& tft = t1 ? tf1 : tf2;
  const Pose<T>& tfs = t1 ? tf2 : tf1;
  const V3<T> ta = vtx(tri, 0), tb = vtx(tri, 1), tc = vtx(tri, 2);
  V3<T> p_tri, p_solid, n;  // n: from the triangle to the solid
  if (sol.kind == K_SPHERE) {
    o.distance = sphere_triangle(sol, tfs, xform(tft, ta), xform(tft, tb), xform(tft, tc), p_solid, p_tri, n);
    n = -n;
  } else {
    const V3<T> guess0 = (q.guess_mode == HFCL_GUESS_CACHED) ? guess_in : mk<T>(T(1), T(0), T(0));
    PairOut<T> og;
    const T d = triangle_solid_distance<T, Grp>(ta, tb, tc, tft, tfs, solid, swept_radius(sol), q, guess0, scratch, og, p_tri,
                                                p_solid, n);
    o = og;
    o.distance = d;
  }
  o.p1 = t1 ? p_tri : p_solid;
  o.p2 = t1 ? p_solid : p_tri;
  o.normal = t1 ? n : -n;
}

#if defined(HFCL_LEAF_CONTRACT_OFF)
#pragma clang fp contract(fast)
#endif
template <typename T>
struct MeshShapeState {  // the CollisionResult fields the traversal maintains
  T dlb, rec_dist;
  V3<T> np1, np2, nn;
  uint32_t ncontacts;
  int first_prim;
  V3<T> guess;  // the solver's cached guess (persists over the leaves of one query)
  bool overflow, unsupported;
};

// One (mesh, solid) query.  stack: cap entries of group-shared memory; scratch: group-shared EPA block.
// on_contact(prim, distance, p1, p2, n) is called for every Contact added (lane-uniform values).
template <typename T, class Grp, class Solid, class OnContact>
HFCL_HD void mesh_shape_collide(const DNode<T>* nodes, const T* mverts, const uint32_t* tris, const Pose<T>& tfm,
                                const DShape<T>& shape, const T* sverts, const Pose<T>& tfs, const Solid& solid,
                                const QParams<T>& q, uint32_t num_max_contacts, T break_distance2, uint32_t* stack, int cap,
                                EpaScratch<T, EPA_MAX_ITER>* scratch, const V3<T>& initial_guess, OnContact on_contact,
                                MeshShapeState<T>& st) {
  const T nanv = Lim<T>::nan();
  st.guess = initial_guess;
  st.dlb = st.rec_dist = Lim<T>::max();
  st.np1 = st.np2 = st.nn = mk<T>(nanv, nanv, nanv);
  st.ncontacts = 0;
  st.first_prim = -1;
  st.overflow = false;
  st.unsupported = false;
  DNode<T> bv2;
  if (!shape_obb(shape, sverts, tfs, bv2)) {
    st.unsupported = true;
    return;
  }
  const T r1 = swept_radius(shape);
  int sp = 0;
  Grp::sync();
  if (Grp::lane() == 0) stack[0] = 0;
  sp = 1;
  Grp::sync();
  while (sp > 0) {
    const uint32_t b1 = stack[--sp];
    const DNode<T> n1 = nodes[b1];
    if (n1.first_child >= 0) {
      T sq;
      const bool disjoint = obb_disjoint(tfm.R, tfm.t, n1, bv2, q.security_margin, break_distance2, sq);
      if (disjoint) {  // updateDistanceLowerBoundFromBV
        if (!(st.dlb <= T(0))) {
          const T nd = hsqrt(sq);
          if (nd < st.dlb) {
            st.dlb = nd;
            st.rec_dist = nd + q.security_margin;
          }
        }
        continue;
      }
      if (sp + 2 > cap) {
        st.overflow = true;
        return;
      }
      Grp::sync();
      if (Grp::lane() == 0) {
        stack[sp] = uint32_t(n1.first_child + 1);  // right child below
        stack[sp + 1] = uint32_t(n1.first_child);  // left child on top
      }
      sp += 2;
      Grp::sync();
      continue;
    }
    // ---- leaf: ShapeShapeDistance<TriangleP, S>(tri, tf_mesh, shape, tf_shape)
    const uint32_t prim = uint32_t(-(n1.first_child + 1));
    const uint32_t* t3 = tris + 3 * size_t(prim);
    auto vtx = [&](uint32_t i) { return mk<T>(mverts[3 * size_t(i)], mverts[3 * size_t(i) + 1], mverts[3 * size_t(i) + 2]); };
    const V3<T> ta = vtx(t3[0]), tb = vtx(t3[1]), tc = vtx(t3[2]);
    T distance;
    V3<T> p1, p2, n;
    if (shape.kind == K_SPHERE) {  // triangle_sphere.cpp:45-56
      distance = sphere_triangle(shape, tfs, xform(tfm, ta), xform(tfm, tb), xform(tfm, tc), p2, p1, n);
      n = -n;
    } else if (kind_is_flat(shape.kind)) {  // triangle_halfspace.cpp:46-57, triangle_plane.cpp:46-57
      distance = flat_triangle_distance(shape, tfs, ta, tb, tc, tfm, p2, p1, n);
      n = -n;
    } else {
      const V3<T> guess0 = (q.guess_mode == HFCL_GUESS_CACHED) ? st.guess : mk<T>(T(1), T(0), T(0));
      PairOut<T> o;
      distance = triangle_solid_distance<T, Grp>(ta, tb, tc, tfm, tfs, solid, r1, q, guess0, scratch, o, p1, p2, n);
      // GJK::Collision without penetration information leaves the solver's cached guess untouched
      // (narrowphase.h:638-656): the previous leaf's value persists
      if (!(o.gjk_status == GJK_COLLISION && !q.compute_penetration)) st.guess = o.cached_guess;
    }
    const T dtc = distance - q.security_margin;
    if (dtc < st.dlb) {  // updateDistanceLowerBoundFromLeaf
      st.dlb = dtc;
      st.rec_dist = distance;
      st.np1 = p1;
      st.np2 = p2;
      st.nn = n;
    }
    if (dtc <= q.collision_distance_threshold) {
      if (st.ncontacts < num_max_contacts) {
        if (st.ncontacts == 0) st.first_prim = int(prim);
        ++st.ncontacts;
        on_contact(int(prim), distance, p1, p2, n);
      }
      if (st.ncontacts >= num_max_contacts) return;  // canStop()
    }
  }
}


// ---------------------------------------------------------------------------------------
// The same traversals with one query per LANE (hfcl_k_bvh.hip: k_bvh_collide<SOLID>, k_bvh_shape_distance_lane, and the
// kernels that continue their long walks 64 stack entries at a time).
//
// The group form above spends a 16-lane group on a walk whose separating-axis tests and (for every solid but a
// ConvexBase) GJK iterations no lane can share: 100k queries against a 5 000-triangle model ran at 1-6 M q/s, below
// what the host's cores reach with the oracle (tools/mesh_solid_bench.py).  Here a lane owns a query: it walks
// the tree and runs the leaf's closed form / GJK by itself.  What a lane cannot hold is EPA's polytope -- but a leaf
// that needs EPA ENDS the walk where the fast path applies: collide() with num_max_contacts == 1 and margin,
// threshold >= 0 (a penetrating triangle is a contact, canStop()), distance() always (every bound left on the stack
// is >= 0 > the penetration found).  Such a leaf is written to a queue (GJK's final simplex, the triangle, the
// running bound) and finished by a group kernel that runs EPA and the leaf's epilogue (k_bvh_shape_finish).
// Same functions, same argument order as the group form: on the host build the two forms produce identical records
// (tests/test_bvh_shape.py).
// ---------------------------------------------------------------------------------------

// obb_disjoint(R0, T0, node, bv2) with the node-independent products taken out of the walk: the same operations in
// the same order (M = R0^T A2, V = R0^T (To2 - T0) are what obb_disjoint computes first)
template <typename T>
struct ObbQuery {
  M3<T> M;
  V3<T> V, ext;
};
template <typename T>
HFCL_HD ObbQuery<T> make_obb_query(const Pose<T>& tfm, const DNode<T>& bv2) {
  ObbQuery<T> o;
  o.M = tmul(tfm.R, bv2.axes);
  o.V = tmul(tfm.R, bv2.To - tfm.t);
  o.ext = bv2.extent;
  return o;
}
template <typename T>
HFCL_HD bool obb_disjoint_q(const ObbQuery<T>& o, const DNode<T>& b1, T security_margin, T break_distance2, T& sq) {
  const V3<T> Ttemp = o.V - b1.To;
  const V3<T> Tv = tmul(b1.axes, Ttemp);
  const M3<T> R = tmul(b1.axes, o.M);
  return obb_disjoint_lb(R, Tv, b1.extent, o.ext, security_margin, break_distance2, sq);
}

// distance(): what rss_lower_bound reads of the solid's fitted OBBRSS (computeBV<OBBRSS, S>, world frame), per query
template <typename T>
struct RssQuery {
  M3<T> axes;
  V3<T> Tr;
  T l0, l1, r;
};

template <typename T>
struct ShapeDeferItem {  // a leaf whose GJK ended inside the solid (seed.pair = the query)
  EpaSeed<T> seed;
  V3<T> a, b, c;       // the triangle in the solid's frame, as GJK saw it
  T bound, rec;        // collide(): distance_lower_bound and the recorded distance before this leaf; distance(): min_distance
  uint32_t prim;
  int32_t prev_prim;   // distance(): the triangle of the minimum so far
  uint32_t parent, order;  // collide(): where the unit that met the leaf hangs in the task tree (0xFFFFFFFF: a whole query)
};

#if defined(HFCL_LEAF_CONTRACT_OFF)
#pragma clang fp contract(off)  // (the leaves' arithmetic: hfcl_bvh.hpp)
#endif
// A leaf on one lane.  Returns true when the leaf must go through EPA (item.seed / a / b / c filled; nothing else
// changed); otherwise distance / p1 (on the triangle) / p2 (on the solid) / n are the leaf's result and `guess` the
// solver's cached guess after it.  tfm_of() / tfs_of(): the poses, produced where they are needed.
template <typename T, class Solid, class PS, class TfM, class TfS>
HFCL_HD bool mesh_shape_leaf_lane(const V3<T>& ta, const V3<T>& tb, const V3<T>& tc, const MDiff<T>& sMt, const TfM& tfm_of,
                                  const TfS& tfs_of, const DShape<T>& shape, const Solid& solid, T r1, const QParams<T>& q,
                                  V3<T>& guess, const PS& ps, T& distance, V3<T>& p1, V3<T>& p2, V3<T>& n,
                                  ShapeDeferItem<T>& item) {
  if (shape.kind == K_SPHERE) {
    const Pose<T> tfm = tfm_of();
    distance = sphere_triangle(shape, tfs_of(), xform(tfm, ta), xform(tfm, tb), xform(tfm, tc), p2, p1, n);
    n = -n;
    return false;
  }
  if (kind_is_flat(shape.kind)) {
    distance = flat_triangle_distance(shape, tfs_of(), ta, tb, tc, tfm_of(), p2, p1, n);
    n = -n;
    return false;
  }
  const V3<T> guess0 = (q.guess_mode == HFCL_GUESS_CACHED) ? guess : mk<T>(T(1), T(0), T(0));
  SolidTriSupport<T, Solid> sup;
  sup.a = mul(sMt.oR1, ta) + sMt.ot1;
  sup.b = mul(sMt.oR1, tb) + sMt.ot1;
  sup.c = mul(sMt.oR1, tc) + sMt.ot1;
  sup.solid = &solid;
  Gjk<T, typename PS::P> g;
  gjk_run(g, q.gjk, guess0, r1, false, sup, ps);
  PairOut<T> o;
  if (gjk_finish(g, q, tfs_of, r1, T(0), guess0, o, item.seed, ps)) {
    item.a = sup.a;
    item.b = sup.b;
    item.c = sup.c;
    return true;
  }
  p1 = o.p2;
  p2 = o.p1;
  n = -o.normal;
  distance = o.distance;
  // GJK::Collision without penetration information leaves the solver's cached guess untouched (narrowphase.h:638-656)
  if (!(o.gjk_status == GJK_COLLISION && !q.compute_penetration)) guess = o.cached_guess;
  return false;
}

// The EPA half of a deferred leaf, by a lane group: returns the leaf's distance, p1 (triangle) / p2 (solid) / n and the
// solver's cached guess after it.
// done (optional): false when the polytope outgrew the CAP-sized block (nothing else is meaningful then: the full-capacity tier redoes it).
template <typename T, class Grp, class Solid, int CAP = EPA_MAX_ITER>
HFCL_HD T mesh_shape_leaf_finish(const ShapeDeferItem<T>& item, const Pose<T>& tfs, const Solid& solid, T r1, const QParams<T>& q,
                                 EpaScratch<T, CAP>* scratch, V3<T>& p1, V3<T>& p2, V3<T>& n, V3<T>& guess, bool* done = nullptr) {
  SolidTriSupport<T, Solid> sup;
  sup.a = item.a;
  sup.b = item.b;
  sup.c = item.c;
  sup.solid = &solid;
  PairOut<T> o;
  Grp::sync();
  const int rc = epa_run<T, Grp, CAP>(scratch, item.seed, q, tfs, r1, T(0), sup, o);
  Grp::sync();
  if (done) *done = rc == 1;
  p1 = o.p2;
  p2 = o.p1;
  n = -o.normal;
  guess = o.cached_guess;
  return o.distance;
}

#if defined(HFCL_LEAF_CONTRACT_OFF)
#pragma clang fp contract(fast)
#endif
// updateDistanceLowerBoundFromLeaf + the contact decision of leafCollides (traversal_node_bvh_shape.h:139-186).
// lowered: the leaf's witness data replace the recorded ones.
template <typename T>
HFCL_HD bool mesh_shape_leaf_bound(T distance, const QParams<T>& q, T& dlb, T& rec_dist, bool& lowered) {
  const T dtc = distance - q.security_margin;
  lowered = dtc < dlb;
  if (lowered) {
    dlb = dtc;
    rec_dist = distance;
  }
  return dtc <= q.collision_distance_threshold;
}
// updateDistanceLowerBoundFromBV for a node found disjoint (sq = the squared lower bound of obb_disjoint)
template <typename T>
HFCL_HD void mesh_shape_bv_bound(T sq, const QParams<T>& q, T& dlb, T& rec_dist) {
  if (!(dlb <= T(0))) {
    const T nd = hsqrt(sq);
    if (nd < dlb) {
      dlb = nd;
      rec_dist = nd + q.security_margin;
    }
  }
}
// Whether a collide() request may take the one-query-per-lane path (see above).
template <typename T>
HFCL_HD bool mesh_shape_lane_request(const QParams<T>& q, uint32_t num_max_contacts) {
  return num_max_contacts == 1u && q.security_margin >= T(0) && q.collision_distance_threshold >= T(0);
}


// ---------------------------------------------------------------------------------------
// distance(): MeshShapeDistanceTraversalNodeOBBRSS (traversal_node_bvh_shape.h:276-478) + distanceRecurse
// (traversal_recurse.cpp:153-203) with a leaf second node, rel_err = abs_err = 0.  stack_n / stack_d: cap
// entries each of group-shared memory (node id, RSS lower bound of the pending subtree).
// ---------------------------------------------------------------------------------------
template <typename T>
struct MeshShapeDist {
  T min_distance;
  int prim;
  V3<T> np1, np2, nn;
  V3<T> guess;
  bool overflow, unsupported;
};

template <typename T, class Grp, class Solid>
HFCL_HD void mesh_shape_distance(const DNode<T>* nodes, const DRss<T>* rss, const T* mverts, const uint32_t* tris,
                                 const Pose<T>& tfm, const DShape<T>& shape, const T* sverts, const Pose<T>& tfs,
                                 const Solid& solid, const QParams<T>& q, uint32_t* stack_n, T* stack_d, int cap,
                                 EpaScratch<T, EPA_MAX_ITER>* scratch, const V3<T>& initial_guess, MeshShapeDist<T>& st) {
  const T nanv = Lim<T>::nan();
  st.min_distance = Lim<T>::max();
  st.prim = -1;
  st.np1 = st.np2 = st.nn = mk<T>(nanv, nanv, nanv);
  st.guess = initial_guess;
  st.overflow = st.unsupported = false;
  DNode<T> bv2;
  DRss<T> rss2;
  if (!shape_obbrss(shape, sverts, tfs, bv2, &rss2)) {
    st.unsupported = true;
    return;
  }
  const T r1 = swept_radius(shape);
  auto leaf = [&](uint32_t prim) {
    const uint32_t* t3 = tris + 3 * size_t(prim);
    auto vtx = [&](uint32_t i) { return mk<T>(mverts[3 * size_t(i)], mverts[3 * size_t(i) + 1], mverts[3 * size_t(i) + 2]); };
    const V3<T> ta = vtx(t3[0]), tb = vtx(t3[1]), tc = vtx(t3[2]);
    T distance;
    V3<T> p1, p2, n;
    if (shape.kind == K_SPHERE) {
      distance = sphere_triangle(shape, tfs, xform(tfm, ta), xform(tfm, tb), xform(tfm, tc), p2, p1, n);
      n = -n;
    } else if (kind_is_flat(shape.kind)) {
      distance = flat_triangle_distance(shape, tfs, ta, tb, tc, tfm, p2, p1, n);
      n = -n;
    } else {
      const V3<T> guess0 = (q.guess_mode == HFCL_GUESS_CACHED) ? st.guess : mk<T>(T(1), T(0), T(0));
      PairOut<T> o;
      distance = triangle_solid_distance<T, Grp>(ta, tb, tc, tfm, tfs, solid, r1, q, guess0, scratch, o, p1, p2, n);
      // GJK::Collision without penetration information leaves the solver's cached guess untouched
      // (narrowphase.h:638-656): the previous leaf's value persists
      if (!(o.gjk_status == GJK_COLLISION && !q.compute_penetration)) st.guess = o.cached_guess;
    }
    if (st.min_distance > distance) {  // DistanceResult::update
      st.min_distance = distance;
      st.prim = int(prim);
      st.np1 = p1;
      st.np2 = p2;
      st.nn = n;
    }
  };
  leaf(0u);  // preprocess(): triangle 0
  int sp = 0;
  Grp::sync();
  if (Grp::lane() == 0) {
    stack_n[0] = 0;
    stack_d[0] = T(-1);
  }
  sp = 1;
  Grp::sync();
  while (sp > 0) {
    --sp;
    const uint32_t b = stack_n[sp];
    const T db = stack_d[sp];
    if (db >= T(0) && db >= st.min_distance) continue;  // canStop(d)
    const DNode<T> n1 = nodes[b];
    if (n1.first_child < 0) {
      leaf(uint32_t(-(n1.first_child + 1)));
      continue;
    }
    const uint32_t a1 = uint32_t(n1.first_child), c1 = a1 + 1;
    // distance(tf1.R, tf1.T, model2_bv, model1.bv(b)): the mesh node placed by the mesh pose, seen from the solid's RSS
    const T d1 = rss_lower_bound(tfm.R, tfm.t, bv2, rss2, nodes[a1], rss[a1]);
    const T d2 = rss_lower_bound(tfm.R, tfm.t, bv2, rss2, nodes[c1], rss[c1]);
    if (sp + 2 > cap) {
      st.overflow = true;
      return;
    }
    const bool c_first = d2 < d1;
    Grp::sync();
    if (Grp::lane() == 0) {
      stack_n[sp] = uint32_t(c_first ? a1 : c1);  // visited second
      stack_d[sp] = c_first ? d1 : d2;
      stack_n[sp + 1] = uint32_t(c_first ? c1 : a1);  // visited first
      stack_d[sp + 1] = c_first ? d2 : d1;
    }
    sp += 2;
    Grp::sync();
  }
}

}  // namespace hfcl
