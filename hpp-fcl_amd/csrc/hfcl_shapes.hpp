// hfcl_shapes.hpp -- device shape records, primitive support functions, closed-form pairs.
//
// Behavioural contract (reference file:line):
//   supports      src/narrowphase/support_functions.cpp:140-317 (Box, Sphere, Ellipsoid, Capsule, Cone, Cylinder;
//                 NoSweptSphere option: sphere/capsule radii are added back after GJK/EPA,
//                 src/narrowphase/minkowski_difference.cpp:103-125,183-201)
//   closed forms  src/narrowphase/details.h:52-101 (sphere-capsule), :215-232 (sphere-sphere),
//                 :435-495 (box-sphere), :107-209 (sphere-cylinder), src/distance/capsule_capsule.cpp:52-167
#pragma once
#include "hfcl_math.hpp"

namespace hfcl {

enum { K_BOX = 9, K_SPHERE = 10, K_CAPSULE = 11, K_CONE = 12, K_CYLINDER = 13, K_CONVEX = 14, K_PLANE = 15, K_HALFSPACE = 16, K_TRIANGLE = 17, K_ELLIPSOID = 19, K_BVH = 5 };

// Device image of one shape-library entry (both an fp64 and an fp32 table are resident).
template <typename T>
struct DShape {
  int32_t kind;
  uint32_t num_points;
  uint32_t vertex_offset;  // in vertices, into the library vertex array of the same precision
  uint32_t bvh_index;
  T p0, p1, p2;            // Box halfSide / Sphere r / Capsule r,halfLength / Ellipsoid radii / Plane, Halfspace n
  T p3;                    // Plane, Halfspace: d
  T ssr;                   // swept sphere radius
};

// pair classes (kernel buckets), from the reference's dispatch table
// (include/hpp/fcl/internal/shape_shape_func.h:185-211)
enum { CLS_CLOSED = 0, CLS_PRIM_GJK = 1, CLS_CONVEX = 2, CLS_BVH = 3, CLS_UNSUPPORTED = 4, CLS_COUNT = 5 };

HFCL_HD bool kind_is_prim(int k) {
  return k == K_BOX || k == K_SPHERE || k == K_CAPSULE || k == K_ELLIPSOID || k == K_CONE || k == K_CYLINDER;
}
HFCL_HD bool kind_is_flat(int k) { return k == K_PLANE || k == K_HALFSPACE; }
HFCL_HD int pair_class(int k1, int k2) {
  if (k1 == K_BVH && k2 == K_BVH) return CLS_BVH;
  if (kind_is_flat(k1) || kind_is_flat(k2)) {  // every Plane / Halfspace row of the table is a closed form
    const int o = kind_is_flat(k1) ? k2 : k1;
    return (kind_is_flat(o) || kind_is_prim(o) || o == K_CONVEX || o == K_TRIANGLE) ? CLS_CLOSED : CLS_UNSUPPORTED;
  }
  const bool p1 = kind_is_prim(k1), p2 = kind_is_prim(k2);
  if (p1 && p2) {
    const bool sc1 = (k1 == K_SPHERE || k1 == K_CAPSULE), sc2 = (k2 == K_SPHERE || k2 == K_CAPSULE);
    if (sc1 && sc2) return CLS_CLOSED;                                              // sph-sph, sph-cap, cap-cap
    if ((k1 == K_BOX && k2 == K_SPHERE) || (k1 == K_SPHERE && k2 == K_BOX)) return CLS_CLOSED;  // box-sphere
    if ((k1 == K_CYLINDER && k2 == K_SPHERE) || (k1 == K_SPHERE && k2 == K_CYLINDER)) return CLS_CLOSED;
    return CLS_PRIM_GJK;
  }
  if ((p1 || k1 == K_CONVEX) && (p2 || k2 == K_CONVEX)) return CLS_CONVEX;
  return CLS_UNSUPPORTED;
}

// The Box support's `inflate` is a function-local static evaluated on the first call of the
// process in the reference (support_functions.cpp:146); pinned to 1+1e-10.
template <typename T> HFCL_HD T box_inflate() { return T(1) + T(1e-10); }

// getShapeSupport<NoSweptSphere> for the primitive kinds, in the shape's own frame.
template <typename T>
HFCL_HD V3<T> prim_support(const DShape<T>& s, const V3<T>& dir) {
  const T tiny = Lim<T>::tiny();
  if (s.kind == K_BOX) {
    const T inf = box_inflate<T>();
    V3<T> r;
    r.x = ((dir.x > tiny) ? s.p0 : T(0)) + ((dir.x < -tiny) ? (-inf * s.p0) : T(0));
    r.y = ((dir.y > tiny) ? s.p1 : T(0)) + ((dir.y < -tiny) ? (-inf * s.p1) : T(0));
    r.z = ((dir.z > tiny) ? s.p2 : T(0)) + ((dir.z < -tiny) ? (-inf * s.p2) : T(0));
    return r;
  }
  if (s.kind == K_ELLIPSOID) {
    const T a2 = s.p0 * s.p0, b2 = s.p1 * s.p1, c2 = s.p2 * s.p2;
    const V3<T> v = mk<T>(a2 * dir.x, b2 * dir.y, c2 * dir.z);
    const T d = hsqrt(dot(v, dir));
    return v / d;
  }
  if (s.kind == K_CAPSULE) {
    T z = T(0);
    if (dir.z > tiny)
      z = s.p1;
    else if (dir.z < -tiny)
      z = -s.p1;
    return mk<T>(T(0), T(0), z);
  }
  if (s.kind == K_CONE) {  // support_functions.cpp:228-274; p0 = radius, p1 = halfLength
    const T inflate = T(1) + T(1e-10);
    const T h = s.p1, r = s.p0;
    if (habs(dir.x) <= tiny && habs(dir.y) <= tiny) return mk<T>(T(0), T(0), (dir.z > tiny) ? h : -inflate * h);
    T zdist = dir.x * dir.x + dir.y * dir.y;
    const T len = hsqrt(zdist + dir.z * dir.z);
    zdist = hsqrt(zdist);
    const T sin_a = r / hsqrt(r * r + T(4) * h * h);
    if (dir.z > T(0) && dir.z > len * sin_a) return mk<T>(T(0), T(0), h);
    const T rad = r / zdist;
    return mk<T>(rad * dir.x, rad * dir.y, -h);
  }
  if (s.kind == K_CYLINDER) {  // support_functions.cpp:280-317
    const T inflate = T(1) + T(1e-10);
    T half_h = s.p1, r = s.p0;
    const bool aligned = habs(dir.x) <= tiny && habs(dir.y) <= tiny;
    if (aligned) half_h *= inflate;
    T z;
    if (dir.z > tiny) {
      z = half_h;
    } else if (dir.z < -tiny) {
      z = -half_h;
    } else {
      z = T(0);
      r *= inflate;
    }
    if (aligned) return mk<T>(T(0), T(0), z);
    const T n2 = dir.x * dir.x + dir.y * dir.y;
    T nx = dir.x, ny = dir.y;
    if (n2 > T(0)) {
      const T n = hsqrt(n2);
      nx = dir.x / n;
      ny = dir.y / n;
    }
    return mk<T>(nx * r, ny * r, z);
  }
  return mk<T>(T(0), T(0), T(0));  // sphere: a point, radius is swept
}

// radius folded into MinkowskiDiff::swept_sphere_radius[i]
template <typename T> HFCL_HD T swept_radius(const DShape<T>& s) {
  return s.ssr + ((s.kind == K_SPHERE || s.kind == K_CAPSULE) ? s.p0 : T(0));
}

// ---------------------------------------------------------------------------------------
// Closed forms.  All return the signed distance and fill world-frame p1, p2, normal (o1->o2).
// ---------------------------------------------------------------------------------------
template <typename T>
HFCL_HD T sphere_sphere(const DShape<T>& s1, const Pose<T>& tf1, const DShape<T>& s2, const Pose<T>& tf2, V3<T>& p1,
                        V3<T>& p2, V3<T>& normal) {
  const T r1 = s1.p0 + s1.ssr, r2 = s2.p0 + s2.ssr;
  const V3<T> c1c2 = tf2.t - tf1.t;
  const T cdist = norm(c1c2);
  V3<T> unit = mk<T>(T(1), T(0), T(0));
  if (cdist > Lim<T>::eps()) unit = c1c2 / cdist;
  normal = unit;
  p1 = tf1.t + r1 * unit;
  p2 = tf2.t - r2 * unit;
  return cdist - r1 - r2;
}

template <typename T>
HFCL_HD T sphere_capsule(const DShape<T>& s1, const Pose<T>& tf1, const DShape<T>& s2, const Pose<T>& tf2, V3<T>& p1,
                         V3<T>& p2, V3<T>& normal) {
  const V3<T> pos1 = xform(tf2, mk<T>(T(0), T(0), s2.p1));
  const V3<T> pos2 = xform(tf2, mk<T>(T(0), T(0), -s2.p1));
  const V3<T> s_c = tf1.t;
  V3<T> seg;
  {
    const V3<T> v = pos2 - pos1, w = s_c - pos1;
    const T c1 = dot(w, v), c2 = dot(v, v);
    if (c1 <= T(0))
      seg = pos1;
    else if (c2 <= c1)
      seg = pos2;
    else
      seg = pos1 + v * (c1 / c2);
  }
  normal = seg - s_c;
  const T nrm = norm(normal);
  const T r1 = s1.p0 + s1.ssr, r2 = s2.p0 + s2.ssr;
  if (nrm > Lim<T>::eps())
    normal = normalized(normal);
  else
    normal = mk<T>(T(1), T(0), T(0));
  p1 = s_c + normal * r1;
  p2 = seg - normal * r2;
  return nrm - r1 - r2;
}

template <typename T>
HFCL_HD V3<T> clamped_linear(const V3<T>& a, T s_n, T s_d, const V3<T>& d) {
  if (s_n <= T(0)) return a;
  if (s_n >= s_d) return a + d;
  return a + (s_n / s_d) * d;
}

template <typename T>
HFCL_HD T capsule_capsule(const DShape<T>& c1s, const Pose<T>& tf1, const DShape<T>& c2s, const Pose<T>& tf2, V3<T>& wp1,
                          V3<T>& wp2, V3<T>& normal) {
  const T EPSILON = Lim<T>::eps() * T(100);
  const V3<T> c1 = tf1.t, c2 = tf2.t;
  const T radius1 = c1s.p0 + c1s.ssr, radius2 = c2s.p0 + c2s.ssr;
  const V3<T> d1 = (T(2) * c1s.p1) * col(tf1.R, 2);
  const V3<T> d2 = (T(2) * c2s.p1) * col(tf2.R, 2);
  const V3<T> p1 = c1 - d1 / T(2);
  const V3<T> p2 = c2 - d2 / T(2);
  const V3<T> r = p1 - p2;
  const T a = dot(d1, d1), b = dot(d1, d2), c = dot(d1, r), e = dot(d2, d2), f = dot(d2, r);
  V3<T> w1, w2;
  if (a <= EPSILON) {
    w1 = p1;
    if (e <= EPSILON)
      w2 = p2;
    else
      w2 = clamped_linear(p2, f, e, d2);
  } else if (e <= EPSILON) {
    w1 = clamped_linear(p1, -c, a, d1);
    w2 = p2;
  } else {
    const T denom = hmax(a * e - b * b, T(0));
    T s, t;
    if (denom > EPSILON) {
      const T num = b * f - c * e;
      s = (num <= T(0)) ? T(0) : ((num >= denom) ? T(1) : num / denom);
      t = b * s + f;
    } else {
      s = T(0);
      t = f;
    }
    if (t <= T(0)) {
      w2 = p2;
      w1 = clamped_linear(p1, -c, a, d1);
    } else if (t >= e) {
      w1 = clamped_linear(p1, b - c, a, d1);
      w2 = p2 + d2;
    } else {
      w1 = p1 + s * d1;
      w2 = p2 + (t / e) * d2;
    }
  }
  const T distance = norm(w1 - w2) - (radius1 + radius2);
  normal = normalized(w2 - w1);
  wp1 = w1 + radius1 * normal;
  wp2 = w2 - radius2 * normal;
  return distance;
}

template <typename T>
HFCL_HD T box_sphere(const DShape<T>& b, const Pose<T>& tfb, const DShape<T>& s, const Pose<T>& tfs, V3<T>& pb,
                     V3<T>& ps, V3<T>& normal) {
  const V3<T> os = tfs.t, ob = tfb.t;
  pb = ob;
  bool outside = false;
  const V3<T> o = tmul(tfb.R, os - ob);
  int axis = -1;
  T min_d = Lim<T>::max();
  const T hs[3] = {b.p0, b.p1, b.p2};
  const T oo[3] = {o.x, o.y, o.z};
  for (int i = 0; i < 3; ++i) {
    const V3<T> ci = col(tfb.R, i);
    if (oo[i] < -hs[i]) {
      pb = pb - hs[i] * ci;
      outside = true;
    } else if (oo[i] > hs[i]) {
      pb = pb + hs[i] * ci;
      outside = true;
    } else {
      pb = pb + oo[i] * ci;
      const T facedist = hs[i] - habs(oo[i]);
      if (!outside && facedist < min_d) {
        axis = i;
        min_d = facedist;
      }
    }
  }
  normal = pb - os;
  const T pdist = norm(normal);
  T dist;
  if (outside) {
    dist = pdist - s.p0;
    normal = normal / (-pdist);
  } else {
    const V3<T> ca = col(tfb.R, axis);
    normal = (comp(o, axis) >= T(0)) ? ca : -ca;
    dist = -min_d - s.p0;
  }
  ps = os - s.p0 * normal;
  if (!outside || dist <= T(0)) pb = ps - dist * normal;
  if (b.ssr > T(0) || s.ssr > T(0)) {
    pb = pb + b.ssr * normal;
    ps = ps - s.ssr * normal;
    dist -= (b.ssr + s.ssr);
  }
  return dist;
}

// details::sphereCylinderDistance (src/narrowphase/details.h:107-209)
template <typename T>
HFCL_HD T sphere_cylinder(const DShape<T>& s1, const Pose<T>& tf1, const DShape<T>& s2, const Pose<T>& tf2, V3<T>& p1,
                          V3<T>& p2, V3<T>& normal) {
  const T eps = hsqrt(Lim<T>::eps());
  const T r1 = s1.p0, r2 = s2.p0, lz2 = s2.p1;
  const V3<T> A = xform(tf2, mk<T>(T(0), T(0), -lz2)), B = xform(tf2, mk<T>(T(0), T(0), lz2));
  const V3<T> S = tf1.t;
  const V3<T> u = mk<T>(tf2.R.r0.z, tf2.R.r1.z, tf2.R.r2.z);  // column 2
  const T s = dot(u, S - A);
  const V3<T> P = A + u * s;
  const V3<T> PS = S - P;
  const T dPS = norm(PS);
  V3<T> v = mk<T>(T(0), T(0), T(0));
  if (dPS > eps) v = PS * (T(1) / dPS);
  T dist;
  const bool below = s <= T(0), inside = !below && s <= lz2 * T(2);
  if (inside) {
    normal = -v;
    dist = dPS - r1 - r2;
    p2 = P + v * r2;
    p1 = S - v * r1;
  } else {
    const V3<T> C = below ? A : B;            // centre of the nearer cap
    const V3<T> un = below ? u : -u;          // normal when the cap's disc is closest
    if (dPS <= r2) {
      dist = (below ? -s : s - lz2 * T(2)) - r1;
      p1 = S + un * r1;
      p2 = C + v * dPS;
      normal = un;
    } else {  // the cap's rim is closest
      p2 = C + v * r2;
      const V3<T> Sp2 = p2 - S;
      const T dSp2 = norm(Sp2);
      if (dSp2 > eps) {
        normal = Sp2 * (T(1) / dSp2);
        p1 = S + normal * r1;
        dist = dSp2 - r1;
      } else {
        normal = normalized(p2 - (A + B) * T(.5));
        dist = -r1;
        p1 = S + normal * r1;
      }
    }
  }
  if (s1.ssr > T(0) || s2.ssr > T(0)) {
    p1 = p1 + normal * s1.ssr;
    p2 = p2 - normal * s2.ssr;
    dist -= (s1.ssr + s2.ssr);
  }
  return dist;
}

// ---------------------------------------------------------------------------------------
// Plane / Halfspace rows (src/narrowphase/details.h:347-428,509-691; src/distance/*_halfspace.cpp,
// *_plane.cpp).  getSupport<WithSweptSphere> of the other shape decides everything.
// ---------------------------------------------------------------------------------------
template <typename T>
HFCL_HD V3<T> support_with_swept_sphere(const DShape<T>& s, const T* verts, const V3<T>& dir) {
  const V3<T> u = normalized(dir);
  if (s.kind == K_SPHERE) return u * (s.p0 + s.ssr);
  V3<T> sup;
  if (s.kind == K_CONVEX) {  // getShapeSupportLinear, first maximum (support_functions.cpp:400-421)
    const T* v = verts + 3 * size_t(s.vertex_offset);
    uint32_t best = 0;
    T bd = v[0] * dir.x + v[1] * dir.y + v[2] * dir.z;
    for (uint32_t i = 1; i < s.num_points; ++i) {
      const T d = v[3 * i] * dir.x + v[3 * i + 1] * dir.y + v[3 * i + 2] * dir.z;
      if (d > bd) {
        bd = d;
        best = i;
      }
    }
    sup = mk<T>(v[3 * best], v[3 * best + 1], v[3 * best + 2]);
  } else if (s.kind == K_TRIANGLE) {  // support_functions.cpp:110-134
    const T* v = verts + 3 * size_t(s.vertex_offset);
    const V3<T> a = mk<T>(v[0], v[1], v[2]), b = mk<T>(v[3], v[4], v[5]), c = mk<T>(v[6], v[7], v[8]);
    const T da = dot(dir, a), db = dot(dir, b), dc = dot(dir, c);
    if (da > db)
      sup = (dc > da) ? c : a;
    else
      sup = (dc > db) ? c : b;
  } else {
    sup = prim_support(s, dir);
  }
  return sup + u * (s.kind == K_CAPSULE ? s.p0 + s.ssr : s.ssr);
}

template <typename T>
struct PlaneEq {  // Plane / Halfspace in the world frame (transform(), geometric_shapes_utility.cpp:249-277)
  V3<T> n;
  T d, ssr;
  HFCL_HD T signed_distance(const V3<T>& p) const { return dot(n, p) - (d + ssr); }  // geometric_shapes.h:913-915
};
template <typename T>
HFCL_HD PlaneEq<T> world_plane(const DShape<T>& h, const Pose<T>& tf) {
  PlaneEq<T> w;
  w.n = mul(tf.R, mk<T>(h.p0, h.p1, h.p2));
  w.d = h.p3 + dot(w.n, tf.t);
  w.ssr = h.ssr;
  return w;
}

// flat (Plane or Halfspace) vs a solid shape; p1 on the flat, p2 on the shape, normal = the flat's normal
template <typename T>
HFCL_HD T flat_solid_distance(const DShape<T>& f, const Pose<T>& tf1, const DShape<T>& s, const T* verts, const Pose<T>& tf2,
                              V3<T>& p1, V3<T>& p2, V3<T>& normal) {
  const PlaneEq<T> h0 = world_plane(f, tf1);
  const V3<T> a = xform(tf2, support_with_swept_sphere(s, verts, -tmul(tf2.R, h0.n)));
  const T dist1 = h0.signed_distance(a);
  if (f.kind == K_PLANE) {  // second halfspace of the plane: (-n, -d)
    PlaneEq<T> h1 = h0;
    h1.n = -h0.n;
    h1.d = -h0.d;
    const V3<T> b = xform(tf2, support_with_swept_sphere(s, verts, -tmul(tf2.R, h1.n)));
    const T dist2 = h1.signed_distance(b);
    if (!(dist1 >= dist2)) {
      p2 = b;
      p1 = p2 - h1.n * dist2;
      normal = h1.n;
      return dist2;
    }
  }
  p2 = a;
  p1 = p2 - h0.n * dist1;
  normal = h0.n;
  return dist1;
}

// flat vs a triangle given by three points in the frame of tf2 (mesh triangles: src/distance/triangle_halfspace.cpp,
// triangle_plane.cpp); p1 on the flat, p2 on the triangle, normal = the flat's normal
template <typename T>
HFCL_HD T flat_triangle_distance(const DShape<T>& f, const Pose<T>& tf1, const V3<T>& a, const V3<T>& b, const V3<T>& c,
                                 const Pose<T>& tf2, V3<T>& p1, V3<T>& p2, V3<T>& normal) {
  auto sup = [&](const V3<T>& dir) {  // getSupport<WithSweptSphere>(TriangleP): swept-sphere radius 0
    const T da = dot(dir, a), db = dot(dir, b), dc = dot(dir, c);
    V3<T> s;
    if (da > db)
      s = (dc > da) ? c : a;
    else
      s = (dc > db) ? c : b;
    return s + normalized(dir) * T(0);
  };
  const PlaneEq<T> h0 = world_plane(f, tf1);
  const V3<T> pa = xform(tf2, sup(-tmul(tf2.R, h0.n)));
  const T dist1 = h0.signed_distance(pa);
  if (f.kind == K_PLANE) {
    PlaneEq<T> h1 = h0;
    h1.n = -h0.n;
    h1.d = -h0.d;
    const V3<T> pb = xform(tf2, sup(-tmul(tf2.R, h1.n)));
    const T dist2 = h1.signed_distance(pb);
    if (!(dist1 >= dist2)) {
      p2 = pb;
      p1 = p2 - h1.n * dist2;
      normal = h1.n;
      return dist2;
    }
  }
  p2 = pa;
  p1 = p2 - h0.n * dist1;
  normal = h0.n;
  return dist1;
}

// flat vs flat: halfspace-halfspace :509-568, halfspace-plane :585-628, plane-plane :646-691
template <typename T>
HFCL_HD T flat_flat_distance(const DShape<T>& s1, const Pose<T>& tf1, const DShape<T>& s2, const Pose<T>& tf2, V3<T>& p1,
                             V3<T>& p2, V3<T>& normal) {
  const PlaneEq<T> a = world_plane(s1, tf1), b = world_plane(s2, tf2);
  const V3<T> dir = cross(a.n, b.n);
  const T dsq = sqnorm(dir);
  T distance;
  if (dsq < Lim<T>::eps()) {  // parallel
    p1 = a.n * a.d;
    p2 = b.n * b.d;
    normal = a.n;
    const bool same = dot(a.n, b.n) > T(0);
    if (s1.kind == K_PLANE) {  // plane-plane
      distance = norm(p1 - p2);
      if (distance > Lim<T>::tiny()) normal = normalized(p2 - p1);
    } else if (s2.kind == K_PLANE) {  // halfspace-plane
      distance = same ? (b.d - a.d) : -(a.d + b.d);
    } else if (!same) {  // opposite halfspaces
      distance = -(a.d + b.d);
    } else {  // nested halfspaces: infinite penetration
      distance = -Lim<T>::max();
      if (a.d <= b.d) {
        p1 = normal * distance;
      } else {
        normal = -a.n;
        p2 = -(normal * distance);
      }
    }
  } else {  // crossing: infinite penetration, witness = origin of the intersection line
    distance = -Lim<T>::max();
    normal = dir;
    p1 = p2 = cross(b.n * a.d - a.n * b.d, dir) / dsq;
  }
  if (s1.ssr > T(0) || s2.ssr > T(0)) {
    p1 = p1 + normal * s1.ssr;
    p2 = p2 - normal * s2.ssr;
    distance -= (s1.ssr + s2.ssr);
  }
  return distance;
}

// Dispatch of the CLS_CLOSED bucket incl. the operand swaps of sphere_capsule.cpp:60-71 and
// box_sphere.cpp:62-75.
template <typename T>
HFCL_HD T closed_form_distance(const DShape<T>& s1, const Pose<T>& tf1, const DShape<T>& s2, const Pose<T>& tf2,
                               const T* verts, V3<T>& p1, V3<T>& p2, V3<T>& n) {
  const bool f1 = kind_is_flat(s1.kind), f2 = kind_is_flat(s2.kind);
  if (f1 && f2) {
    if (s1.kind == K_PLANE && s2.kind == K_HALFSPACE) {  // halfspace_plane.cpp:57-68
      const T d = flat_flat_distance(s2, tf2, s1, tf1, p2, p1, n);
      n = -n;
      return d;
    }
    return flat_flat_distance(s1, tf1, s2, tf2, p1, p2, n);
  }
  if (f1) return flat_solid_distance(s1, tf1, s2, verts, tf2, p1, p2, n);
  if (f2) {  // e.g. box_halfspace.cpp:50-61
    const T d = flat_solid_distance(s2, tf2, s1, verts, tf1, p2, p1, n);
    n = -n;
    return d;
  }
  if (s1.kind == K_SPHERE && s2.kind == K_SPHERE) return sphere_sphere(s1, tf1, s2, tf2, p1, p2, n);
  if (s1.kind == K_SPHERE && s2.kind == K_CAPSULE) return sphere_capsule(s1, tf1, s2, tf2, p1, p2, n);
  if (s1.kind == K_CAPSULE && s2.kind == K_SPHERE) {
    const T d = sphere_capsule(s2, tf2, s1, tf1, p2, p1, n);
    n = -n;
    return d;
  }
  if (s1.kind == K_CAPSULE && s2.kind == K_CAPSULE) return capsule_capsule(s1, tf1, s2, tf2, p1, p2, n);
  if (s1.kind == K_BOX && s2.kind == K_SPHERE) return box_sphere(s1, tf1, s2, tf2, p1, p2, n);
  if (s1.kind == K_SPHERE && s2.kind == K_CYLINDER) return sphere_cylinder(s1, tf1, s2, tf2, p1, p2, n);
  if (s1.kind == K_CYLINDER && s2.kind == K_SPHERE) {  // sphere_cylinder.cpp:63-74
    const T d = sphere_cylinder(s2, tf2, s1, tf1, p2, p1, n);
    n = -n;
    return d;
  }
  // sphere - box
  const T d = box_sphere(s2, tf2, s1, tf1, p2, p1, n);
  n = -n;
  return d;
}

}  // namespace hfcl
