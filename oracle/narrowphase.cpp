// ORACLE -- TEST INFRASTRUCTURE ONLY.  Not part of the product path (see vec3.hpp).
#include "narrowphase.hpp"
#include <algorithm>
#include <cmath>
#include <cstring>

namespace orc {

static const double kMax = std::numeric_limits<double>::max();

void query_request_defaults(hfcl_query_request* q) {  // collision_data.h:205-222, narrowphase_defaults.h:47-62
  q->gjk_initial_guess = HFCL_GUESS_DEFAULT;
  q->gjk_variant = HFCL_GJK_DEFAULT;
  q->gjk_convergence_criterion = HFCL_CRIT_DEFAULT;
  q->gjk_convergence_criterion_type = HFCL_CRIT_RELATIVE;
  q->gjk_max_iterations = 128;
  q->epa_max_iterations = 64;
  q->gjk_tolerance = 1e-6;
  q->epa_tolerance = 1e-6;
  q->collision_distance_threshold = 1e-12;
  q->cached_gjk_guess[0] = 1;
  q->cached_gjk_guess[1] = 0;
  q->cached_gjk_guess[2] = 0;
  q->cached_support_func_guess[0] = q->cached_support_func_guess[1] = 0;
}
void distance_request_defaults(hfcl_distance_request* r) {  // collision_data.h:987-1031
  std::memset(r, 0, sizeof(*r));
  query_request_defaults(&r->q);
  r->enable_nearest_points = 1;
  r->enable_signed_distance = 1;
  r->rel_err = 0;
  r->abs_err = 0;
}
void collision_request_defaults(hfcl_collision_request* r) {  // collision_data.h:312-366
  std::memset(r, 0, sizeof(*r));
  query_request_defaults(&r->q);
  r->num_max_contacts = 1;
  r->enable_contact = 1;
  r->security_margin = 0;
  r->break_distance = 1e-3;
  r->distance_upper_bound = kMax;
}

void GJKSolver::set_query(const hfcl_query_request& q) {
  gjk_initial_guess = q.gjk_initial_guess;
  cached_guess = V3(1, 0, 0);  // GJKSolver(request) ctor, narrowphase.h:152-153
  support_func_cached_guess[0] = support_func_cached_guess[1] = 0;
  if (gjk_initial_guess == HFCL_GUESS_CACHED) {
    cached_guess = V3(q.cached_gjk_guess[0], q.cached_gjk_guess[1], q.cached_gjk_guess[2]);
    support_func_cached_guess[0] = q.cached_support_func_guess[0];
    support_func_cached_guess[1] = q.cached_support_func_guess[1];
  }
  gjk_max_iterations = q.gjk_max_iterations;
  gjk_tolerance = q.gjk_tolerance;
  gjk_variant = q.gjk_variant;
  gjk_convergence_criterion = q.gjk_convergence_criterion;
  gjk_convergence_criterion_type = q.gjk_convergence_criterion_type;
  epa_max_iterations = q.epa_max_iterations;
  epa_tolerance = q.epa_tolerance;
}
void GJKSolver::set(const hfcl_distance_request& r) {
  set_query(r.q);
  distance_upper_bound = kMax;  // narrowphase.h:175
}
void GJKSolver::set(const hfcl_collision_request& r) {
  set_query(r.q);
  distance_upper_bound = std::max(0., std::max(r.distance_upper_bound, r.security_margin));  // :228-229
}

static V3 aabb_local_center(const Shape& s) {
  if (s.kind != K_CONVEX && s.kind != K_TRIANGLE) return V3(0, 0, 0);
  V3 mn(s.verts[0], s.verts[1], s.verts[2]), mx = mn;
  for (int i = 1; i < s.nverts; ++i)
    for (int k = 0; k < 3; ++k) {
      mn[k] = std::min(mn[k], s.verts[3 * i + k]);
      mx[k] = std::max(mx[k], s.verts[3 * i + k]);
    }
  return (mn + mx) * 0.5;
}

double GJKSolver::run_gjk_epa(const Shape& s1, const Tf& tf1, const Shape& s2, const Tf& tf2,
                              bool compute_penetration, V3& p1, V3& p2, V3& normal,
                              bool relative_precomputed) const {
  MinkowskiDiff md;
  if (relative_precomputed)
    md.set(&s1, &s2);
  else
    md.set(&s1, &s2, tf1, tf2);
  GJK gjk(gjk_max_iterations, gjk_tolerance);
  gjk.distance_upper_bound = distance_upper_bound;
  gjk.gjk_variant = gjk_variant;
  gjk.convergence_criterion = gjk_convergence_criterion;
  gjk.convergence_criterion_type = gjk_convergence_criterion_type;
  stats = SolverStats();

  // getGJKInitialGuess, narrowphase.h:353-391
  V3 guess(1, 0, 0);
  int support_hint[2] = {support_func_cached_guess[0], support_func_cached_guess[1]};
  switch (gjk_initial_guess) {
    case HFCL_GUESS_DEFAULT:
      guess = V3(1, 0, 0);
      break;
    case HFCL_GUESS_CACHED:
      guess = cached_guess;
      break;
    case HFCL_GUESS_BOUNDING_VOLUME:
      guess = aabb_local_center(s1) - (md.oR1 * aabb_local_center(s2) + md.ot1);
      break;
  }

  out_cached_guess = cached_guess;
  out_support_guess[0] = support_func_cached_guess[0];
  out_support_guess[1] = support_func_cached_guess[1];

  gjk.evaluate(md, guess, support_hint);
  stats.gjk_status = gjk.status;
  stats.gjk_iterations = unsigned(gjk.iterations);

  double distance = 0;
  const V3 nan = nan3();

  auto gjk_extract = [&]() {  // GJKExtractWitnessPointsAndNormal, narrowphase.h:610-636
    out_cached_guess = gjk.ray;
    out_support_guess[0] = gjk.support_hint[0];
    out_support_guess[1] = gjk.support_hint[1];
    distance = gjk.distance;
    gjk.get_witness_points_and_normal(md, p1, p2, normal);
    V3 p = tf1.transform(0.5 * (p1 + p2));
    normal = tf1.R * normal;
    p1 = p - 0.5 * distance * normal;
    p2 = p + 0.5 * distance * normal;
  };

  switch (gjk.status) {
    case GJK::DidNotRun:
      distance = -kMax;
      p1 = p2 = normal = nan;
      break;
    case GJK::Failed:
      gjk_extract();
      break;
    case GJK::NoCollisionEarlyStopped:  // :589-608
      out_cached_guess = gjk.ray;
      out_support_guess[0] = gjk.support_hint[0];
      out_support_guess[1] = gjk.support_hint[1];
      distance = gjk.distance;
      p1 = p2 = normal = nan;
      break;
    case GJK::NoCollision:
    case GJK::CollisionWithPenetrationInformation:
      gjk_extract();
      break;
    case GJK::Collision:
      if (!compute_penetration) {  // :638-656
        out_support_guess[0] = gjk.support_hint[0];
        out_support_guess[1] = gjk.support_hint[1];
        distance = gjk.distance;
        p1 = p2 = normal = nan;
      } else {
        EPA epa(epa_max_iterations, epa_tolerance);
        epa.evaluate(gjk, -guess);
        stats.epa_status = epa.status;
        stats.epa_iterations = unsigned(epa.iterations);
        switch (epa.status) {
          case EPA::DidNotRun:
          case EPA::FallBack:  // EPAFailedExtractWitnessPointsAndNormal :713-723
            out_cached_guess = V3(1, 0, 0);
            out_support_guess[0] = out_support_guess[1] = 0;
            distance = -kMax;
            p1 = p2 = normal = nan;
            break;
          default: {  // EPAExtractWitnessPointsAndNormal :658-711
            out_cached_guess = -(epa.depth * epa.normal);
            out_support_guess[0] = epa.support_hint[0];
            out_support_guess[1] = epa.support_hint[1];
            distance = std::min(0., -epa.depth);
            epa.get_witness_points_and_normal(md, p1, p2, normal);
            V3 p = tf1.transform(0.5 * (p1 + p2));
            normal = tf1.R * normal;
            p1 = p - 0.5 * distance * normal;
            p2 = p + 0.5 * distance * normal;
          }
        }
      }
      break;
  }
  return distance;
}

// ---------------------------------------------------------------------------------------
// closed forms
// ---------------------------------------------------------------------------------------
// details::sphereSphereDistance, src/narrowphase/details.h:215-232
static double sphere_sphere(const Shape& s1, const Tf& tf1, const Shape& s2, const Tf& tf2, V3& p1, V3& p2,
                            V3& normal) {
  const V3& center1 = tf1.T;
  const V3& center2 = tf2.T;
  double r1 = s1.p[0] + s1.ssr;
  double r2 = s2.p[0] + s2.ssr;
  V3 c1c2 = center2 - center1;
  double cdist = norm(c1c2);
  V3 unit(1, 0, 0);
  if (cdist > std::numeric_limits<double>::epsilon()) unit = c1c2 / cdist;
  double dist = cdist - r1 - r2;
  normal = unit;
  p1 = center1 + r1 * unit;
  p2 = center2 - r2 * unit;
  return dist;
}

// getSupport<WithSweptSphere> (support_functions.cpp:47-63 + the WithSweptSphere tails of :110-421)
static V3 support_with_swept_sphere(const Shape& s, const V3& dir) {
  int hint = 0;
  V3 sup = shape_support(s, dir, hint);
  if (s.kind == K_SPHERE) return normalized(dir) * (s.p[0] + s.ssr);
  if (s.kind == K_CAPSULE) return sup + normalized(dir) * (s.p[0] + s.ssr);
  return sup + normalized(dir) * s.ssr;
}

struct PlaneEq {  // Halfspace / Plane expressed in the world frame: transform() geometric_shapes_utility.cpp:249-277
  V3 n;
  double d, ssr;
  double signed_distance(const V3& p) const { return dot(n, p) - (d + ssr); }  // Halfspace::signedDistance :913-915
};
static PlaneEq world_plane(const Shape& h, const Tf& tf) {
  PlaneEq w;
  w.n = tf.R * V3(h.p[0], h.p[1], h.p[2]);
  w.d = h.p[3] + dot(w.n, tf.T);
  w.ssr = h.ssr;
  return w;
}

// details::halfspaceDistance :347-375 (p1 on the halfspace, p2 on the shape, normal = halfspace normal)
static double halfspace_distance(const Shape& h, const Tf& tf1, const Shape& s, const Tf& tf2, V3& p1, V3& p2, V3& normal) {
  const PlaneEq nh = world_plane(h, tf1);
  const V3 n_2 = transpose(tf2.R) * nh.n;
  p2 = tf2.transform(support_with_swept_sphere(s, -n_2));
  const double dist = nh.signed_distance(p2);
  p1 = p2 - nh.n * dist;
  normal = nh.n;
  return dist;
}

// details::planeDistance :381-428: the plane as two opposite halfspaces, the farther one decides
static double plane_distance(const Shape& pl, const Tf& tf1, const Shape& s, const Tf& tf2, V3& p1, V3& p2, V3& normal) {
  PlaneEq h0 = world_plane(pl, tf1), h1 = h0;
  h1.n = -h0.n;
  h1.d = -h0.d;
  const V3 n_h1 = transpose(tf2.R) * h0.n, n_h2 = transpose(tf2.R) * h1.n;
  const V3 a = tf2.transform(support_with_swept_sphere(s, -n_h1));
  const V3 b = tf2.transform(support_with_swept_sphere(s, -n_h2));
  const double dist1 = h0.signed_distance(a), dist2 = h1.signed_distance(b);
  if (dist1 >= dist2) {
    p2 = a;
    p1 = p2 - h0.n * dist1;
    normal = h0.n;
    return dist1;
  }
  p2 = b;
  p1 = p2 - h1.n * dist2;
  normal = h1.n;
  return dist2;
}

static void apply_ssr(const Shape& s1, const Shape& s2, V3& p1, V3& p2, const V3& normal, double& distance) {
  if (s1.ssr > 0 || s2.ssr > 0) {
    p1 = p1 + normal * s1.ssr;
    p2 = p2 - normal * s2.ssr;
    distance -= (s1.ssr + s2.ssr);
  }
}
static V3 intersection_line_origin(const PlaneEq& a, const PlaneEq& b, const V3& dir, double dir_sq_norm) {
  return cross(b.n * a.d - a.n * b.d, dir) / dir_sq_norm;
}

// details::halfspaceHalfspaceDistance :509-568
static double halfspace_halfspace(const Shape& s1, const Tf& tf1, const Shape& s2, const Tf& tf2, V3& p1, V3& p2, V3& normal) {
  const PlaneEq a = world_plane(s1, tf1), b = world_plane(s2, tf2);
  double distance;
  const V3 dir = cross(a.n, b.n);
  const double dir_sq_norm = sqnorm(dir);
  if (dir_sq_norm < std::numeric_limits<double>::epsilon()) {
    if (dot(a.n, b.n) > 0) {
      distance = -std::numeric_limits<double>::max();
      if (a.d <= b.d) {
        normal = a.n;
        p1 = normal * distance;
        p2 = b.n * b.d;
      } else {
        normal = -a.n;
        p1 = a.n * a.d;
        p2 = -(normal * distance);
      }
    } else {
      distance = -(a.d + b.d);
      normal = a.n;
      p1 = a.n * a.d;
      p2 = b.n * b.d;
    }
  } else {
    distance = -std::numeric_limits<double>::max();
    normal = dir;
    p1 = p2 = intersection_line_origin(a, b, dir, dir_sq_norm);
  }
  apply_ssr(s1, s2, p1, p2, normal, distance);
  return distance;
}

// details::halfspacePlaneDistance :585-628
static double halfspace_plane(const Shape& s1, const Tf& tf1, const Shape& s2, const Tf& tf2, V3& p1, V3& p2, V3& normal) {
  const PlaneEq a = world_plane(s1, tf1), b = world_plane(s2, tf2);
  double distance;
  const V3 dir = cross(a.n, b.n);
  const double dir_sq_norm = sqnorm(dir);
  if (dir_sq_norm < std::numeric_limits<double>::epsilon()) {
    normal = a.n;
    distance = dot(a.n, b.n) > 0 ? (b.d - a.d) : -(a.d + b.d);
    p1 = a.n * a.d;
    p2 = b.n * b.d;
  } else {
    distance = -std::numeric_limits<double>::max();
    normal = dir;
    p1 = p2 = intersection_line_origin(a, b, dir, dir_sq_norm);
  }
  apply_ssr(s1, s2, p1, p2, normal, distance);
  return distance;
}

// details::planePlaneDistance :646-691
static double plane_plane(const Shape& s1, const Tf& tf1, const Shape& s2, const Tf& tf2, V3& p1, V3& p2, V3& normal) {
  const PlaneEq a = world_plane(s1, tf1), b = world_plane(s2, tf2);
  double distance;
  const V3 dir = cross(a.n, b.n);
  const double dir_sq_norm = sqnorm(dir);
  if (dir_sq_norm < std::numeric_limits<double>::epsilon()) {
    p1 = a.n * a.d;
    p2 = b.n * b.d;
    distance = norm(p1 - p2);
    if (distance > kDummyPrecision)
      normal = normalized(p2 - p1);
    else
      normal = a.n;
  } else {
    distance = -std::numeric_limits<double>::max();
    normal = dir;
    p1 = p2 = intersection_line_origin(a, b, dir, dir_sq_norm);
  }
  apply_ssr(s1, s2, p1, p2, normal, distance);
  return distance;
}

// details::segmentSqrDistance :235-255, projectInTriangle :258-279, sphereTriangleDistance :286-340
static double segment_sqr_distance(const V3& from, const V3& to, const V3& p, V3& nearest) {
  V3 diff = p - from;
  const V3 v = to - from;
  double t = dot(v, diff);
  if (t > 0) {
    const double dotVV = sqnorm(v);
    if (t < dotVV) {
      t /= dotVV;
      diff = diff - v * t;
    } else {
      t = 1;
      diff = diff - v;
    }
  } else
    t = 0;
  nearest = from + v * t;
  return sqnorm(diff);
}
static bool project_in_triangle(const V3& p1, const V3& p2, const V3& p3, const V3& normal, const V3& p) {
  const V3 e1 = p2 - p1, e2 = p3 - p2, e3 = p1 - p3;
  const double r1 = dot(cross(e1, normal), p - p1), r2 = dot(cross(e2, normal), p - p2), r3 = dot(cross(e3, normal), p - p3);
  return (r1 > 0 && r2 > 0 && r3 > 0) || (r1 <= 0 && r2 <= 0 && r3 <= 0);
}
static double sphere_triangle(const Shape& s, const Tf& tf1, const Shape& tri, const Tf& tf2, V3& p1, V3& p2, V3& normal) {
  const V3 P1 = tf2.transform(V3(tri.verts[0], tri.verts[1], tri.verts[2]));
  const V3 P2 = tf2.transform(V3(tri.verts[3], tri.verts[4], tri.verts[5]));
  const V3 P3 = tf2.transform(V3(tri.verts[6], tri.verts[7], tri.verts[8]));
  V3 tri_normal = normalized(cross(P2 - P1, P3 - P1));
  const V3 center = tf1.T;
  const double radius = s.p[0] + s.ssr + tri.ssr;
  double distance_from_plane = dot(center - P1, tri_normal);
  const double nanv = std::numeric_limits<double>::quiet_NaN();
  V3 closest_point(nanv, nanv, nanv);
  double min_distance_sqr, distance_sqr;
  if (distance_from_plane < 0) {
    distance_from_plane *= -1;
    tri_normal = tri_normal * -1.0;
  }
  if (project_in_triangle(P1, P2, P3, tri_normal, center)) {
    closest_point = center - tri_normal * distance_from_plane;
    min_distance_sqr = distance_from_plane * distance_from_plane;
  } else {
    V3 nearest_on_edge;
    min_distance_sqr = segment_sqr_distance(P1, P2, center, closest_point);
    distance_sqr = segment_sqr_distance(P2, P3, center, nearest_on_edge);
    if (distance_sqr < min_distance_sqr) {
      min_distance_sqr = distance_sqr;
      closest_point = nearest_on_edge;
    }
    distance_sqr = segment_sqr_distance(P3, P1, center, nearest_on_edge);
    if (distance_sqr < min_distance_sqr) {
      min_distance_sqr = distance_sqr;
      closest_point = nearest_on_edge;
    }
  }
  normal = normalized(closest_point - center);
  p1 = center + normal * (s.p[0] + s.ssr);
  p2 = closest_point - normal * tri.ssr;
  return std::sqrt(min_distance_sqr) - radius;
}

// details::sphereCylinderDistance :107-209
static double sphere_cylinder(const Shape& s1, const Tf& tf1, const Shape& s2, const Tf& tf2, V3& p1, V3& p2,
                              V3& normal) {
  const double eps = std::sqrt(std::numeric_limits<double>::epsilon());
  const double r1 = s1.p[0], r2 = s2.p[0], lz2 = s2.p[1];
  const V3 A = tf2.transform(V3(0, 0, -lz2)), B = tf2.transform(V3(0, 0, lz2));
  const V3 S = tf1.T;
  const V3 u = tf2.R.col(2);
  const V3 AS = S - A;
  const double s = dot(u, AS);
  const V3 P = A + u * s;
  const V3 PS = S - P;
  const double dPS = norm(PS);
  V3 v(0, 0, 0);
  double dist;
  if (dPS > eps) v = PS * (1 / dPS);
  auto rim = [&](const V3& C) {  // closest point on a cylinder circle basis
    p2 = C + v * r2;
    const V3 Sp2 = p2 - S;
    const double dSp2 = norm(Sp2);
    if (dSp2 > eps) {
      normal = Sp2 * (1 / dSp2);
      p1 = S + normal * r1;
      dist = dSp2 - r1;
    } else {  // centre of the sphere on the cylinder boundary
      normal = normalized(p2 - (A + B) * .5);
      dist = -r1;
      p1 = S + normal * r1;
    }
  };
  if (s <= 0) {
    if (dPS <= r2) {
      dist = -s - r1;
      p1 = S + u * r1;
      p2 = A + v * dPS;
      normal = u;
    } else {
      rim(A);
    }
  } else if (s <= (lz2 * 2)) {
    normal = -v;
    dist = dPS - r1 - r2;
    p2 = P + v * r2;
    p1 = S - v * r1;
  } else {
    if (dPS <= r2) {
      dist = s - (lz2 * 2) - r1;
      p1 = S - u * r1;
      p2 = B + v * dPS;
      normal = -u;
    } else {
      rim(B);
    }
  }
  const double ssr1 = s1.ssr, ssr2 = s2.ssr;
  if (ssr1 > 0 || ssr2 > 0) {
    p1 = p1 + normal * ssr1;
    p2 = p2 - normal * ssr2;
    dist -= (ssr1 + ssr2);
  }
  return dist;
}

// details::lineSegmentPointClosestToPoint :52-70 + sphereCapsuleDistance :76-101
static double sphere_capsule(const Shape& s1, const Tf& tf1, const Shape& s2, const Tf& tf2, V3& p1, V3& p2,
                             V3& normal) {
  V3 pos1 = tf2.transform(V3(0., 0., s2.p[1]));
  V3 pos2 = tf2.transform(V3(0., 0., -s2.p[1]));
  V3 s_c = tf1.T;
  V3 segment_point;
  {
    V3 v = pos2 - pos1;
    V3 w = s_c - pos1;
    double c1 = dot(w, v);
    double c2 = dot(v, v);
    if (c1 <= 0)
      segment_point = pos1;
    else if (c2 <= c1)
      segment_point = pos2;
    else {
      double b = c1 / c2;
      segment_point = pos1 + v * b;
    }
  }
  normal = segment_point - s_c;
  double nrm = norm(normal);
  double r1 = s1.p[0] + s1.ssr;
  double r2 = s2.p[0] + s2.ssr;
  double dist = nrm - r1 - r2;
  if (nrm > std::numeric_limits<double>::epsilon())
    normal = normalized(normal);
  else
    normal = V3(1, 0, 0);
  p1 = s_c + normal * r1;
  p2 = segment_point - normal * r2;
  return dist;
}

// src/distance/capsule_capsule.cpp:52-167
static double clamp01(double num, double denom) {
  if (num <= 0.) return 0.;
  if (num >= denom) return 1.;
  return num / denom;
}
static void clamped_linear(V3& a_sd, const V3& a, double s_n, double s_d, const V3& d) {
  if (s_n <= 0.)
    a_sd = a;
  else if (s_n >= s_d)
    a_sd = a + d;
  else
    a_sd = a + (s_n / s_d) * d;
}
static double capsule_capsule(const Shape& c1s, const Tf& tf1, const Shape& c2s, const Tf& tf2, V3& wp1, V3& wp2,
                              V3& normal) {
  double EPSILON = std::numeric_limits<double>::epsilon() * 100;
  const V3& c1 = tf1.T;
  const V3& c2 = tf2.T;
  double halfLength1 = c1s.p[1], halfLength2 = c2s.p[1];
  double radius1 = c1s.p[0] + c1s.ssr, radius2 = c2s.p[0] + c2s.ssr;
  const V3 d1 = (2 * halfLength1) * tf1.R.col(2);
  const V3 d2 = (2 * halfLength2) * tf2.R.col(2);
  const V3 p1 = c1 - d1 / 2;
  const V3 p2 = c2 - d2 / 2;
  const V3 r = p1 - p2;
  double a = dot(d1, d1), b = dot(d1, d2), c = dot(d1, r), e = dot(d2, d2), f = dot(d2, r);
  V3 w1, w2;
  if (a <= EPSILON) {
    w1 = p1;
    if (e <= EPSILON)
      w2 = p2;
    else
      clamped_linear(w2, p2, f, e, d2);
  } else if (e <= EPSILON) {
    clamped_linear(w1, p1, -c, a, d1);
    w2 = p2;
  } else {
    double denom = std::fmax(a * e - b * b, 0);
    double s, t;
    if (denom > EPSILON) {
      s = clamp01((b * f - c * e), denom);
      t = b * s + f;
    } else {
      s = 0.;
      t = f;
    }
    if (t <= 0.0) {
      w2 = p2;
      clamped_linear(w1, p1, -c, a, d1);
    } else if (t >= e) {
      clamped_linear(w1, p1, (b - c), a, d1);
      w2 = p2 + d2;
    } else {
      w1 = p1 + s * d1;
      w2 = p2 + (t / e) * d2;
    }
  }
  double distance = norm(w1 - w2);
  distance = distance - (radius1 + radius2);
  normal = normalized(w2 - w1);
  wp1 = w1 + radius1 * normal;
  wp2 = w2 - radius2 * normal;
  return distance;
}

// details::boxSphereDistance, src/narrowphase/details.h:435-495
static double box_sphere(const Shape& b, const Tf& tfb, const Shape& s, const Tf& tfs, V3& pb, V3& ps, V3& normal) {
  const V3& os = tfs.T;
  const V3& ob = tfb.T;
  const M3& Rb = tfb.R;
  pb = ob;
  bool outside = false;
  const V3 os_in_b_frame = tmul(Rb, os - ob);
  int axis = -1;
  double min_d = kMax;
  for (int i = 0; i < 3; ++i) {
    double facedist;
    if (os_in_b_frame[i] < -b.p[i]) {
      pb -= b.p[i] * Rb.col(i);
      outside = true;
    } else if (os_in_b_frame[i] > b.p[i]) {
      pb += b.p[i] * Rb.col(i);
      outside = true;
    } else {
      pb += os_in_b_frame[i] * Rb.col(i);
      if (!outside && (facedist = b.p[i] - std::fabs(os_in_b_frame[i])) < min_d) {
        axis = i;
        min_d = facedist;
      }
    }
  }
  normal = pb - os;
  double pdist = norm(normal);
  double dist;
  if (outside) {
    dist = pdist - s.p[0];
    normal = normal / (-pdist);
  } else {
    if (os_in_b_frame[axis] >= 0)
      normal = Rb.col(axis);
    else
      normal = -Rb.col(axis);
    dist = -min_d - s.p[0];
  }
  ps = os - s.p[0] * normal;
  if (!outside || dist <= 0) pb = ps - dist * normal;
  const double ssrb = b.ssr, ssrs = s.ssr;
  if (ssrb > 0 || ssrs > 0) {
    pb += ssrb * normal;
    ps -= ssrs * normal;
    dist -= (ssrb + ssrs);
  }
  return dist;
}

// details::computePenetration, src/narrowphase/details.h:699-731
static double compute_penetration_tri(const V3& P1, const V3& P2, const V3& P3, const V3& Q1, const V3& Q2,
                                      const V3& Q3, V3& normal) {
  V3 u = cross(P2 - P1, P3 - P1);
  normal = normalized(u);
  double depth1 = dot(P1 - Q1, normal);
  double depth2 = dot(P1 - Q2, normal);
  double depth3 = dot(P1 - Q3, normal);
  return std::max(depth1, std::max(depth2, depth3));
}

// GJKSolver::shapeDistance(const S1&, tf1, const TriangleP&, tf2, ...), narrowphase.h:320-336
static double solid_triangle(const Shape& s1, const Tf& tf1, const Shape& tri, const Tf& tf2, const GJKSolver& sv,
                             bool compute_penetration, V3& p1, V3& p2, V3& normal) {
  const Tf tf_1M2 = inverse_times(tf1, tf2);
  double t[9];
  for (int k = 0; k < 3; ++k) {
    const V3 v = tf_1M2.transform(V3(tri.verts[3 * k], tri.verts[3 * k + 1], tri.verts[3 * k + 2]));
    t[3 * k] = v[0];
    t[3 * k + 1] = v[1];
    t[3 * k + 2] = v[2];
  }
  Shape moved = tri;
  moved.verts = t;
  return sv.run_gjk_epa(s1, tf1, moved, tf_1M2, compute_penetration, p1, p2, normal, /*relative_precomputed=*/true);
}

// ShapeShapeDistance<TriangleP,TriangleP>, src/distance/triangle_triangle.cpp:46-105
static double triangle_triangle(const Shape& s1, const Tf& tf1, const Shape& s2, const Tf& tf2, const GJKSolver& sv,
                                V3& p1, V3& p2, V3& normal) {
  auto vtx = [](const Shape& s, int i) { return V3(s.verts[3 * i], s.verts[3 * i + 1], s.verts[3 * i + 2]); };
  double t1w[9], t2w[9];
  for (int i = 0; i < 3; ++i) {
    V3 a = tf1.transform(vtx(s1, i)), b = tf2.transform(vtx(s2, i));
    for (int k = 0; k < 3; ++k) {
      t1w[3 * i + k] = a[k];
      t2w[3 * i + k] = b[k];
    }
  }
  Shape t1 = s1, t2 = s2;
  t1.verts = t1w;
  t2.verts = t2w;
  MinkowskiDiff md;
  md.set(&t1, &t2);  // world-frame triangles, identity relative transform (:61-63)
  // solver->gjk.reset(max_it, tol) only: the callers construct a fresh GJKSolver(request), whose
  // gjk member was initialize()d -> DefaultGJK, Default/Relative criterion, no early stop
  // (gjk.cpp:51-57); the request's variant is never forwarded on this path.
  GJK gjk(sv.gjk_max_iterations, sv.gjk_tolerance);
  V3 guess;
  if (sv.gjk_initial_guess == HFCL_GUESS_CACHED)
    guess = sv.cached_guess;
  else
    guess = (vtx(t1, 0) + vtx(t1, 1) + vtx(t1, 2) - vtx(t2, 0) - vtx(t2, 1) - vtx(t2, 2)) / 3;
  int hint[2] = {0, 0};  // uninitialised in the reference; unused by the triangle support
  GJK::Status st = gjk.evaluate(md, guess, hint);
  sv.stats = SolverStats();
  sv.stats.gjk_status = gjk.status;
  sv.stats.gjk_iterations = unsigned(gjk.iterations);
  sv.out_cached_guess = gjk.ray;
  sv.out_support_guess[0] = gjk.support_hint[0];
  sv.out_support_guess[1] = gjk.support_hint[1];
  gjk.get_witness_points_and_normal(md, p1, p2, normal);
  double distance = gjk.distance;
  if (st == GJK::Collision) {
    double depth =
        compute_penetration_tri(vtx(t1, 0), vtx(t1, 1), vtx(t1, 2), vtx(t2, 0), vtx(t2, 1), vtx(t2, 2), normal);
    distance = -depth;
  }
  return distance;
}

bool shape_shape_distance(const Shape& s1, const Tf& tf1, const Shape& s2, const Tf& tf2, const GJKSolver& solver,
                          bool compute_signed_distance, double& dist, V3& p1, V3& p2, V3& normal) {
  auto is_gjk_kind = [](int k) {
    return k == K_BOX || k == K_SPHERE || k == K_CAPSULE || k == K_ELLIPSOID || k == K_CONVEX || k == K_CONE ||
           k == K_CYLINDER;
  };
  solver.stats = SolverStats();
  // closed-form table, shape_shape_func.h:185-211 + 281-306
  if (s1.kind == K_SPHERE && s2.kind == K_SPHERE) {
    dist = sphere_sphere(s1, tf1, s2, tf2, p1, p2, normal);
    return true;
  }
  if (s1.kind == K_SPHERE && s2.kind == K_CAPSULE) {
    dist = sphere_capsule(s1, tf1, s2, tf2, p1, p2, normal);
    return true;
  }
  if (s1.kind == K_CAPSULE && s2.kind == K_SPHERE) {  // sphere_capsule.cpp:60-71
    dist = sphere_capsule(s2, tf2, s1, tf1, p2, p1, normal);
    normal = -normal;
    return true;
  }
  if (s1.kind == K_CAPSULE && s2.kind == K_CAPSULE) {
    dist = capsule_capsule(s1, tf1, s2, tf2, p1, p2, normal);
    return true;
  }
  if (s1.kind == K_BOX && s2.kind == K_SPHERE) {
    dist = box_sphere(s1, tf1, s2, tf2, p1, p2, normal);
    return true;
  }
  if (s1.kind == K_SPHERE && s2.kind == K_BOX) {  // box_sphere.cpp:62-75
    dist = box_sphere(s2, tf2, s1, tf1, p2, p1, normal);
    normal = -normal;
    return true;
  }
  if (s1.kind == K_SPHERE && s2.kind == K_CYLINDER) {
    dist = sphere_cylinder(s1, tf1, s2, tf2, p1, p2, normal);
    return true;
  }
  if (s1.kind == K_CYLINDER && s2.kind == K_SPHERE) {  // sphere_cylinder.cpp:63-74
    dist = sphere_cylinder(s2, tf2, s1, tf1, p2, p1, normal);
    normal = -normal;
    return true;
  }
  {  // Halfspace / Plane rows of the table: src/distance/*_halfspace.cpp, *_plane.cpp, halfspace_*.cpp, plane_plane.cpp
    const bool h1 = s1.kind == K_HALFSPACE, h2 = s2.kind == K_HALFSPACE, q1 = s1.kind == K_PLANE, q2 = s2.kind == K_PLANE;
    auto solid = [&](int k) { return is_gjk_kind(k) || k == K_TRIANGLE; };
    if (h1 && h2) {
      dist = halfspace_halfspace(s1, tf1, s2, tf2, p1, p2, normal);
      return true;
    }
    if (h1 && q2) {
      dist = halfspace_plane(s1, tf1, s2, tf2, p1, p2, normal);
      return true;
    }
    if (q1 && h2) {  // halfspace_plane.cpp:57-68
      dist = halfspace_plane(s2, tf2, s1, tf1, p2, p1, normal);
      normal = -normal;
      return true;
    }
    if (q1 && q2) {
      dist = plane_plane(s1, tf1, s2, tf2, p1, p2, normal);
      return true;
    }
    if ((h1 || q1) && solid(s2.kind)) {
      dist = h1 ? halfspace_distance(s1, tf1, s2, tf2, p1, p2, normal) : plane_distance(s1, tf1, s2, tf2, p1, p2, normal);
      return true;
    }
    if (solid(s1.kind) && (h2 || q2)) {  // e.g. box_halfspace.cpp:50-61
      dist = h2 ? halfspace_distance(s2, tf2, s1, tf1, p2, p1, normal) : plane_distance(s2, tf2, s1, tf1, p2, p1, normal);
      normal = -normal;
      return true;
    }
  }
  if (s1.kind == K_SPHERE && s2.kind == K_TRIANGLE) {
    dist = sphere_triangle(s1, tf1, s2, tf2, p1, p2, normal);
    return true;
  }
  if (s1.kind == K_TRIANGLE && s2.kind == K_SPHERE) {  // triangle_sphere.cpp:45-56
    dist = sphere_triangle(s2, tf2, s1, tf1, p2, p1, normal);
    normal = -normal;
    return true;
  }
  if (s1.kind == K_TRIANGLE && s2.kind == K_TRIANGLE) {
    dist = triangle_triangle(s1, tf1, s2, tf2, solver, p1, p2, normal);
    return true;
  }
  // TriangleP against the other solids: generic GJK/EPA (the "1" entries of the triangle column,
  // shape_shape_func.h:185-211) through GJKSolver::shapeDistance's TriangleP overloads (narrowphase.h:320-348):
  // the solid is always shape 0 of the Minkowski difference, the triangle is moved into the solid's frame
  // (relative transform precomputed = identity), and (TriangleP, S) is the swapped call with the points
  // exchanged and the normal negated.  TriangleP is not a ConvexBase: no support-direction normalisation.
  if (is_gjk_kind(s1.kind) && s2.kind == K_TRIANGLE) {
    dist = solid_triangle(s1, tf1, s2, tf2, solver, compute_signed_distance, p1, p2, normal);
    return true;
  }
  if (s1.kind == K_TRIANGLE && is_gjk_kind(s2.kind)) {  // narrowphase.h:339-348
    dist = solid_triangle(s2, tf2, s1, tf1, solver, compute_signed_distance, p2, p1, normal);
    normal = -normal;
    return true;
  }
  if (is_gjk_kind(s1.kind) && is_gjk_kind(s2.kind)) {
    dist = solver.run_gjk_epa(s1, tf1, s2, tf2, compute_signed_distance, p1, p2, normal);
    return true;
  }
  return false;
}

static uint32_t pack_status(const SolverStats& st, bool contact) {
  uint32_t epa = (st.epa_status < 0) ? 15u : uint32_t(st.epa_status);
  return (uint32_t(st.gjk_status) & 7u) | ((epa & 15u) << 3) | (contact ? 128u : 0u) |
         ((std::min(st.gjk_iterations, 255u)) << 8) | ((std::min(st.epa_iterations, 127u)) << 16);
}

static void fill_result(hfcl_result& out, double d, const V3& n, const V3& p1, const V3& p2) {
  out.distance = d;
  for (int k = 0; k < 3; ++k) {
    out.normal[k] = n[k];
    out.p1[k] = p1[k];
    out.p2[k] = p2[k];
  }
  out.b1 = out.b2 = -1;
}

static void apply_guess(hfcl_query_request& q, const hfcl_guess* g) {
  if (!g) return;
  for (int k = 0; k < 3; ++k) q.cached_gjk_guess[k] = g->gjk_guess[k];
  q.cached_support_func_guess[0] = g->support_guess[0];
  q.cached_support_func_guess[1] = g->support_guess[1];
}
static void store_guess(const GJKSolver& s, hfcl_guess* g) {
  if (!g) return;
  for (int k = 0; k < 3; ++k) g->gjk_guess[k] = s.out_cached_guess[k];
  g->support_guess[0] = s.out_support_guess[0];
  g->support_guess[1] = s.out_support_guess[1];
}

// hpp::fcl::distance(), src/distance.cpp:60-109 + ShapeShapeDistancer::run, shape_shape_func.h:53-70.
// The result record is a fresh DistanceResult (min_distance = DBL_MAX) so update() always stores.
int distance_pair(const Shape& s1, const Tf& tf1, const Shape& s2, const Tf& tf2, const hfcl_distance_request& req_,
                  const hfcl_guess* guess_in, hfcl_result& out, hfcl_guess* guess_out) {
  hfcl_distance_request req = req_;
  apply_guess(req.q, guess_in);
  GJKSolver solver;
  solver.set(req);
  solver.out_cached_guess = solver.cached_guess;
  solver.out_support_guess[0] = solver.support_func_cached_guess[0];
  solver.out_support_guess[1] = solver.support_func_cached_guess[1];
  double d;
  V3 p1, p2, n;
  // the distance function matrix has no GEOM_TRIANGLE row or column (src/distance_func_matrix.cpp:283-560):
  // distance() throws "not yet supported" for a top-level TriangleP (src/distance.cpp:69-75)
  if (s1.kind == K_TRIANGLE || s2.kind == K_TRIANGLE) return HFCL_ERR_UNSUPPORTED_PAIR;
  if (!shape_shape_distance(s1, tf1, s2, tf2, solver, req.enable_signed_distance != 0, d, p1, p2, n))
    return HFCL_ERR_UNSUPPORTED_PAIR;
  fill_result(out, d, n, p1, p2);
  out.num_contacts = 0;
  out.status = pack_status(solver.stats, d <= 0);
  store_guess(solver, guess_out);
  return HFCL_OK;
}

// hpp::fcl::collide(), src/collision.cpp:69-130 + ShapeShapeCollider::run, shape_shape_func.h:134-163
int collide_pair(const Shape& s1, const Tf& tf1, const Shape& s2, const Tf& tf2, const hfcl_collision_request& req_,
                 const hfcl_guess* guess_in, hfcl_result& out, hfcl_guess* guess_out) {
  hfcl_collision_request req = req_;
  apply_guess(req.q, guess_in);
  if (req.security_margin == -std::numeric_limits<double>::infinity()) {  // collision.cpp:73-76
    fill_result(out, kMax, nan3(), nan3(), nan3());
    out.num_contacts = 0;
    out.status = 0x80000000u;
    return HFCL_OK;
  }
  if (req.num_max_contacts == 0) return HFCL_ERR_INVALID_ARGUMENT;  // collision.cpp:82-85
  GJKSolver solver;
  solver.set(req);
  solver.out_cached_guess = solver.cached_guess;
  solver.out_support_guess[0] = solver.support_func_cached_guess[0];
  solver.out_support_guess[1] = solver.support_func_cached_guess[1];
  const bool compute_penetration = req.enable_contact || (req.security_margin < 0);
  double d;
  V3 p1, p2, n;
  if (!shape_shape_distance(s1, tf1, s2, tf2, solver, compute_penetration, d, p1, p2, n))
    return HFCL_ERR_UNSUPPORTED_PAIR;
  const double distToCollision = d - req.security_margin;
  // updateDistanceLowerBoundFromLeaf on a fresh result (distance_lower_bound = DBL_MAX):
  // stored iff distToCollision < DBL_MAX (collision_data.h:1186-1197); otherwise NaN.
  fill_result(out, d, n, p1, p2);
  if (!(distToCollision < kMax)) fill_result(out, d, nan3(), nan3(), nan3());
  bool contact = (distToCollision <= req.q.collision_distance_threshold);
  out.num_contacts = contact ? 1 : 0;
  out.status = pack_status(solver.stats, contact);
  store_guess(solver, guess_out);
  return HFCL_OK;
}

}  // namespace orc
