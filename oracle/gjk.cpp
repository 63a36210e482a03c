// ORACLE -- TEST INFRASTRUCTURE ONLY.  Not part of the product path (see vec3.hpp).
// Restatement of /root/reference/src/narrowphase/gjk.cpp and src/intersect.cpp:570-705.
#include "gjk.hpp"
#include <algorithm>
#include <cassert>

namespace orc {

// ---------------------------------------------------------------------------------------
// Project::project*Origin, src/intersect.cpp:570-705
// ---------------------------------------------------------------------------------------
ProjectResult project_line_origin(const V3& a, const V3& b) {  // :570-594
  ProjectResult res;
  const V3 d = b - a;
  const double l = sqnorm(d);
  if (l > 0) {
    const double t = -dot(a, d);
    res.param[1] = (t >= l) ? 1 : ((t <= 0) ? 0 : (t / l));
    res.param[0] = 1 - res.param[1];
    if (t >= l) {
      res.sqr_distance = sqnorm(b);
      res.encode = 2;
    } else if (t <= 0) {
      res.sqr_distance = sqnorm(a);
      res.encode = 1;
    } else {
      res.sqr_distance = sqnorm(a + d * res.param[1]);
      res.encode = 3;
    }
  }
  return res;
}

ProjectResult project_triangle_origin(const V3& a, const V3& b, const V3& c) {  // :596-646
  ProjectResult res;
  static const size_t nexti[3] = {1, 2, 0};
  const V3* vt[] = {&a, &b, &c};
  const V3 dl[] = {a - b, b - c, c - a};
  const V3 n = cross(dl[0], dl[1]);
  const double l = sqnorm(n);
  if (l > 0) {
    double mindist = -1;
    for (size_t i = 0; i < 3; ++i) {
      if (dot(*vt[i], cross(dl[i], n)) > 0) {
        size_t j = nexti[i];
        ProjectResult rl = project_line_origin(*vt[i], *vt[j]);
        if (mindist < 0 || rl.sqr_distance < mindist) {
          mindist = rl.sqr_distance;
          res.encode = static_cast<unsigned>(((rl.encode & 1) ? 1 << i : 0) + ((rl.encode & 2) ? 1 << j : 0));
          res.param[i] = rl.param[0];
          res.param[j] = rl.param[1];
          res.param[nexti[j]] = 0;
        }
      }
    }
    if (mindist < 0) {
      double d = dot(a, n);
      double s = std::sqrt(l);
      V3 o_to_project = n * (d / l);
      mindist = sqnorm(o_to_project);
      res.encode = 7;
      res.param[0] = norm(cross(dl[1], b - o_to_project)) / s;
      res.param[1] = norm(cross(dl[2], c - o_to_project)) / s;
      res.param[2] = 1 - res.param[0] - res.param[1];
    }
    res.sqr_distance = mindist;
  }
  return res;
}

ProjectResult project_tetrahedra_origin(const V3& a, const V3& b, const V3& c, const V3& d) {  // :648-705
  ProjectResult res;
  static const size_t nexti[] = {1, 2, 0};
  const V3* vt[] = {&a, &b, &c, &d};
  const V3 dl[3] = {a - d, b - d, c - d};
  double vl = triple(dl[0], dl[1], dl[2]);
  bool ng = (vl * dot(a, cross(b - c, a - b))) <= 0;
  if (ng && std::abs(vl) > 0) {
    double mindist = -1;
    for (size_t i = 0; i < 3; ++i) {
      size_t j = nexti[i];
      double s = vl * dot(d, cross(dl[i], dl[j]));
      if (s > 0) {
        ProjectResult rt = project_triangle_origin(*vt[i], *vt[j], d);
        if (mindist < 0 || rt.sqr_distance < mindist) {
          mindist = rt.sqr_distance;
          res.encode = static_cast<unsigned>((rt.encode & 1 ? 1 << i : 0) + (rt.encode & 2 ? 1 << j : 0) +
                                             (rt.encode & 4 ? 8 : 0));
          res.param[i] = rt.param[0];
          res.param[j] = rt.param[1];
          res.param[nexti[j]] = 0;
          res.param[3] = rt.param[2];
        }
      }
    }
    if (mindist < 0) {
      mindist = 0;
      res.encode = 15;
      res.param[0] = triple(c, b, d) / vl;
      res.param[1] = triple(a, c, d) / vl;
      res.param[2] = triple(b, a, d) / vl;
      res.param[3] = 1 - (res.param[0] + res.param[1] + res.param[2]);
    }
    res.sqr_distance = mindist;
  } else if (!ng) {
    res = project_triangle_origin(a, b, c);
    res.param[3] = 0;
  }
  return res;
}

// ---------------------------------------------------------------------------------------
// details::getClosestPoints, gjk.cpp:94-151
// ---------------------------------------------------------------------------------------
void get_closest_points(const Simplex& s, V3& w0, V3& w1) {
  ProjectResult proj;
  switch (s.rank) {
    case 1:
      w0 = s.v[0].w0;
      w1 = s.v[0].w1;
      return;
    case 2: {
      const V3 &a = s.v[0].w, &a0 = s.v[0].w0, &a1 = s.v[0].w1, &b = s.v[1].w, &b0 = s.v[1].w0,
               &b1 = s.v[1].w1;
      double la, lb;
      V3 N = b - a;
      la = dot(N, -a);
      if (la <= 0) {
        w0 = a0;
        w1 = a1;
      } else {
        lb = sqnorm(N);
        if (la > lb) {
          w0 = b0;
          w1 = b1;
        } else {
          lb = la / lb;
          la = 1 - lb;
          w0 = la * a0 + lb * b0;
          w1 = la * a1 + lb * b1;
        }
      }
      return;
    }
    case 3:
      proj = project_triangle_origin(s.v[0].w, s.v[1].w, s.v[2].w);
      break;
    case 4:
      proj = project_tetrahedra_origin(s.v[0].w, s.v[1].w, s.v[2].w, s.v[3].w);
      break;
    default:
      assert(false);
  }
  w0 = V3(0, 0, 0);
  w1 = V3(0, 0, 0);
  for (int i = 0; i < s.rank; ++i) {
    w0 += proj.param[i] * s.v[i].w0;
    w1 += proj.param[i] * s.v[i].w1;
  }
}

// details::inflate<>, gjk.cpp:158-173 (both instantiations are identical)
static void inflate(const MinkowskiDiff& shape, const V3& normal, V3& w0, V3& w1) {
  const double* I = shape.swept_sphere_radius;
  if (!(I[0] > 0 || I[1] > 0)) return;
  if (I[0] > 0) w0 += I[0] * normal;
  if (I[1] > 0) w1 -= I[1] * normal;
}

void GJK::get_witness_points_and_normal(const MinkowskiDiff& sh, V3& w0, V3& w1, V3& normal) const {  // :177-186
  get_closest_points(simplex, w0, w1);
  if (norm(w1 - w0) > kDummyPrecision)
    normal = normalized(w1 - w0);
  else
    normal = -normalized(ray);
  inflate(sh, normal, w0, w1);
}

// ---------------------------------------------------------------------------------------
// GJK::evaluate, gjk.cpp:188-370
// ---------------------------------------------------------------------------------------
GJK::Status GJK::evaluate(const MinkowskiDiff& shape_, const V3& guess, const int hint_in[2]) {
  double alpha = 0;
  iterations = 0;
  iterations_momentum_stop = 0;
  const double swept_sphere_radius = shape_.swept_sphere_radius[0] + shape_.swept_sphere_radius[1];
  const double upper_bound = distance_upper_bound + swept_sphere_radius;

  Simplex simplices[2];
  int current = 0;
  status = NoCollision;
  shape = &shape_;
  distance = 0.0;
  simplices[current].rank = 0;
  support_hint[0] = hint_in[0];
  support_hint[1] = hint_in[1];

  double rl = norm(guess);
  if (rl < tolerance) {
    ray = V3(-1, 0, 0);
    rl = 1;
  } else
    ray = guess;

  int current_gjk_variant = gjk_variant;
  V3 w = ray;
  V3 dir = ray;
  V3 y;
  double momentum;
  const bool normalize_support_direction = shape->normalize_support_direction;
  do {
    int next = 1 - current;
    Simplex& curr_simplex = simplices[current];
    Simplex& next_simplex = simplices[next];

    // check A
    if (rl < tolerance) {
      status = Collision;
      distance = rl;
      break;
    }

    switch (current_gjk_variant) {
      case DefaultGJK:
        dir = ray;
        break;
      case NesterovAcceleration:
        if (normalize_support_direction) {
          momentum = (double(iterations) + 2) / (double(iterations) + 3);
          y = momentum * ray + (1 - momentum) * w;
          double y_norm = norm(y);
          dir = (momentum * dir) / norm(dir) + ((1 - momentum) * y) / y_norm;
        } else {
          momentum = (double(iterations) + 1) / (double(iterations) + 3);
          y = momentum * ray + (1 - momentum) * w;
          dir = momentum * dir + (1 - momentum) * y;
        }
        break;
      case PolyakAcceleration:
        momentum = 1 / (double(iterations) + 1);
        dir = momentum * dir + (1 - momentum) * ray;
        break;
    }

    // appendVertex(curr_simplex, -dir, support_hint)  :281, :431-435
    get_support(-dir, curr_simplex.v[curr_simplex.rank], support_hint);
    ++curr_simplex.rank;
    w = curr_simplex.v[curr_simplex.rank - 1].w;

    // check B
    double omega = dot(dir, w) / norm(dir);
    if (omega > upper_bound) {
      distance = omega - swept_sphere_radius;
      status = NoCollisionEarlyStopped;
      break;
    }

    // momentum removal :296-304
    if (current_gjk_variant != DefaultGJK) {
      double frank_wolfe_duality_gap = 2 * dot(ray, ray - w);
      if (frank_wolfe_duality_gap - tolerance <= 0) {
        --simplices[current].rank;  // removeVertex
        current_gjk_variant = DefaultGJK;
        iterations_momentum_stop = iterations;
        continue;
      }
    }

    // check C
    bool cv_check_passed = check_convergence(w, rl, alpha, omega);
    if (iterations > 0 && cv_check_passed) {
      if (iterations > 0) --simplices[current].rank;  // removeVertex
      if (current_gjk_variant != DefaultGJK) {
        current_gjk_variant = DefaultGJK;
        iterations_momentum_stop = iterations;
        continue;
      }
      distance = rl - swept_sphere_radius;
      if (distance < tolerance)
        status = CollisionWithPenetrationInformation;
      else
        status = NoCollision;
      break;
    }

    bool inside = false;
    switch (curr_simplex.rank) {
      case 1:
        ray = w;
        inside = false;
        next_simplex.rank = 1;
        next_simplex.v[0] = curr_simplex.v[0];
        break;
      case 2:
        inside = project_line(curr_simplex, next_simplex);
        break;
      case 3:
        inside = project_triangle(curr_simplex, next_simplex);
        break;
      case 4:
        inside = project_tetra(curr_simplex, next_simplex);
        break;
      default:
        assert(false);
    }
    current = next;
    rl = norm(ray);
    if (inside || rl == 0) {
      status = Collision;
      distance = rl;
      break;
    }

    status = ((++iterations) < max_iterations) ? status : Failed;
  } while (status == NoCollision);

  simplex = simplices[current];
  return status;
}

bool GJK::check_convergence(const V3& w, double rl, double& alpha, double omega) const {  // :372-425
  switch (convergence_criterion) {
    case CritDefault: {
      alpha = std::max(alpha, omega);
      const double diff = rl - alpha;
      return ((diff - (tolerance + tolerance * rl)) <= 0);
    }
    case CritDualityGap: {
      const double diff = 2 * dot(ray, ray - w);
      if (convergence_criterion_type == Absolute) return ((diff - tolerance) <= 0);
      return (((diff / tolerance * rl) - tolerance * rl) <= 0);
    }
    case CritHybrid: {
      alpha = std::max(alpha, omega);
      const double diff = rl * rl - alpha * alpha;
      if (convergence_criterion_type == Absolute) return ((diff - tolerance) <= 0);
      return (((diff / tolerance * rl) - tolerance * rl) <= 0);
    }
  }
  return false;
}

// originToPoint / originToSegment / originToTriangle, gjk.cpp:494-541
static inline void origin_to_point(const Simplex& cur, int a, const V3& A, Simplex& next, V3& ray) {
  ray = A;
  next.v[0] = cur.v[a];
  next.rank = 1;
}
static inline void origin_to_segment(const Simplex& cur, int a, int b, const V3& A, const V3& B, const V3& AB,
                                     double ABdotAO, Simplex& next, V3& ray) {
  ray = dot(AB, B) * A + ABdotAO * B;
  next.v[0] = cur.v[b];
  next.v[1] = cur.v[a];
  next.rank = 2;
  ray = ray / sqnorm(AB);
}
static inline bool origin_to_triangle(const Simplex& cur, int a, int b, int c, const V3& ABC, double ABCdotAO,
                                      Simplex& next, V3& ray) {
  next.rank = 3;
  next.v[2] = cur.v[a];
  if (ABCdotAO == 0) {
    next.v[0] = cur.v[c];
    next.v[1] = cur.v[b];
    ray = V3(0, 0, 0);
    return true;
  }
  if (ABCdotAO > 0) {
    next.v[0] = cur.v[c];
    next.v[1] = cur.v[b];
  } else {
    next.v[0] = cur.v[b];
    next.v[1] = cur.v[c];
  }
  ray = (-ABCdotAO / sqnorm(ABC)) * ABC;
  return false;
}

bool GJK::project_line(const Simplex& cur, Simplex& next) {  // :543-569
  const int a = 1, b = 0;
  const V3 A = cur.v[a].w, B = cur.v[b].w;
  const V3 AB = B - A;
  const double d = dot(AB, -A);
  if (d == 0) {
    origin_to_point(cur, a, A, next, ray);
    return is_zero(A);
  } else if (d < 0) {
    origin_to_point(cur, a, A, next, ray);
  } else
    origin_to_segment(cur, a, b, A, B, AB, d, next, ray);
  return false;
}

bool GJK::project_triangle(const Simplex& cur, Simplex& next) {  // :571-611
  const int a = 2, b = 1, c = 0;
  const V3 A = cur.v[a].w, B = cur.v[b].w, C = cur.v[c].w;
  const V3 AB = B - A, AC = C - A, ABC = cross(AB, AC);

  double edgeAC2o = dot(cross(ABC, AC), -A);
  if (edgeAC2o >= 0) {
    double towardsC = dot(AC, -A);
    if (towardsC >= 0) {
      origin_to_segment(cur, a, c, A, C, AC, towardsC, next, ray);
    } else {
      double towardsB = dot(AB, -A);
      if (towardsB < 0)
        origin_to_point(cur, a, A, next, ray);
      else
        origin_to_segment(cur, a, b, A, B, AB, towardsB, next, ray);
    }
  } else {
    double edgeAB2o = dot(cross(AB, ABC), -A);
    if (edgeAB2o >= 0) {
      double towardsB = dot(AB, -A);
      if (towardsB < 0)
        origin_to_point(cur, a, A, next, ray);
      else
        origin_to_segment(cur, a, b, A, B, AB, towardsB, next, ray);
    } else {
      return origin_to_triangle(cur, a, b, c, ABC, dot(ABC, -A), next, ray);
    }
  }
  return false;
}

// projectTetrahedraOrigin, gjk.cpp:613-1010.  The reference inlines the action at every
// leaf of a generated decision tree (doc/gjk.py); here the tree only *classifies* the
// Voronoi region (same predicates, same order, same <= 0 tests) and one switch applies
// the action.  The predicates are pure functions of the four vertices, so evaluating
// them lazily gives the reference's path.
namespace {
enum Region { R_A, R_AB, R_AC, R_AD, R_ABC, R_ACD, R_ADB, R_INSIDE };
}

bool GJK::project_tetra(const Simplex& cur, Simplex& next) {
  const int a = 3, b = 2, c = 1, d = 0;
  const V3 A = cur.v[a].w, B = cur.v[b].w, C = cur.v[c].w, D = cur.v[d].w;
  const double aa = sqnorm(A);
  const double da = dot(D, A), db = dot(D, B), dc = dot(D, C), dd = dot(D, D);
  const double da_aa = da - aa;
  const double ca = dot(C, A), cb = dot(C, B), cc = dot(C, C);
  const double cd = dc;
  const double ca_aa = ca - aa;
  const double ba = dot(B, A), bb = dot(B, B);
  const double bc = cb, bd = db;
  const double ba_aa = ba - aa, ba_ca = ba - ca, ca_da = ca - da, da_ba = da - ba;
  const V3 a_cross_b = cross(A, B);
  const V3 a_cross_c = cross(A, C);

  // predicate names follow the "aN" labels of the reference's comments
  const bool a10 = ba_aa <= 0;                                     // AB.AO >= 0
  const bool a11 = ca_aa <= 0;                                     // AC.AO >= 0
  const bool a12 = da_aa <= 0;                                     // AD.AO >= 0
  const bool a3 = -dot(D, a_cross_b) <= 0;                         // ADB.AO >= 0
  const bool a1 = dot(C, a_cross_b) <= 0;                          // ABC.AO >= 0
  const bool a2 = dot(D, a_cross_c) <= 0;                          // ACD.AO >= 0
  const bool a9 = ba * da_ba + bd * ba_aa - bb * da_aa <= 0;       // (ADB ^ AB).AO >= 0
  const bool a8 = da * da_ba + dd * ba_aa - db * da_aa <= 0;       // (ADB ^ AD).AO >= 0
  const bool a4 = ba * ba_ca + bb * ca_aa - bc * ba_aa <= 0;       // (ABC ^ AB).AO >= 0
  const bool a5 = ca * ba_ca + cb * ca_aa - cc * ba_aa <= 0;       // (ABC ^ AC).AO >= 0
  const bool a6 = ca * ca_da + cc * da_aa - cd * ca_aa <= 0;       // (ACD ^ AC).AO >= 0
  const bool a7 = da * ca_da + dc * da_aa - dd * ca_aa <= 0;       // (ACD ^ AD).AO >= 0

  Region r;
  if (a10) {
    if (a3) {
      if (a9) {
        if (a12) r = a4 ? R_ABC : R_AB;
        else if (a4) r = a5 ? (a6 ? R_ACD : R_AC) : R_ABC;
        else r = R_AB;
      } else {
        if (a8) r = R_ADB;
        else if (a6) r = a7 ? R_AD : R_ACD;
        else r = a7 ? R_AD : R_AC;
      }
    } else {
      if (a1) {
        if (a4) r = a5 ? (a6 ? R_ACD : R_AC) : R_ABC;
        else r = R_AB;
      } else {
        if (a2) {
          if (a6) r = a7 ? R_AD : R_ACD;
          else r = a11 ? R_AC : R_AD;
        } else
          r = R_INSIDE;
      }
    }
  } else {
    if (a11) {
      if (a2) {
        if (a12) {
          if (a6) r = a7 ? (a8 ? R_ADB : R_AD) : R_ACD;
          else r = a5 ? R_AC : R_ABC;
        } else {
          if (a5) r = a6 ? R_ACD : R_AC;
          else r = a1 ? R_ABC : R_ACD;
        }
      } else {
        if (a1) r = a5 ? R_AC : R_ABC;
        else if (a3) r = a8 ? R_ADB : R_AD;
        else r = R_INSIDE;
      }
    } else {
      if (a12) {
        if (a3) {
          if (a7) r = a8 ? R_ADB : R_AD;
          else r = a2 ? R_ACD : R_ADB;
        } else {
          if (a2) r = a7 ? R_AD : R_ACD;
          else r = R_INSIDE;
        }
      } else
        r = R_A;
    }
  }

  switch (r) {
    case R_A:
      origin_to_point(cur, a, A, next, ray);
      return false;
    case R_AB:
      origin_to_segment(cur, a, b, A, B, B - A, -ba_aa, next, ray);
      return false;
    case R_AC:
      origin_to_segment(cur, a, c, A, C, C - A, -ca_aa, next, ray);
      return false;
    case R_AD:
      origin_to_segment(cur, a, d, A, D, D - A, -da_aa, next, ray);
      return false;
    case R_ABC:
      origin_to_triangle(cur, a, b, c, cross(B - A, C - A), -dot(C, a_cross_b), next, ray);
      return false;
    case R_ACD:
      origin_to_triangle(cur, a, c, d, cross(C - A, D - A), -dot(D, a_cross_c), next, ray);
      return false;
    case R_ADB:
      origin_to_triangle(cur, a, d, b, cross(D - A, B - A), dot(D, a_cross_b), next, ray);
      return false;
    case R_INSIDE:
      ray = V3(0, 0, 0);
      next.v[0] = cur.v[d];
      next.v[1] = cur.v[c];
      next.v[2] = cur.v[b];
      next.v[3] = cur.v[a];
      next.rank = 4;
      return true;
  }
  return false;
}

// GJK::encloseOrigin, gjk.cpp:437-492
bool GJK::enclose_origin() {
  V3 axis(0, 0, 0);
  int hint[2] = {0, 0};
  Simplex& s = simplex;
  auto append = [&](const V3& v) {
    get_support(v, s.v[s.rank], hint);
    ++s.rank;
  };
  auto remove = [&]() { --s.rank; };
  switch (s.rank) {
    case 1:
      for (int i = 0; i < 3; ++i) {
        axis[i] = 1;
        append(axis);
        if (enclose_origin()) return true;
        remove();
        axis[i] = -1;
        append(-axis);
        if (enclose_origin()) return true;
        remove();
        axis[i] = 0;
      }
      break;
    case 2: {
      V3 d = s.v[1].w - s.v[0].w;
      for (int i = 0; i < 3; ++i) {
        axis[i] = 1;
        V3 p = cross(d, axis);
        if (!is_zero(p)) {
          append(p);
          if (enclose_origin()) return true;
          remove();
          append(-p);
          if (enclose_origin()) return true;
          remove();
        }
        axis[i] = 0;
      }
    } break;
    case 3:
      axis = cross(s.v[1].w - s.v[0].w, s.v[2].w - s.v[0].w);
      if (!is_zero(axis)) {
        append(axis);
        if (enclose_origin()) return true;
        remove();
        append(-axis);
        if (enclose_origin()) return true;
        remove();
      }
      break;
    case 4:
      if (std::abs(triple(s.v[0].w - s.v[3].w, s.v[1].w - s.v[3].w, s.v[2].w - s.v[3].w)) > 0) return true;
      break;
  }
  return false;
}

// ---------------------------------------------------------------------------------------
// EPA, gjk.cpp:1012-1466
// ---------------------------------------------------------------------------------------
void EPA::list_append(FaceList& l, int f) {  // gjk.h:292-298
  fc_store[f].prev = -1;
  fc_store[f].next = l.root;
  if (l.root != -1) fc_store[l.root].prev = f;
  l.root = f;
  ++l.count;
}
void EPA::list_remove(FaceList& l, int f) {  // gjk.h:300-307
  Face& face = fc_store[f];
  if (face.next != -1) fc_store[face.next].prev = face.prev;
  if (face.prev != -1) fc_store[face.prev].next = face.next;
  if (f == l.root) l.root = face.next;
  --l.count;
}
void EPA::bind(int fa, size_t ea, int fb, size_t eb) {  // gjk.h:312-320
  fc_store[fa].adjacent_edge[ea] = eb;
  fc_store[fa].adjacent_faces[ea] = fb;
  fc_store[fb].adjacent_edge[eb] = ea;
  fc_store[fb].adjacent_faces[eb] = fa;
}

void EPA::reset(size_t max_it, double tol) {  // :1014-1037
  max_iterations = max_it;
  tolerance = tol;
  sv_store.assign(max_iterations + 4, SimplexV());
  fc_store.assign(2 * max_iterations + 4, Face());
  status = DidNotRun;
  normal = V3(0, 0, 0);
  support_hint[0] = support_hint[1] = 0;
  depth = 0;
  closest_face = -1;
  result.rank = 0;
  hull = FaceList();
  num_vertices = 0;
  stock = FaceList();
  for (size_t i = 0; i < fc_store.size(); ++i) list_append(stock, int(fc_store.size() - i - 1));
  iterations = 0;
}

int EPA::new_face(size_t id_a, size_t id_b, size_t id_c, bool force) {  // :1068-1138
  if (stock.root != -1) {
    int fi = stock.root;
    list_remove(stock, fi);
    list_append(hull, fi);
    Face& face = fc_store[fi];
    face.pass = 0;
    face.vertex_id[0] = id_a;
    face.vertex_id[1] = id_b;
    face.vertex_id[2] = id_c;
    const SimplexV& a = sv_store[id_a];
    const SimplexV& b = sv_store[id_b];
    const SimplexV& c = sv_store[id_c];
    face.n = cross(b.w - a.w, c.w - a.w);

    if (norm(face.n) > std::numeric_limits<double>::epsilon()) {
      face.n = normalized(face.n);
      double a_dot_nab = dot(a.w, cross(b.w - a.w, face.n));
      double b_dot_nbc = dot(b.w, cross(c.w - b.w, face.n));
      double c_dot_nca = dot(c.w, cross(a.w - c.w, face.n));
      if (a_dot_nab >= -tolerance && b_dot_nbc >= -tolerance && c_dot_nca >= -tolerance) {
        face.d = dot(a.w, face.n);
        face.ignore = false;
      } else {
        face.d = std::numeric_limits<double>::max();
        face.ignore = true;
      }
      if (face.d >= -tolerance || force)
        return fi;
      else
        status = NonConvex;
    } else
      status = Degenerated;

    list_remove(hull, fi);
    list_append(stock, fi);
    return -1;
  }
  status = OutOfFaces;
  return -1;
}

int EPA::find_closest_face() {  // :1141-1154
  int minf = hull.root;
  double mind = std::numeric_limits<double>::max();
  for (int f = minf; f != -1; f = fc_store[f].next) {
    if (fc_store[f].ignore) continue;
    double sqd = fc_store[f].d * fc_store[f].d;
    if (sqd < mind) {
      minf = f;
      mind = sqd;
    }
  }
  return minf;
}

EPA::Status EPA::evaluate(GJK& gjk, const V3& guess) {  // :1156-1316
  Simplex& simplex = gjk.simplex;
  support_hint[0] = gjk.support_hint[0];
  support_hint[1] = gjk.support_hint[1];

  bool enclosed_origin = gjk.enclose_origin();
  if ((simplex.rank > 1) && enclosed_origin) {
    while (hull.root != -1) {
      int f = hull.root;
      list_remove(hull, f);
      list_append(stock, f);
    }
    status = Valid;
    num_vertices = 0;

    if (dot(simplex.v[0].w - simplex.v[3].w,
            cross(simplex.v[1].w - simplex.v[3].w, simplex.v[2].w - simplex.v[3].w)) < 0) {
      std::swap(simplex.v[0], simplex.v[1]);
    }
    for (size_t i = 0; i < 4; ++i) sv_store[num_vertices++] = simplex.v[i];

    int tetra[4];
    tetra[0] = new_face(0, 1, 2, true);
    tetra[1] = new_face(1, 0, 3, true);
    tetra[2] = new_face(2, 1, 3, true);
    tetra[3] = new_face(0, 2, 3, true);

    if (hull.count == 4) {
      bind(tetra[0], 0, tetra[1], 0);
      bind(tetra[0], 1, tetra[2], 0);
      bind(tetra[0], 2, tetra[3], 0);
      bind(tetra[1], 1, tetra[3], 2);
      bind(tetra[1], 2, tetra[2], 1);
      bind(tetra[2], 2, tetra[3], 1);

      closest_face = find_closest_face();
      Face outer = fc_store[closest_face];

      status = Valid;
      iterations = 0;
      size_t pass = 0;
      for (; iterations < max_iterations; ++iterations) {
        if (num_vertices >= sv_store.size()) {
          status = OutOfVertices;
          break;
        }
        Horizon horizon;
        SimplexV& w = sv_store[num_vertices++];
        bool valid = true;
        fc_store[closest_face].pass = ++pass;
        gjk.get_support(fc_store[closest_face].n, w, support_hint);

        const Face& cf = fc_store[closest_face];
        const SimplexV& vf1 = sv_store[cf.vertex_id[0]];
        const SimplexV& vf2 = sv_store[cf.vertex_id[1]];
        const SimplexV& vf3 = sv_store[cf.vertex_id[2]];
        double fdist = dot(cf.n, w.w - vf1.w);
        double wnorm = norm(w.w);
        if (fdist <= tolerance + tolerance * wnorm) {
          status = AccuracyReached;
          break;
        }
        if (norm(w.w - vf1.w) <= tolerance + tolerance * wnorm ||
            norm(w.w - vf2.w) <= tolerance + tolerance * wnorm ||
            norm(w.w - vf3.w) <= tolerance + tolerance * wnorm) {
          status = AccuracyReached;
          break;
        }

        for (size_t j = 0; (j < 3) && valid; ++j)
          valid &= expand(pass, w, fc_store[closest_face].adjacent_faces[j],
                          fc_store[closest_face].adjacent_edge[j], horizon);

        if (!valid || horizon.num_faces < 3) break;  // status already set by expand
        bind(horizon.first_face, 2, horizon.current_face, 1);
        list_remove(hull, closest_face);
        list_append(stock, closest_face);
        closest_face = find_closest_face();
        outer = fc_store[closest_face];
      }

      status = (iterations < max_iterations) ? status : Failed;
      normal = outer.n;
      depth = outer.d + gjk.shape->swept_sphere_radius[0] + gjk.shape->swept_sphere_radius[1];
      result.rank = 3;
      result.v[0] = sv_store[outer.vertex_id[0]];
      result.v[1] = sv_store[outer.vertex_id[1]];
      result.v[2] = sv_store[outer.vertex_id[2]];
      return status;
    }
  }

  // FallBack :1299-1315
  status = FallBack;
  normal = -guess;
  double nl = norm(normal);
  if (nl > 0)
    normal = normal / nl;
  else
    normal = V3(1, 0, 0);
  depth = 0;
  result.rank = 1;
  result.v[0] = simplex.v[0];
  return status;
}

bool EPA::expand(size_t pass, const SimplexV& w, int fi, size_t e, Horizon& horizon) {  // :1361-1449
  static const size_t nexti[] = {1, 2, 0};
  static const size_t previ[] = {2, 0, 1};
  const size_t id_w = num_vertices - 1;

  if (fc_store[fi].pass == pass) {
    status = InvalidHull;
    return false;
  }
  const size_t e1 = nexti[e];
  const double dummy_precision = 3 * std::sqrt(std::numeric_limits<double>::epsilon());
  const SimplexV& vf = sv_store[fc_store[fi].vertex_id[e]];
  if (dot(fc_store[fi].n, w.w - vf.w) < dummy_precision) {
    // case 1: support point "below" f
    int nf = new_face(fc_store[fi].vertex_id[e1], fc_store[fi].vertex_id[e], id_w);
    if (nf != -1) {
      bind(nf, 0, fi, e);
      if (horizon.current_face != -1)
        bind(nf, 2, horizon.current_face, 1);
      else
        horizon.first_face = nf;
      horizon.current_face = nf;
      ++horizon.num_faces;
      return true;
    }
    return false;
  }
  // case 2: "above" f
  const size_t e2 = previ[e];
  fc_store[fi].pass = pass;
  if (expand(pass, w, fc_store[fi].adjacent_faces[e1], fc_store[fi].adjacent_edge[e1], horizon) &&
      expand(pass, w, fc_store[fi].adjacent_faces[e2], fc_store[fi].adjacent_edge[e2], horizon)) {
    list_remove(hull, fi);
    list_append(stock, fi);
    return true;
  }
  return false;
}

void EPA::get_witness_points_and_normal(const MinkowskiDiff& sh, V3& w0, V3& w1, V3& nrm) const {  // :1451-1466
  get_closest_points(result, w0, w1);
  if (norm(w0 - w1) > kDummyPrecision) {
    if (depth >= 0)
      nrm = normalized(w0 - w1);
    else
      nrm = normalized(w1 - w0);
  } else {
    nrm = normal;
  }
  inflate(sh, nrm, w0, w1);
}

}  // namespace orc
