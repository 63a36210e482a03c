#!/usr/bin/env python
"""Extracts the default values of QueryRequest / CollisionRequest / DistanceRequest from the reference's headers
(include/hpp/fcl/collision_data.h default constructors, narrowphase/narrowphase_defaults.h constants) and the enum
orders the ABI's integer codes rely on -> tests/golden/request_defaults.json.  Run where /root/reference exists."""
import json
import os
import re
import sys

REF = sys.argv[1] if len(sys.argv) > 1 else "/root/reference"
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
cd = open(os.path.join(REF, "include/hpp/fcl/collision_data.h")).read()
nd = open(os.path.join(REF, "include/hpp/fcl/narrowphase/narrowphase_defaults.h")).read()
dt = open(os.path.join(REF, "include/hpp/fcl/data_types.h")).read()

const = {k: float(v) for k, v in re.findall(r"constexpr\s+\w+\s+(\w+)\s*=\s*([0-9.e+-]+);", nd)}


def ctor_inits(text, start_pat):
    """member(value) pairs of the initialiser list that follows the first match of start_pat"""
    m = re.search(start_pat, text)
    body = text[m.end():text.index("{", m.end())]
    return {k: " ".join(v.split()) for k, v in re.findall(r"(\w+)\(((?:[^()]|\([^()]*\))*)\)", body)}


def sym(v):
    v = v.strip()
    if v in const:
        return const[v]
    if "dummy_precision" in v:
        return 1e-12  # Eigen::NumTraits<double>::dummy_precision()
    if "numeric_limits<FCL_REAL>::max" in v:
        return sys.float_info.max
    if v in ("true", "false"):
        return v == "true"
    if "::" in v:
        return v.split("::")[-1]
    try:
        return float(v)
    except ValueError:
        return v


q = ctor_inits(cd, r"QueryRequest\(\)\s*:")
c = ctor_inits(cd, r"CollisionRequest\(\)\s*:")
dm = re.search(r"DistanceRequest\(bool enable_nearest_points_ = (\w+),\s*bool enable_signed_distance_ = (\w+),\s*FCL_REAL rel_err_ = ([0-9.]+),"
               r"\s*FCL_REAL abs_err_ = ([0-9.]+)\)", cd)


def enum_order(text, name):
    m = re.search(r"enum\s+" + name + r"\s*\{([^}]*)\}", text)
    return [e.strip().split("=")[0].strip() for e in m.group(1).split(",") if e.strip()]


out = {
    "source": "hpp-fcl include/hpp/fcl/collision_data.h, narrowphase/narrowphase_defaults.h, data_types.h",
    "QueryRequest": {k: sym(q[k]) for k in ("gjk_initial_guess", "cached_gjk_guess", "gjk_max_iterations", "gjk_tolerance",
                                            "gjk_variant", "gjk_convergence_criterion", "gjk_convergence_criterion_type",
                                            "epa_max_iterations", "epa_tolerance", "collision_distance_threshold")},
    "CollisionRequest": {k: sym(c[k]) for k in ("num_max_contacts", "enable_contact", "security_margin", "break_distance",
                                                "distance_upper_bound")},
    "DistanceRequest": {"enable_nearest_points": dm.group(1) == "true", "enable_signed_distance": dm.group(2) == "true",
                        "rel_err": float(dm.group(3)), "abs_err": float(dm.group(4))},
    "enums": {n: enum_order(dt, n) for n in ("GJKInitialGuess", "GJKVariant", "GJKConvergenceCriterion",
                                             "GJKConvergenceCriterionType")},
}
dst = os.path.join(ROOT, "tests", "golden", "request_defaults.json")
json.dump(out, open(dst, "w"), indent=1)
print(open(dst).read())
