#!/usr/bin/env python
"""Extracts which (node_type1, node_type2) entries the reference's collision / distance function matrices set
(src/collision_func_matrix.cpp, src/distance_func_matrix.cpp: lines `collision_matrix[A][B] = ...`) for the node
types in scope -> tests/golden/function_matrices.json.  Run in the build container (needs /root/reference); the
fixture travels, the reference does not."""
import json
import os
import re
import sys

REF = sys.argv[1] if len(sys.argv) > 1 else "/root/reference"
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
IN_SCOPE = ["BV_OBBRSS", "GEOM_BOX", "GEOM_SPHERE", "GEOM_CAPSULE", "GEOM_CONE", "GEOM_CYLINDER", "GEOM_CONVEX",
            "GEOM_PLANE", "GEOM_HALFSPACE", "GEOM_TRIANGLE", "GEOM_ELLIPSOID"]


def entries(path, name):
    txt = open(path).read()
    # strip the #ifdef HPP_FCL_HAS_OCTOMAP blocks' entries by keeping only in-scope names anyway
    found = set(re.findall(name + r"\[(\w+)\]\[(\w+)\]\s*=", txt))
    return sorted([a, b] for a, b in found if a in IN_SCOPE and b in IN_SCOPE)


co = open(os.path.join(REF, "include/hpp/fcl/collision_object.h")).read()
m = re.search(r"enum\s+NODE_TYPE\s*\{([^}]*)\}", co)
node_type_values = {e.strip(): i for i, e in enumerate(x for x in m.group(1).split(",") if x.strip())}

out = {"source": "hpp-fcl src/collision_func_matrix.cpp, src/distance_func_matrix.cpp, include/hpp/fcl/collision_object.h",
       "node_types": IN_SCOPE, "node_type_values": node_type_values,
       "collision": entries(os.path.join(REF, "src/collision_func_matrix.cpp"), "collision_matrix"),
       "distance": entries(os.path.join(REF, "src/distance_func_matrix.cpp"), "distance_matrix")}
dst = os.path.join(ROOT, "tests", "golden", "function_matrices.json")
json.dump(out, open(dst, "w"), indent=0)
print(dst, len(out["collision"]), "collision entries,", len(out["distance"]), "distance entries")
