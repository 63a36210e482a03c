#!/usr/bin/env python
"""Generates tests/golden/env_rob.npz from the reference's in-tree test meshes
(/root/reference/test/fcl_resources/env.obj, rob.obj -- the pair test/collision.cpp:625-654 and
test/distance.cpp:89-175 run on).  Only the parsed vertex / triangle arrays are stored (the GPU box
has no /root/reference).  Parsing = bvh_builder.load_obj (restating test/utility.cpp:98-162).

    python tests/golden/make_mesh_fixtures.py
"""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))
from __graft_entry__ import load_pkg  # noqa: E402

RES = "/root/reference/test/fcl_resources"

if __name__ == "__main__":
    bb = load_pkg().bvh_builder
    ev, et = bb.load_obj(os.path.join(RES, "env.obj"))
    rv, rt = bb.load_obj(os.path.join(RES, "rob.obj"))
    assert ev.shape == (6540, 3) and et.shape == (2180, 3), (ev.shape, et.shape)
    assert rv.shape == (648, 3) and rt.shape == (216, 3), (rv.shape, rt.shape)
    out = os.path.join(HERE, "env_rob.npz")
    np.savez_compressed(out, env_vertices=ev, env_triangles=et.astype(np.uint16), rob_vertices=rv,
                        rob_triangles=rt.astype(np.uint16))
    print(out, os.path.getsize(out), "bytes")
    print("env bbox", ev.min(0), ev.max(0), "rob bbox", rv.min(0), rv.max(0))
