"""BVHModel<OBBRSS> inputs on the host (plumbing; the construction itself is native:
csrc/hfcl_bvh_build.cpp behind `hfcl_bvh_build`).

The device traverses node arrays in the reference's encoding (include/hpp/fcl/BV/BV_node.h:52-101:
`first_child > 0`: children at first_child, first_child+1; `< 0`: leaf, primitive = -(first_child+1);
node 0 = root; 2*T-1 nodes).  This module offers

  * `uv_sphere(seg, ring, r)`   generateBVHModel(Sphere), shape/geometric_shape_to_BVH_model.h:92-150
  * `bumpy_sphere(...)`         cfg4's synthetic non-convex mesh
  * `load_obj(path)`            the OBJ subset the reference's tests read (test/utility.cpp:98-162)
  * `Mesh`, `MeshLibrary`       node / vertex / triangle buffers + the offset table of a mesh set."""
import numpy as np

from . import abi, engine


def uv_sphere(seg=50, ring=50, r=1.0):
    """Vertices/triangles of generateBVHModel(Sphere): seg*ring+2 vertices, 2*seg*ring triangles."""
    phid = np.pi * 2 / seg
    thetad = np.pi / (ring + 1)
    pts = []
    for i in range(ring):
        th = thetad * (i + 1)
        for j in range(seg):
            pts.append((r * np.sin(th) * np.cos(j * phid), r * np.sin(th) * np.sin(j * phid), r * np.cos(th)))
    pts.append((0, 0, r))
    pts.append((0, 0, -r))
    tris = []
    for i in range(ring - 1):
        for j in range(seg):
            a = i * seg + j
            b = i * seg if j == seg - 1 else i * seg + j + 1
            c = (i + 1) * seg + j
            d = (i + 1) * seg if j == seg - 1 else (i + 1) * seg + j + 1
            tris.append((a, c, b))
            tris.append((b, c, d))
    for j in range(seg):
        a, b = j, (0 if j == seg - 1 else j + 1)
        tris.append((ring * seg, a, b))
        a = (ring - 1) * seg + j
        b = (ring - 1) * seg if j == seg - 1 else (ring - 1) * seg + j + 1
        tris.append((a, ring * seg + 1, b))
    return np.array(pts, dtype=np.float64), np.array(tris, dtype=np.uint32)


def bumpy_sphere(seg=50, ring=50, r=1.0, amp=0.15, freq=3, phase=0.0):
    """cfg4's "bunny-like" mesh: a UV sphere displaced radially by a fixed low-frequency bump
    (non-convex).  seg=ring=50 -> 5000 triangles, 2502 vertices, 9999 BV nodes."""
    v, t = uv_sphere(seg, ring, 1.0)
    bump = 1.0 + amp * np.sin(freq * v[:, 0] + phase) * np.cos(freq * v[:, 1] - phase) * np.sin(freq * v[:, 2] + 0.5)
    return v * (r * bump)[:, None], t


def load_obj(path):
    """Vertices / triangles of a Wavefront OBJ, with the reference reader's behaviour
    (test/utility.cpp:98-162): faces are fanned around their first vertex; when the file has
    neither `vn` nor `vt` records every fan triangle repeats the first three indices (:139-143)."""
    pts, tris = [], []
    has_normal = has_texture = False
    with open(path, "rb") as f:
        for raw in f:
            tok = raw.decode("latin-1").split()
            if not tok or tok[0].startswith("#"):
                continue
            if tok[0][0] == "v":
                if tok[0][1:2] == "n":
                    has_normal = True
                elif tok[0][1:2] == "t":
                    has_texture = True
                else:
                    pts.append((float(tok[1]), float(tok[2]), float(tok[3])))
            elif tok[0][0] == "f":
                data = tok[1:]
                vid = lambda s: int(s.split("/")[0]) - 1  # atoi stops at '/'
                for t in range(len(data) - 2):
                    if not has_texture and not has_normal:
                        tris.append((vid(data[0]), vid(data[1]), vid(data[2])))
                    else:
                        tris.append((vid(data[0]), vid(data[t + 1]), vid(data[t + 2])))
    return np.array(pts, dtype=np.float64).reshape(-1, 3), np.array(tris, dtype=np.uint32).reshape(-1, 3)


def build_obbrss(vertices, triangles, n_threads=0):
    """-> (nodes[BVH_NODE_DTYPE], primitive_indices) as BVHModel<OBBRSS>::endModel() lays them out."""
    return engine.bvh_build(vertices, triangles, n_threads)


class Mesh:
    """One BVHModel<OBBRSS>: node array + vertex / triangle buffers."""

    def __init__(self, vertices, triangles):
        self.vertices = np.ascontiguousarray(vertices, dtype=np.float64)
        self.triangles = np.ascontiguousarray(triangles, dtype=np.uint32)
        self.nodes, self.primitive_indices = build_obbrss(self.vertices, self.triangles)

    @property
    def num_tris(self):
        return len(self.triangles)


class MeshLibrary:
    """Concatenated node / vertex / triangle arrays of several meshes + offset table
    (node_off, n_nodes, vert_off, tri_off) per mesh."""

    def __init__(self, meshes):
        self.meshes = list(meshes)
        self.nodes = np.concatenate([m.nodes for m in self.meshes])
        self.verts = np.ascontiguousarray(np.concatenate([m.vertices for m in self.meshes]))
        self.tris = np.ascontiguousarray(np.concatenate([m.triangles for m in self.meshes]))
        tab, no, vo, to = [], 0, 0, 0
        for m in self.meshes:
            tab.append((no, len(m.nodes), vo, to))
            no += len(m.nodes)
            vo += len(m.vertices)
            to += len(m.triangles)
        self.table = np.array(tab, dtype=np.uint64)
