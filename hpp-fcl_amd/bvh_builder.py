"""Host-side construction of BVHModel<OBBRSS> inputs (plumbing; numpy only).

The device traverses node arrays in the reference's encoding (include/hpp/fcl/BV/BV_node.h:52-101:
`first_child > 0`: children at first_child, first_child+1; `< 0`: leaf, primitive = -(first_child+1);
node 0 = root; 2*T-1 nodes).  This module produces such arrays:

  * `uv_sphere(seg, ring, r)`            generateBVHModel(Sphere), shape/geometric_shape_to_BVH_model.h:92-150
  * `build_obbrss(vertices, triangles)`  the recipe of BVHModel::recursiveBuildTree
    (src/BVH/BVH_model.cpp:892-960): fit (covariance of the triangle vertices
    src/BVH/BVH_utility.cpp:183-259 -> eigenvectors -> axes ordered max/mid/cross
    src/BVH/BV_fitter.cpp:50-76 -> OBB centre/extent from min/max projections
    src/BVH/BVH_utility.cpp:529-575), mean split along OBB axis 0
    (src/BVH/BV_splitter.cpp:81-118,276-279), children allocated adjacently.

Differences from the reference builder (it is a "next" row): the eigen-decomposition
is LAPACK's (numpy.linalg.eigh) instead of the 50-sweep Jacobi of internal/tools.h:103-202, so axis
signs -- and with them left/right child order -- can differ from hpp-fcl's trees; the RSS part of
each node is a valid but looser fit (rectangle = OBB mid-plane, radius = OBB half-thickness) than
the reference's PQP-style fit (BVH_utility.cpp:264-482).  Any tree produced here is a legal input;
oracle and device traverse the same arrays."""
import numpy as np

from . import abi


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


def _fit(verts, tris, idx):
    """BVFitter<OBBRSS>::fit (BV_fitter.cpp:501-531) for the triangles `idx`."""
    P = verts[tris[idx].reshape(-1)]  # every triangle contributes its 3 vertices
    n_points = len(P)
    S1 = P.sum(axis=0)
    S2 = P.T @ P
    M = S2 - np.outer(S1, S1) / n_points
    w, V = np.linalg.eigh(M)  # ascending
    # axisFromEigen: col0 = largest, col1 = middle, col2 = col0 x col1
    a0, a1 = V[:, 2], V[:, 1]
    a2 = np.cross(a0, a1)
    axes = np.stack([a0, a1, a2], axis=1)
    proj = P @ axes
    mx, mn = proj.max(axis=0), proj.min(axis=0)
    center = axes @ ((mx + mn) / 2)
    extent = (mx - mn) / 2
    return axes, center, extent


def build_obbrss(vertices, triangles):
    """-> (nodes[BVH_NODE_DTYPE], primitive_indices).  Node numbering as recursiveBuildTree."""
    verts = np.ascontiguousarray(vertices, dtype=np.float64)
    tris = np.ascontiguousarray(triangles, dtype=np.int64)
    nt = len(tris)
    nodes = np.zeros(2 * nt - 1, dtype=abi.BVH_NODE_DTYPE)
    prim = np.arange(nt, dtype=np.int64)
    centroids = verts[tris].mean(axis=1)
    vsum = verts[tris].sum(axis=1)  # p1+p2+p3 per triangle
    num_bvs = 1
    # explicit stack, left subtree built before right one (same node numbering as the recursion)
    stack = [(0, 0, nt)]
    while stack:
        bv_id, first, num = stack.pop()
        idx = prim[first:first + num]
        axes, center, extent = _fit(verts, tris, idx)
        nd = nodes[bv_id]
        nd["first_primitive"] = first
        nd["num_primitives"] = num
        nd["obb_axes"] = axes.T.reshape(-1)  # column-major
        nd["obb_To"] = center
        nd["obb_extent"] = extent
        nd["rss_axes"] = axes.T.reshape(-1)
        nd["rss_Tr"] = center - axes[:, 0] * extent[0] - axes[:, 1] * extent[1]
        nd["rss_length"] = (2 * extent[0], 2 * extent[1])
        nd["rss_radius"] = extent[2]
        if num == 1:
            nd["first_child"] = -(int(idx[0]) + 1)
            continue
        nd["first_child"] = num_bvs
        left, right = num_bvs, num_bvs + 1
        num_bvs += 2
        split_vector = axes[:, 0]
        split_value = vsum[idx].sum(axis=0) @ split_vector / (3 * num)
        right_side = centroids[idx] @ split_vector > split_value
        # the reference's in-place swap loop (BVH_model.cpp:917-948)
        cur = idx.copy()
        c1 = 0
        for i in range(num):
            if not right_side[i]:
                cur[i], cur[c1] = cur[c1], cur[i]
                # keep right_side aligned with cur for the elements not yet visited: the loop only
                # ever swaps position i (being visited) with c1 <= i, whose flag is not read again
                c1 += 1
        if c1 == 0 or c1 == num:
            c1 = num // 2
        prim[first:first + num] = cur
        stack.append((right, first + c1, num - c1))
        stack.append((left, first, c1))
    return nodes, prim.astype(np.uint32)


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
