"""Deterministic BWR-like mesh fixture (substitute for the mesh `test/bwr-gmsh.jl` generates).

The reference does not ship its BWR mesh: `test/bwr-gmsh.jl:64-137` builds it by driving gmsh,
which is unavailable here.  This script builds a substitute with the same geometry parameters
(`test/bwr-gmsh.jl:55-62,75-104`): a 4×4 lattice of pins, pitch 1.6, fuel radius 0.5, clad
thickness 0.1 (outer radius 0.6), domain [0, 6.4]², target edge length 0.1.  Points are placed on
both circles of every pin, on the boundary at 0.1 spacing, and on a jittered interior lattice
(fixed seed), then triangulated with scipy's Delaunay.  Material conformity is irrelevant to
segmentize!; what matters is an irregular, unstructured connectivity larger than the pincell's.

Writes raytracing.jl_amd/data/bwr_like.msh (gmsh 4.1 ASCII; cells keep the generator's order and
each cell's node ids are ascending, like Gridap's oriented grids).  Every number it prints is a
SUBSTITUTE-mesh number, not a number of the reference's own BWR mesh.
"""
import os

import numpy as np
from scipy.spatial import Delaunay

N, PITCH, RI, RO, LC = 4, 1.6, 0.5, 0.6, 0.1
S = N * PITCH
OUT = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "raytracing.jl_amd", "data", "bwr_like.msh")


def build():
    rng = np.random.default_rng(20250718)
    pts = []
    # boundary (corners included), exact coordinates on the sides
    nb = int(round(S / LC))
    t = np.arange(nb) * (S / nb)
    pts += [(v, 0.0) for v in t] + [(S, v) for v in t] + [(S - v, S) for v in t] + [(0.0, S - v) for v in t]
    fixed = len(pts)
    # circles of every pin
    for i in range(N):
        for j in range(N):
            cx, cy = PITCH / 2 + i * PITCH, PITCH / 2 + j * PITCH
            for r in (RI, RO):
                n = int(round(2 * np.pi * r / LC))
                a = 2 * np.pi * (np.arange(n) + 0.5 * (r == RO)) / n
                pts += list(zip(cx + r * np.cos(a), cy + r * np.sin(a)))
            pts.append((cx, cy))
    ring = np.array(pts)
    # jittered hexagonal interior lattice
    h = LC
    rows = int(S / (h * np.sqrt(3) / 2))
    cand = []
    for r_ in range(1, rows):
        y = r_ * h * np.sqrt(3) / 2
        off = 0.5 * h if r_ % 2 else 0.0
        for x in np.arange(h / 2 + off, S - h / 4, h):
            cand.append((x, y))
    cand = np.array(cand)
    cand += rng.uniform(-0.22 * h, 0.22 * h, cand.shape)
    keep = (cand[:, 0] > 0.6 * h) & (cand[:, 0] < S - 0.6 * h) & (cand[:, 1] > 0.6 * h) & (cand[:, 1] < S - 0.6 * h)
    cand = cand[keep]
    # drop lattice points closer than 0.62 h to a fixed (boundary / circle) point
    from scipy.spatial import cKDTree

    d, _ = cKDTree(ring).query(cand)
    cand = cand[d > 0.62 * h]
    xy = np.vstack((ring, cand))
    tri = Delaunay(xy, qhull_options="Qbb Qc Qz Q12")
    cells = tri.simplices.astype(np.int64)
    x, y = xy[:, 0], xy[:, 1]
    area2 = (x[cells[:, 1]] - x[cells[:, 0]]) * (y[cells[:, 2]] - y[cells[:, 0]]) - \
            (x[cells[:, 2]] - x[cells[:, 0]]) * (y[cells[:, 1]] - y[cells[:, 0]])
    cells = cells[np.abs(area2) > 1e-9]  # collinear boundary points can yield flat hull triangles
    cells = np.sort(cells, axis=1) + 1
    # keep only nodes that are used (all should be)
    used = np.zeros(len(xy), bool)
    used[cells.reshape(-1) - 1] = True
    assert used.all(), "unused nodes"
    return xy, cells, fixed


def write_msh(xy, cells, nb_nodes):
    n, m = len(xy), len(cells)
    # boundary line elements: consecutive boundary nodes (first nb_nodes points run around the square)
    lines = [(i + 1, (i + 1) % nb_nodes + 1) for i in range(nb_nodes)]
    per_side = nb_nodes // 4
    with open(OUT, "w") as f:
        f.write("$MeshFormat\n4.1 0 8\n$EndMeshFormat\n")
        f.write('$PhysicalNames\n5\n1 1 "bottom"\n1 2 "right"\n1 3 "top"\n1 4 "left"\n2 5 "domain"\n$EndPhysicalNames\n')
        f.write("$Entities\n4 4 1 0\n")
        for k, (px, py) in enumerate(((0, 0), (S, 0), (S, S), (0, S)), 1):
            f.write(f"{k} {px:.17g} {py:.17g} 0 0 \n")
        boxes = ((0, 0, S, 0), (S, 0, S, S), (0, S, S, S), (0, 0, 0, S))
        ends = ((1, 2), (2, 3), (3, 4), (4, 1))
        for k in range(4):
            b = boxes[k]
            f.write(f"{k + 1} {b[0]:.17g} {b[1]:.17g} 0 {b[2]:.17g} {b[3]:.17g} 0 1 {k + 1} 2 {ends[k][0]} -{ends[k][1]} \n")
        f.write(f"1 0 0 0 {S:.17g} {S:.17g} 0 1 5 4 1 2 3 4 \n$EndEntities\n")
        f.write(f"$Nodes\n1 {n} 1 {n}\n2 1 0 {n}\n")
        f.write("\n".join(str(i + 1) for i in range(n)) + "\n")
        f.write("\n".join(f"{p[0]:.17g} {p[1]:.17g} 0" for p in xy) + "\n$EndNodes\n")
        f.write(f"$Elements\n5 {len(lines) + m} 1 {len(lines) + m}\n")
        tag = 1
        for k in range(4):
            f.write(f"1 {k + 1} 1 {per_side}\n")
            for a, b in lines[k * per_side:(k + 1) * per_side]:
                f.write(f"{tag} {a} {b} \n")
                tag += 1
        f.write(f"2 1 2 {m}\n")
        for c in cells:
            f.write(f"{tag} {c[0]} {c[1]} {c[2]} \n")
            tag += 1
        f.write("$EndElements\n")


if __name__ == "__main__":
    xy, cells, nb_nodes = build()
    write_msh(xy, cells, nb_nodes)
    x, y = xy[:, 0], xy[:, 1]
    c = cells - 1
    e = [np.hypot(x[c[:, i]] - x[c[:, (i + 1) % 3]], y[c[:, i]] - y[c[:, (i + 1) % 3]]) for i in range(3)]
    a2 = np.abs((x[c[:, 1]] - x[c[:, 0]]) * (y[c[:, 2]] - y[c[:, 0]]) - (x[c[:, 2]] - x[c[:, 0]]) * (y[c[:, 1]] - y[c[:, 0]]))
    alt = a2 / np.maximum.reduce(e)
    print(f"wrote {OUT}: {len(xy)} nodes, {len(cells)} cells, area {a2.sum() / 2:.12f} (domain {S * S}), "
          f"mean edge {np.mean(e):.4f}, min altitude {alt.min():.4f}")
