"""Inputs of tests/sanitize/run.sh: N seeded fuzz meshes as Gridap JSON models (what rt_msh_load reads) — every shape
class of tests/meshgen.py, plus a mesh with a degenerate cell and one with an edge shared by three cells — and
malformed files (truncated, negative / huge counts, garbage) that the loader must refuse."""
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import meshgen  # noqa: E402
import raytracing_jl_amd as rt  # noqa: E402

out, n = sys.argv[1], int(sys.argv[2])


def write(model_xy, cells, name):
    xy = np.asarray(model_xy, float)
    cells = np.asarray(cells, int)
    d = {"grid": {"node_coordinates": xy.ravel().tolist(),
                  "cell_node_ids": {"ptrs": (1 + 3 * np.arange(len(cells) + 1)).tolist(), "data": cells.ravel().tolist()}}}
    json.dump(d, open(os.path.join(out, name), "w"))


for seed in range(n):
    rng = np.random.default_rng(seed)
    kind = seed % 5
    kw = dict(w=float(rng.choice([1.0, 0.3, 2.5])), h=float(rng.choice([1.0, 0.4])), x0=float(rng.choice([0.0, -3.25, 110.0])),
              y0=float(rng.choice([0.0, 2.5])))
    if kind == 0:
        m = meshgen.random_model(rt, seed, int(rng.integers(30, 600)), **kw)
    elif kind == 1:
        m = meshgen.random_model(rt, seed, int(rng.integers(90, 600)), cluster=True, **kw)
    elif kind == 2:
        k = int(rng.integers(4, 20)); m = meshgen.lattice_model(rt, seed, k, k, **kw)
    elif kind == 3:
        k = int(rng.integers(5, 16)); m = meshgen.sliver_model(rt, seed, k, k, gap=float(rng.choice([1e-3, 1e-6, 1e-9])), **kw)
    else:
        k = int(rng.integers(4, 12)); m = meshgen.lattice_model(rt, seed, k, k, **kw)
        xy = m.node_coordinates.copy(); cells = np.asarray(m.cell_node_ids)
        a, b, c = cells[len(cells) // 2]
        if seed % 2:
            xy[c - 1] = 0.5 * (xy[a - 1] + xy[b - 1])  # a degenerate cell
            write(xy, cells, "fuzz_%04d.json" % seed); continue
        xy = np.vstack([xy, xy[[a - 1, b - 1, c - 1]].mean(axis=0)])  # an edge shared by three cells
        write(xy, np.vstack([cells, np.sort([a, b, len(xy)])]), "fuzz_%04d.json" % seed); continue
    write(m.node_coordinates, m.cell_node_ids, "fuzz_%04d.json" % seed)

good = open(rt.data_path("pincell.msh")).read()
bad = {
    "bad_truncated.msh": good[: len(good) // 3],
    "bad_negative_nodes.msh": good.replace("$Nodes\n", "$Nodes\n-5 -7 1 2\n", 1),
    "bad_huge_nodes.msh": "$MeshFormat\n4.1 0 8\n$EndMeshFormat\n$Nodes\n1 99999999999 1 99999999999\n2 1 0 99999999999\n",
    "bad_huge_block.msh": "$MeshFormat\n4.1 0 8\n$EndMeshFormat\n$Nodes\n1 3 1 3\n2 1 0 4000000000\n1\n2\n3\n0 0 0\n1 0 0\n0 1 0\n$EndNodes\n",
    "bad_huge_elements.msh": "$MeshFormat\n4.1 0 8\n$EndMeshFormat\n$Nodes\n1 3 1 3\n2 1 0 3\n1\n2\n3\n0 0 0\n1 0 0\n0 1 0\n$EndNodes\n$Elements\n1 77777777777 1 1\n2 1 2 88888888888\n",
    "bad_format.msh": "$MeshFormat\n2.2 0 8\n$EndMeshFormat\n",
    "bad_node_ref.msh": "$MeshFormat\n4.1 0 8\n$EndMeshFormat\n$Nodes\n1 3 1 3\n2 1 0 3\n1\n2\n3\n0 0 0\n1 0 0\n0 1 0\n$EndNodes\n$Elements\n1 1 1 1\n2 1 2 1\n1 1 2 9\n$EndElements\n",
    "bad_garbage.msh": "\x00\x01\x02 not a mesh \xff",
    "bad_empty.msh": "",
    "bad_json_unbalanced.json": '{"grid": {"node_coordinates": [0, 0, 1, 0, 0, 1], "cell_node_ids": {"ptrs": [1, 4], "data": [1, 2, 3]',
    "bad_json_quad.json": '{"grid": {"node_coordinates": [0,0,1,0,1,1,0,1], "cell_node_ids": {"ptrs": [1, 5], "data": [1, 2, 3, 4]}}}',
    "bad_json_ref.json": '{"grid": {"node_coordinates": [0,0,1,0,0,1], "cell_node_ids": {"ptrs": [1, 4], "data": [1, 2, 7]}}}',
}
for name, text in bad.items():
    open(os.path.join(out, name), "w", encoding="latin-1").write(text)
print("wrote", n, "fuzz meshes and", len(bad), "malformed files to", out)
