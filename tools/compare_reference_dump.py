"""Compare a dump made by oracle/reference_dump.jl (real RayTracing.jl) with the oracle, fed the dumped tracks (so libm
differences in trace! do not enter).  Reports the records (counts, element ids, coordinates bit for bit) and, each on its
own, the four assumptions SURVEY.md §9 lists about third-party behaviour: (1) node→cells order, (2) nn / knn incl. exactly
equidistant nodes, (3) Base.isapprox defaults, (4) StaticArrays' 3x3 solve and norm.  One Julia run anywhere settles them.

    python tools/compare_reference_dump.py <dir> [--mesh pincell.json]
    python tools/compare_reference_dump.py --self-test     # the plumbing, on a dump written from the oracle itself"""
import argparse
import math
import os
import sys
import tempfile

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

TOL = 1.4901161193847656e-8


def isapprox(a, b, atol=0.0):
    """Base.isapprox for scalars: x == y || (isfinite(x) && isfinite(y) && |x − y| <= max(atol, rtol·max(|x|, |y|))), rtol = √eps iff atol == 0."""
    if a == b:
        return True
    if not (math.isfinite(a) and math.isfinite(b)):
        return False
    rtol = TOL if atol == 0 else 0.0
    return abs(a - b) <= max(atol, rtol * max(abs(a), abs(b)))


def isapprox_vec(u, v):
    """Array form: ‖u − v‖₂ <= √eps·max(‖u‖₂, ‖v‖₂) (non-finite norms: element-wise)."""
    d = math.sqrt(sum((p - q) ** 2 for p, q in zip(u, v))) if all(map(math.isfinite, list(u) + list(v))) else float("nan")
    if math.isfinite(d):
        return d <= TOL * max(math.sqrt(sum(p * p for p in u)), math.sqrt(sum(q * q for q in v)))
    return all(isapprox(p, q) for p, q in zip(u, v))


def lambdas(x1, y1, x2, y2, x3, y3, x, y):
    """λ = [x1 x2 x3; y1 y2 y3; 1 1 1] \\ [x, y, 1] as the restatement evaluates it (closed form, rt_device.hpp:181-189)."""
    d = x1 * (y2 - y3) + y1 * (x3 - x2) + (x2 * y3 - y2 * x3)
    return (((y2 - y3) * x + (x3 - x2) * y + (x2 * y3 - x3 * y2)) / d, ((y3 - y1) * x + (x1 - x3) * y + (x3 * y1 - x1 * y3)) / d,
            ((y1 - y2) * x + (x2 - x1) * y + (x1 * y2 - x2 * y1)) / d)


def compare(dirname, mesh_name):
    import raytracing_jl_amd as rt
    from oracle import oracle as orc

    model = rt.DiscreteModelFromFile(rt.data_path(mesh_name)) if mesh_name.endswith(".json") else rt.GmshDiscreteModel(rt.data_path(mesh_name))
    mesh = rt.Mesh(model)
    om = orc.OracleMesh.from_mesh(mesh)
    ok = True
    path = lambda f: os.path.join(dirname, f)
    # ---- the records
    tr = np.loadtxt(path("tracks.csv"), delimiter=",", ndmin=2)
    sg = np.loadtxt(path("segments.csv"), delimiter=",", ndmin=2)
    r = om.segmentize(tr[:, 2], tr[:, 3], tr[:, 6], tr[:, 8], tr[:, 9], tr[:, 10], tr[:, 7])
    counts = np.bincount(sg[:, 0].astype(int) - 1, minlength=len(tr))
    print("tracks:", len(tr), "reference segments:", len(sg), "oracle segments:", r["total"])
    same_counts = np.array_equal(counts, np.diff(r["offsets"]))
    print("per-track counts equal:", same_counts)
    ok &= same_counts
    if len(sg) == r["total"]:
        e = np.array_equal(sg[:, 2].astype(np.int32), r["element"])
        print("element ids equal:", e)
        ok &= e
        for j, k in enumerate(("px", "py", "qx", "qy", "ell")):
            d = np.abs(sg[:, 3 + j] - r[k])
            rel = d / np.maximum(np.abs(r[k]), 1e-300)
            print(k, "max abs diff", d.max(), "max rel diff", rel.max(), "bitwise equal:", bool((d == 0).all()))
            ok &= bool(rel.max() <= 1e-10)
    if os.path.exists(path("volumes.csv")):
        vol = np.loadtxt(path("volumes.csv"), delimiter=",", ndmin=2)[:, 1]
        azim = tr[:, 1].astype(np.int32)
        # δs per azimuthal index from the dumped tracks is not available; volumes are compared through Σ only
        print("Σ volumes (reference):", vol.sum())
    # ---- §9.1 node -> cells order
    if os.path.exists(path("node_cells.csv")):
        bad = 0
        with open(path("node_cells.csv")) as f:
            for line in f:
                v = [int(t) for t in line.strip().split(",")]
                n, cells = v[0], v[1:]
                mine = list(mesh.node_cells_data[mesh.node_cells_ptrs[n - 1] - mesh.node_cells_ptrs[0]: mesh.node_cells_ptrs[n] - mesh.node_cells_ptrs[0]])
                bad += cells != [int(c) for c in mine]
        print("§9.1 node→cells lists in the reference's stored order equal to this package's (ascending cell id):", bad == 0, f"({bad} nodes differ)")
        ok &= bad == 0
    # ---- §9.2 nn / knn / find_element, ties reported on their own
    if os.path.exists(path("nn_probes.csv")):
        n_nn = n_tie = n_knn = n_fe = total = 0
        x, y = np.asarray(mesh.x), np.asarray(mesh.y)
        with open(path("nn_probes.csv")) as f:
            for line in f:
                t = line.strip().split(",")
                px, py, nn_ref = float(t[0]), float(t[1]), int(t[2])
                knn_ref = [int(v) for v in t[3].split(";") if v]
                e2, e5 = int(t[4]), int(t[5])
                total += 1
                nn_me = om.nn(px, py)
                if nn_me != nn_ref:
                    d_me = (x[nn_me - 1] - px) ** 2 + (y[nn_me - 1] - py) ** 2
                    d_ref = (x[nn_ref - 1] - px) ** 2 + (y[nn_ref - 1] - py) ** 2
                    if d_me == d_ref:
                        n_tie += 1  # exactly equidistant: NearestNeighbors picked the other one (the oracle ranks by (d², id))
                    else:
                        n_nn += 1
                elif [int(v) for v in om.knn(px, py, 5, skip=nn_ref)] != knn_ref:
                    d = [(x[i - 1] - px) ** 2 + (y[i - 1] - py) ** 2 for i in knn_ref]
                    if sorted(knn_ref) == sorted(int(v) for v in om.knn(px, py, 5, skip=nn_ref)) and len(set(d)) < len(d):
                        n_tie += 1
                    else:
                        n_knn += 1
                if om.find_element(px, py, 2) != e2 or om.find_element(px, py, 5) != e5:
                    n_fe += 1
        print(f"§9.2 nn / knn on {total} probes: {n_nn} different nearest nodes, {n_knn} different knn lists, {n_tie} differences on EXACT ties "
              f"(tie order unspecified in NearestNeighbors), find_element differs on {n_fe}")
        ok &= n_nn == 0 and n_knn == 0
        if n_fe:
            print("     (find_element differences on tie probes change element ids only where two cells both pass at the same point)")
    # ---- §9.3 isapprox defaults
    if os.path.exists(path("isapprox.csv")):
        bad = 0
        for a, b, r0, r1, r2 in np.loadtxt(path("isapprox.csv"), delimiter=",", ndmin=2):
            bad += (isapprox(a, b) != bool(r0)) + (isapprox(a, b, atol=1e-8) != bool(r1)) + (isapprox_vec([a, b], [b, a]) != bool(r2))
        print("§9.3 Base.isapprox (scalar, atol form, array form) as restated:", bad == 0, f"({bad} probe results differ)")
        ok &= bad == 0
    # ---- §9.4 3x3 solve and norm
    if os.path.exists(path("solve_probes.csv")):
        cn = np.asarray(mesh.cell_nodes).reshape(-1, 3) - 1
        x, y = np.asarray(mesh.x), np.asarray(mesh.y)
        n_bits = n_in = n_norm = total = 0
        worst = 0.0
        for c, px, py, inside, l1, l2, l3, nrm in np.loadtxt(path("solve_probes.csv"), delimiter=",", ndmin=2):
            c = int(c) - 1
            lam = lambdas(x[cn[c, 0]], y[cn[c, 0]], x[cn[c, 1]], y[cn[c, 1]], x[cn[c, 2]], y[cn[c, 2]], px, py)
            total += 1
            if lam != (l1, l2, l3):
                n_bits += 1
                worst = max(worst, max(abs(p - q) for p, q in zip(lam, (l1, l2, l3))))
            n_in += om.point_in_triangle(c + 1, px, py) != bool(inside)
            n_norm += math.sqrt(px * px + py * py) != nrm
        print(f"§9.4 StaticArrays 3x3 solve on {total} probes: λ differ in their bits on {n_bits} (largest difference {worst:.1e}), "
              f"point_in_triangle differs on {n_in}; norm(SVector) != sqrt(x² + y²) on {n_norm}")
        ok &= n_in == 0 and n_norm == 0
    print("ALL PINNED" if ok else "DIFFERENCES — see above")
    return ok


def write_oracle_dump(dirname, mesh_name, n_azim, delta):
    """The dump's files, written from the oracle and this package's mesh — only to test the comparison's plumbing."""
    import raytracing_jl_amd as rt
    from oracle import oracle as orc

    model = rt.DiscreteModelFromFile(rt.data_path(mesh_name))
    tg = rt.TrackGenerator(model, n_azim, delta)
    rt.trace(tg)
    mesh = tg.mesh
    om = orc.OracleMesh.from_mesh(mesh)
    r = om.segmentize(tg.px, tg.py, tg.phi, tg.A, tg.B, tg.C, tg.ell)
    g = lambda v: "%.17g" % v
    with open(os.path.join(dirname, "tracks.csv"), "w") as f:
        for u in range(tg.n_total_tracks):
            f.write(",".join([str(u + 1), str(tg.azim_idx[u])] + [g(a[u]) for a in (tg.px, tg.py, tg.qx, tg.qy, tg.phi, tg.ell, tg.A, tg.B, tg.C)]) + "\n")
    with open(os.path.join(dirname, "segments.csv"), "w") as f:
        for u in range(tg.n_total_tracks):
            for k, s in enumerate(range(r["offsets"][u], r["offsets"][u + 1])):
                f.write(",".join([str(u + 1), str(k + 1), str(r["element"][s])] + [g(r[q][s]) for q in ("px", "py", "qx", "qy", "ell")]) + "\n")
    p0 = mesh.node_cells_ptrs[0]
    with open(os.path.join(dirname, "node_cells.csv"), "w") as f:
        for n in range(len(mesh.x)):
            f.write(",".join([str(n + 1)] + [str(int(c)) for c in mesh.node_cells_data[mesh.node_cells_ptrs[n] - p0: mesh.node_cells_ptrs[n + 1] - p0]]) + "\n")
    cn = np.asarray(mesh.cell_nodes).reshape(-1, 3) - 1
    x, y = np.asarray(mesh.x), np.asarray(mesh.y)
    with open(os.path.join(dirname, "nn_probes.csv"), "w") as f:
        for c in range(60):
            for px, py in (((x[cn[c, 0]] + x[cn[c, 1]]) / 2, (y[cn[c, 0]] + y[cn[c, 1]]) / 2), (x[cn[c]].sum() / 3, y[cn[c]].sum() / 3), (x[cn[c, 2]], y[cn[c, 2]])):
                nn_id = om.nn(px, py)
                f.write(",".join([g(px), g(py), str(nn_id), ";".join(str(int(v)) for v in om.knn(px, py, 5, skip=nn_id)),
                                  str(om.find_element(px, py, 2)), str(om.find_element(px, py, 5))]) + "\n")
    with open(os.path.join(dirname, "isapprox.csv"), "w") as f:
        for a, b in ((1.0, 1.0 + 1e-8), (1.0, 1.0 + 2e-8), (0.0, 1e-9), (0.0, 1.0000000000000002e-8), (1e8, 1e8 + 2.0), (math.inf, math.inf)):
            f.write(",".join([g(a), g(b), str(int(isapprox(a, b))), str(int(isapprox(a, b, atol=1e-8))), str(int(isapprox_vec([a, b], [b, a])))]) + "\n")
    with open(os.path.join(dirname, "solve_probes.csv"), "w") as f:
        for c in range(60):
            x1, y1, x2, y2, x3, y3 = x[cn[c, 0]], y[cn[c, 0]], x[cn[c, 1]], y[cn[c, 1]], x[cn[c, 2]], y[cn[c, 2]]
            for t, s in ((0.3, 0.0), (0.6, -TOL), (0.6, -2e-8), (0.25, 0.25)):
                px, py = x1 + t * (x2 - x1) + s * (x3 - x1), y1 + t * (y2 - y1) + s * (y3 - y1)
                lam = lambdas(x1, y1, x2, y2, x3, y3, px, py)
                f.write(",".join([str(c + 1), g(px), g(py), str(int(om.point_in_triangle(c + 1, px, py)))] + [g(v) for v in lam] + [g(math.sqrt(px * px + py * py))]) + "\n")


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("dir", nargs="?")
    ap.add_argument("--mesh", default="pincell.json")
    ap.add_argument("--self-test", action="store_true")
    a = ap.parse_args()
    if a.self_test:
        with tempfile.TemporaryDirectory() as d:
            write_oracle_dump(d, "pincell.json", 8, 0.02)
            sys.exit(0 if compare(d, "pincell.json") else 1)
    if not a.dir:
        ap.error("give the dump directory (or --self-test)")
    sys.exit(0 if compare(a.dir, a.mesh) else 1)


if __name__ == "__main__":
    main()
