# reference_dump.jl — run by whoever has Julia + RayTracing.jl + Gridap installed, to turn
# the oracle's "parity unpinned" status into a pinned one.  Uses the reference's public API
# (TrackGenerator / trace! / segmentize!) for the records and, for the four assumptions SURVEY.md §9 lists
# (items 1–4), the reference's own fields and the packages it calls, and dumps, for one (mesh, nφ, δ):
#   tracks.csv        uid, azim_idx, px, py, qx, qy, phi, ell, A, B, C          (17 significant digits)
#   segments.csv      uid, k, element, px, py, qx, qy, ell
#   volumes.csv       cell, volume
#   node_cells.csv    §9.1  node, its cells in the stored order of mesh.node_cells (src/mesh.jl:27)
#   nn_probes.csv     §9.2  x, y, nn id, knn(k=5, sorted, skipping nn) ids, find_element(x), find_element(x, 5)
#                           probes: edge midpoints (two equidistant nodes), circumcentres (three), nodes, centroids
#   isapprox.csv      §9.3  a, b, isapprox(a, b), isapprox(a, b; atol=1e-8), isapprox([a, b], [b, a])
#   solve_probes.csv  §9.4  cell, x, y, point_in_triangle, λ1, λ2, λ3 (R \ r of src/mesh.jl:166-168), norm(SVector(x, y))
# Compare with:  python tools/compare_reference_dump.py <dir> [--mesh pincell.json]
#
#   julia oracle/reference_dump.jl raytracing.jl_amd/data/pincell.json 8 0.02 out_dir
using RayTracing, Gridap, Printf, StaticArrays, LinearAlgebra, NearestNeighbors
using Gridap.Geometry: get_grid, get_node_coordinates

jsonfile, nφ, δ, outdir = ARGS[1], parse(Int, ARGS[2]), parse(Float64, ARGS[3]), ARGS[4]
mkpath(outdir)
model = DiscreteModelFromFile(jsonfile)
tg = TrackGenerator(model, nφ, δ)
trace!(tg)
segmentize!(tg)
open(joinpath(outdir, "tracks.csv"), "w") do io
    for t in tg.tracks_by_uid
        @printf(io, "%d,%d,%.17g,%.17g,%.17g,%.17g,%.17g,%.17g,%.17g,%.17g,%.17g\n", t.uid, t.azim_idx,
                t.p[1], t.p[2], t.q[1], t.q[2], t.ϕ, t.ℓ, t.ABC[1], t.ABC[2], t.ABC[3])
    end
end
open(joinpath(outdir, "segments.csv"), "w") do io
    for t in tg.tracks_by_uid, (k, s) in enumerate(t.segments)
        @printf(io, "%d,%d,%d,%.17g,%.17g,%.17g,%.17g,%.17g\n", t.uid, k, s.element, s.p[1], s.p[2], s.q[1], s.q[2], s.ℓ)
    end
end
open(joinpath(outdir, "volumes.csv"), "w") do io
    for (i, v) in enumerate(tg.volumes)
        @printf(io, "%d,%.17g\n", i, v)
    end
end

# ---- SURVEY §9 items 1-4, each on its own
mesh = tg.mesh
coords = get_node_coordinates(get_grid(mesh.model))
ncells = length(mesh.cell_nodes)
open(joinpath(outdir, "node_cells.csv"), "w") do io              # §9.1
    for n in 1:length(mesh.node_cells)
        println(io, join(vcat(n, collect(Int, mesh.node_cells[n])), ","))
    end
end
probes = Vector{NTuple{2,Float64}}()
for c in 1:min(ncells, 400)
    ids = mesh.cell_nodes[c]
    p = [coords[ids[i]] for i in 1:3]
    for (i, j) in ((1, 2), (2, 3), (3, 1))                        # two exactly equidistant nodes (up to rounding of the midpoint)
        push!(probes, ((p[i][1] + p[j][1]) / 2, (p[i][2] + p[j][2]) / 2))
    end
    push!(probes, ((p[1][1] + p[2][1] + p[3][1]) / 3, (p[1][2] + p[2][2] + p[3][2]) / 3))
    ax, ay, bx, by, cx, cy = p[1][1], p[1][2], p[2][1], p[2][2], p[3][1], p[3][2]
    d = 2 * (ax * (by - cy) + bx * (cy - ay) + cx * (ay - by))
    if d != 0                                                     # circumcentre: three equidistant nodes
        ux = ((ax^2 + ay^2) * (by - cy) + (bx^2 + by^2) * (cy - ay) + (cx^2 + cy^2) * (ay - by)) / d
        uy = ((ax^2 + ay^2) * (cx - bx) + (bx^2 + by^2) * (ax - cx) + (cx^2 + cy^2) * (bx - ax)) / d
        push!(probes, (ux, uy))
    end
    push!(probes, (p[1][1], p[1][2]))
end
open(joinpath(outdir, "nn_probes.csv"), "w") do io              # §9.2
    for (x, y) in probes
        pt = SVector(x, y)
        nn_id, _ = nn(mesh.kdtree, pt)
        ids, _ = knn(mesh.kdtree, pt, 5, true, i -> isequal(i, nn_id))
        e2 = RayTracing.find_element(mesh, RayTracing.Point2D(x, y))
        e5 = RayTracing.find_element(mesh, RayTracing.Point2D(x, y), 5)
        @printf(io, "%.17g,%.17g,%d,%s,%d,%d\n", x, y, nn_id, join(ids, ";"), e2, e5)
    end
end
open(joinpath(outdir, "isapprox.csv"), "w") do io               # §9.3
    for (a, b) in ((1.0, 1.0 + 1e-8), (1.0, 1.0 + 2e-8), (1.0, 1.0 + 1.4901161193847656e-8), (0.0, 1e-9), (0.0, 1e-8), (0.0, 1.0000000000000002e-8),
                   (1e-300, 0.0), (1e8, 1e8 + 1.0), (1e8, 1e8 + 2.0), (-3.0, -3.0 - 4e-8), (Inf, Inf), (NaN, NaN), (1.6, 1.6 - 1e-8))
        @printf(io, "%.17g,%.17g,%d,%d,%d\n", a, b, isapprox(a, b), isapprox(a, b; atol=1e-8), isapprox([a, b], [b, a]))
    end
end
open(joinpath(outdir, "solve_probes.csv"), "w") do io           # §9.4
    for c in 1:min(ncells, 300)
        ids = mesh.cell_nodes[c]
        x1, y1 = coords[ids[1]]; x2, y2 = coords[ids[2]]; x3, y3 = coords[ids[3]]
        for (t, s) in ((0.3, 0.0), (0.3, 1e-9), (0.3, -1e-9), (0.6, 1.4901161193847656e-8), (0.6, -1.4901161193847656e-8), (0.6, -2e-8), (0.25, 0.25))
            # a point on / just off edge (1, 2), moved towards vertex 3 by the barycentric amount s
            x = x1 + t * (x2 - x1) + s * (x3 - x1); y = y1 + t * (y2 - y1) + s * (y3 - y1)
            R = @SMatrix [x1 x2 x3; y1 y2 y3; 1 1 1]
            r = @SVector [x, y, 1]
            λ = R \ r
            inside = RayTracing.point_in_triangle(mesh, ids, RayTracing.Point2D(x, y))
            @printf(io, "%d,%.17g,%.17g,%d,%.17g,%.17g,%.17g,%.17g\n", c, x, y, inside, λ[1], λ[2], λ[3], norm(SVector(x, y)))
        end
    end
end
