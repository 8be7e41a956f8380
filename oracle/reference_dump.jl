# reference_dump.jl — run by whoever has Julia + RayTracing.jl + Gridap installed, to turn
# the oracle's "parity unpinned" status into a pinned one.  Uses only the reference's public
# API (TrackGenerator / trace! / segmentize!) and dumps, for one (mesh, nφ, δ):
#   tracks.csv    uid, azim_idx, px, py, qx, qy, phi, ell, A, B, C          (17 significant digits)
#   segments.csv  uid, k, element, px, py, qx, qy, ell
#   volumes.csv   cell, volume
# Compare with:  python tools/compare_reference_dump.py <dir> --n-azim N --delta D
#
#   julia oracle/reference_dump.jl raytracing.jl_amd/data/pincell.json 8 0.02 out_dir
using RayTracing, Gridap, Printf

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
