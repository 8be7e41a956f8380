# RayTracingAMD.jl — thin `ccall` shim that routes RayTracing.jl's `segmentize!` to the
# MI355X-native library (include/rt_segmentize.h).  Drop-in for the hot path only: the
# `TrackGenerator` / `trace!` API, the `Track` / `Segment` layout and `t.volumes` are what
# RayTracing.jl defines; NeutronTransport.jl consumes the result unchanged.
#
# NOT EXECUTED in the build container or on the GPU box (no Julia there); it mirrors, call
# for call, the tested Python/ctypes binding in raytracing.jl_amd/_capi.py.  Field names of
# Gridap's `Table` (`data`, `ptrs`) are as of Gridap 0.19.
module RayTracingAMD

using RayTracing
using RayTracing: TrackGenerator, Track, Segment, Point2D, nazim2
using Gridap: get_grid, get_node_coordinates

const LIB = get(ENV, "RT_SEGMENTIZE_LIB", "librt_segmentize.so")

lasterror() = unsafe_string(ccall((:rt_last_error, LIB), Cstring, ()))

"""
    segmentize_amd!(t::TrackGenerator{Float64}; k=5, rtol=Base.rtoldefault(Float64), device=0)

Same contract as `RayTracing.segmentize!` (src/trackgenerator.jl:357-369): requires `trace!`,
refills every `track.segments` in march order, overwrites `t.volumes`, returns `t`, and
throws the reference's `ErrorException`s for point-location failure and Σℓ mismatch.
"""
function segmentize_amd!(t::TrackGenerator{Float64}; k::Int=5, rtol::Real=Base.rtoldefault(Float64),
                         device::Int=0)
    tracks = t.tracks_by_uid
    !isassigned(tracks, 1) && error("Segmentation is intended after tracing. Please, " *
                                    "call `trace!` first!")
    mesh = t.mesh
    # ---- flatten the mesh (src/mesh.jl:10-31) to the SoA arrays rt_mesh_create takes
    coords = get_node_coordinates(get_grid(mesh.model))
    x = Float64[c[1] for c in coords]
    y = Float64[c[2] for c in coords]
    cell_nodes = Vector{Int32}(mesh.cell_nodes.data)            # 3 per cell, 1-based
    nc_ptrs = Vector{Int32}(mesh.node_cells.ptrs)               # 1-based CSR offsets (accepted as is)
    nc_data = Vector{Int32}(mesh.node_cells.data)
    bb = Float64[mesh.bb_min[1], mesh.bb_min[2], mesh.bb_max[1], mesh.bb_max[2]]
    n_nodes, n_cells = Int32(length(x)), Int32(length(cell_nodes) ÷ 3)
    hm = ccall((:rt_mesh_create, LIB), Ptr{Cvoid},
               (Int32, Ptr{Float64}, Ptr{Float64}, Int32, Ptr{Int32}, Int32, Ptr{Int32}, Ptr{Int32}, Ptr{Float64}),
               device, x, y, n_nodes, cell_nodes, n_cells, nc_ptrs, nc_data, bb)
    hm == C_NULL && error("rt_mesh_create: " * lasterror())
    ht = C_NULL
    try
        # ---- per-track inputs in uid order (src/track.jl:42-54); cos/sin by the host libm,
        #      exactly the values advance_step (src/point.jl:43) would use
        n = length(tracks)
        px = Float64[tr.p[1] for tr in tracks]; py = Float64[tr.p[2] for tr in tracks]
        ϕ = Float64[tr.ϕ for tr in tracks]
        cϕ = cos.(ϕ); sϕ = sin.(ϕ)
        A = Float64[tr.ABC[1] for tr in tracks]; B = Float64[tr.ABC[2] for tr in tracks]
        C = Float64[tr.ABC[3] for tr in tracks]
        ℓ = Float64[tr.ℓ for tr in tracks]
        azim = Int32[tr.azim_idx for tr in tracks]
        ht = ccall((:rt_tracks_create, LIB), Ptr{Cvoid},
                   (Ptr{Cvoid}, Int64, Ptr{Float64}, Ptr{Float64}, Ptr{Float64}, Ptr{Float64}, Ptr{Float64},
                    Ptr{Float64}, Ptr{Float64}, Ptr{Float64}, Ptr{Float64}, Ptr{Int32}),
                   hm, n, px, py, ϕ, cϕ, sϕ, A, B, C, ℓ, azim)
        ht == C_NULL && error("rt_tracks_create: " * lasterror())
        δs = t.azimuthal_quadrature.δs
        total = ccall((:rt_segmentize, LIB), Int64,
                      (Ptr{Cvoid}, Float64, Int32, Float64, Ptr{Float64}, Int32),
                      ht, t.tiny_step, k, rtol, δs, nazim2(t.azimuthal_quadrature))
        total < 0 && error("rt_segmentize: " * lasterror())
        # ---- the reference throws on the first failing track; so do we, with its text
        nfail = Ref{Int64}(0); uid = Ref{Int64}(0); st = Ref{Int32}(0)
        ccall((:rt_failed_tracks, LIB), Int32, (Ptr{Cvoid}, Ref{Int64}, Ref{Int64}, Ref{Int32}), ht, nfail, uid, st)
        if nfail[] > 0
            msg = unsafe_string(ccall((:rt_status_message, LIB), Cstring, (Int32,), st[]))
            error(replace(msg, "%d" => string(uid[])))
        end
        # ---- fetch SoA results and rebuild Vector{Segment} per track (5-arg ctor, src/segment.jl:23-29)
        offs = Vector{Int64}(undef, n + 1); status = Vector{Int32}(undef, n)
        ccall((:rt_fetch_offsets, LIB), Int32, (Ptr{Cvoid}, Ptr{Int64}, Ptr{Int32}), ht, offs, status)
        # page-locked buffers owned by the handle: the 44 B/segment arrive at the PCIe rate instead of
        # page-faulting into fresh Julia arrays (C3: 7 ms instead of 25-40 ms); read-only views, copied
        # into the Segments below, gone with the handle
        hp = Vector{Ptr{Cvoid}}(undef, 6)
        rc = ccall((:rt_fetch_segments_pinned, LIB), Int32, (Ptr{Cvoid}, Ptr{Ptr{Cvoid}}), ht, hp)
        rc != 0 && error("rt_fetch_segments_pinned: " * lasterror())
        spx = unsafe_wrap(Array, Ptr{Float64}(hp[1]), total); spy = unsafe_wrap(Array, Ptr{Float64}(hp[2]), total)
        sqx = unsafe_wrap(Array, Ptr{Float64}(hp[3]), total); sqy = unsafe_wrap(Array, Ptr{Float64}(hp[4]), total)
        sℓ = unsafe_wrap(Array, Ptr{Float64}(hp[5]), total); sel = unsafe_wrap(Array, Ptr{Int32}(hp[6]), total)
        Threads.@threads for u in 1:n
            segs = tracks[u].segments
            empty!(segs)
            sizehint!(segs, offs[u+1] - offs[u])
            for s in (offs[u]+1):offs[u+1]
                push!(segs, Segment(Point2D(spx[s], spy[s]), Point2D(sqx[s], sqy[s]), sℓ[s], Float64[], sel[s]))
            end
        end
        ccall((:rt_fetch_volumes, LIB), Int32, (Ptr{Cvoid}, Ptr{Float64}), ht, t.volumes)
    finally
        ht != C_NULL && ccall((:rt_tracks_destroy, LIB), Cvoid, (Ptr{Cvoid},), ht)
        ccall((:rt_mesh_destroy, LIB), Cvoid, (Ptr{Cvoid},), hm)
    end
    return t
end

# Opt-in replacement of the reference entry point:  RayTracingAMD.install!()
function install!()
    @eval RayTracing segmentize!(t::TrackGenerator{Float64}; k::Int=5, rtol::Real=Base.rtoldefault(Float64)) =
        $(segmentize_amd!)(t; k=k, rtol=rtol)
    return nothing
end

end # module
