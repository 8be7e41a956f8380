# RayTracingAMD.jl — thin `ccall` shim that routes RayTracing.jl's `segmentize!` to the
# MI355X-native library (include/rt_segmentize.h).  Drop-in for the hot path only: the
# `TrackGenerator` / `trace!` API, the `Track` / `Segment` layout and `t.volumes` are what
# RayTracing.jl defines; NeutronTransport.jl consumes the result unchanged.
#
# NOT EXECUTED in the build container or on the GPU box (no Julia there); it mirrors, call
# for call, the tested Python/ctypes binding in raytracing.jl_amd/_capi.py, and tests/c_abi_smoke.c makes
# the same call sequence with the same argument types from plain C on the GPU box (dlopen by path,
# 1-based CSR ptrs, pinned fetch, "%d" substitution).  Field names of Gridap's `Table` (`data`, `ptrs`)
# are as of Gridap 0.19.
module RayTracingAMD

using RayTracing
using RayTracing: TrackGenerator, Track, Segment, Point2D, nazim2
using Gridap: get_grid, get_node_coordinates

const LIB = get(ENV, "RT_SEGMENTIZE_LIB", "librt_segmentize.so")

lasterror() = unsafe_string(ccall((:rt_last_error, LIB), Cstring, ()))

# ---- the device-resident mesh is kept across calls: rt_mesh_create flattens the mesh, builds the nearest-node grid
#      and the walk records with their certificates (≈12 ms of host work for the pincell mesh, plus the upload) —
#      per mesh, not per segmentize!.  Keyed by the Mesh object (the reference's `Mesh` is an immutable struct, so it
#      cannot carry a finalizer) and the device; released by `release_mesh!(mesh)` or at exit.
const MESH_HANDLES = IdDict{Any,Dict{Int,Ptr{Cvoid}}}()
const MESH_LOCK = ReentrantLock()

function mesh_handle(mesh, device::Int)
    lock(MESH_LOCK) do
        per = get!(() -> Dict{Int,Ptr{Cvoid}}(), MESH_HANDLES, mesh)
        get!(per, device) do
            coords = get_node_coordinates(get_grid(mesh.model))
            x = Float64[c[1] for c in coords]
            y = Float64[c[2] for c in coords]
            cell_nodes = Vector{Int32}(mesh.cell_nodes.data)            # 3 per cell, 1-based
            nc_ptrs = Vector{Int32}(mesh.node_cells.ptrs)               # 1-based CSR offsets (accepted as is)
            nc_data = Vector{Int32}(mesh.node_cells.data)
            bb = Float64[mesh.bb_min[1], mesh.bb_min[2], mesh.bb_max[1], mesh.bb_max[2]]
            hm = ccall((:rt_mesh_create, LIB), Ptr{Cvoid},
                       (Int32, Ptr{Float64}, Ptr{Float64}, Int32, Ptr{Int32}, Int32, Ptr{Int32}, Ptr{Int32}, Ptr{Float64}),
                       device, x, y, Int32(length(x)), cell_nodes, Int32(length(cell_nodes) ÷ 3), nc_ptrs, nc_data, bb)
            hm == C_NULL && error("rt_mesh_create: " * lasterror())
            hm
        end
    end
end

# Track-set handles that outlive their call (`materialize=false`: the SegmentsViews read the handle's pinned arrays) are
# registered under their mesh handle: rt_tracks_destroy reads the rt_mesh behind it, so a mesh is only destroyed after
# every track set made from it — in `release_mesh!` and in the atexit hook, which Julia runs BEFORE the finalizers.
const LIVE_TRACKSETS = Dict{Ptr{Cvoid},Vector{Any}}()   # mesh handle => WeakRefs of TrackSetHandles

function release_mesh!(mesh)
    lock(MESH_LOCK) do
        per = pop!(MESH_HANDLES, mesh, nothing)
        per === nothing && return
        for hm in values(per)
            for w in pop!(LIVE_TRACKSETS, hm, Any[])
                ts = w.value
                ts === nothing || destroy!(ts)
            end
            ccall((:rt_mesh_destroy, LIB), Cvoid, (Ptr{Cvoid},), hm)
        end
    end
    return nothing
end

function __init__()
    atexit() do
        foreach(release_multi!, collect(keys(MULTI_HANDLES)))
        foreach(release_mesh!, collect(keys(MESH_HANDLES)))
    end
end

"""
    SegmentsView <: AbstractVector{Segment{Float64}}

A track's segments read straight from the SoA arrays the library fetched (page-locked, owned by the kept track-set
handle `keep`): `view[i]` builds `Segment(p, q, ℓ, τ, element)` on demand, so a consumer that only iterates
(NeutronTransport.jl's sweep reads `ℓ`, `element` and fills `τ`) pays no per-segment allocation up front.  `τ` vectors
are created on first access and then kept, one per segment, as in the reference (src/segment.jl:14,28).
Opt in with `segmentize_amd!(t; materialize=false)`; the default rebuilds real `Vector{Segment}`s.
"""
struct SegmentsView <: AbstractVector{Segment{Float64}}
    px::Vector{Float64}; py::Vector{Float64}; qx::Vector{Float64}; qy::Vector{Float64}
    ℓ::Vector{Float64}; element::Vector{Int32}
    first::Int; len::Int
    τ::Dict{Int,Vector{Float64}}
    keep::Any
end
Base.size(v::SegmentsView) = (v.len,)
Base.IndexStyle(::Type{SegmentsView}) = IndexLinear()
function Base.getindex(v::SegmentsView, i::Int)
    @boundscheck checkbounds(v, i)
    s = v.first + i - 1
    τ = get!(() -> Float64[], v.τ, i)
    return Segment(Point2D(v.px[s], v.py[s]), Point2D(v.qx[s], v.qy[s]), v.ℓ[s], τ, v.element[s])
end

# keeps the last track-set handle of a generator alive while SegmentsViews of it exist
mutable struct TrackSetHandle
    h::Ptr{Cvoid}
    function TrackSetHandle(h, hm)
        x = new(h)
        lock(MESH_LOCK) do
            push!(get!(() -> Any[], LIVE_TRACKSETS, hm), WeakRef(x))
        end
        finalizer(destroy!, x)
        x
    end
end
function destroy!(y::TrackSetHandle)   # idempotent: release_mesh! / atexit may have been here before the finalizer
    y.h != C_NULL && ccall((:rt_tracks_destroy, LIB), Cvoid, (Ptr{Cvoid},), y.h)
    y.h = C_NULL
    return nothing
end

"""
    segmentize_amd!(t::TrackGenerator{Float64}; k=5, rtol=Base.rtoldefault(Float64), device=0, materialize=true, pinned=!materialize)

Same contract as `RayTracing.segmentize!` (src/trackgenerator.jl:357-369): requires `trace!`,
refills every `track.segments` in march order, overwrites `t.volumes`, returns `t`, and
throws the reference's `ErrorException`s for point-location failure and Σℓ mismatch.
`materialize=false` returns `(t, views)` instead, `views[uid]::SegmentsView` over the fetched SoA arrays, and leaves
`track.segments` untouched (no per-segment allocation; see `SegmentsView`).
"""
function segmentize_amd!(t::TrackGenerator{Float64}; k::Int=5, rtol::Real=Base.rtoldefault(Float64),
                         device::Int=0, materialize::Bool=true, pinned::Bool=!materialize)
    tracks = t.tracks_by_uid
    !isassigned(tracks, 1) && error("Segmentation is intended after tracing. Please, " *
                                    "call `trace!` first!")
    # ---- the mesh handle (src/mesh.jl:10-31 flattened to the SoA arrays rt_mesh_create takes) is cached per mesh
    hm = mesh_handle(t.mesh, device)
    ht = C_NULL
    hr = C_NULL     # rt_result: the host block the library owns for this call's results (round 6)
    views = nothing
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
        # ---- the destination FIRST (eager rebuild without page-locking): a block the library maps, asks huge pages for and faults in
        #      on threads of its own from here on — beside the upload and the kernels below (rt_result_alloc returns at once)
        if !pinned
            hr = ccall((:rt_result_alloc, LIB), Ptr{Cvoid}, (Ptr{Cvoid}, Int64, Float64, Int64), hm, n, sum(ℓ), 0)
        end
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
        # page-locked buffers owned by the handle, all eight arrays in one call and one synchronisation: the 44 B/segment
        # arrive at the PCIe rate instead of page-faulting into fresh Julia arrays (C3: 7 ms instead of 25-40 ms), and the
        # offsets / status no longer cost two synchronous copies of their own (6 ms); read-only views, copied into the
        # Segments below, gone with the handle
        # `pinned=false` (the default of the eager rebuild): plain Julia arrays, filled by rt_fetch_offsets / rt_fetch_segments — pipelined
        # through a small page-locked block inside the library, close to the PCIe rate WITHOUT page-locking the whole result first
        # (the first rt_fetch_pinned of a process pins its buffers: 0.08 ms per MB, 260-550 ms for a BWR assembly's 5 GB)
        local offs, spx, spy, sqx, sqy, sℓ, sel
        if pinned
            hp = Vector{Ptr{Cvoid}}(undef, 8)
            rc = ccall((:rt_fetch_pinned, LIB), Int32, (Ptr{Cvoid}, Ptr{Ptr{Cvoid}}), ht, hp)
            rc != 0 && error("rt_fetch_pinned: " * lasterror())
            offs = unsafe_wrap(Array, Ptr{Int64}(hp[1]), n + 1)
            spx = unsafe_wrap(Array, Ptr{Float64}(hp[3]), total); spy = unsafe_wrap(Array, Ptr{Float64}(hp[4]), total)
            sqx = unsafe_wrap(Array, Ptr{Float64}(hp[5]), total); sqy = unsafe_wrap(Array, Ptr{Float64}(hp[6]), total)
            sℓ = unsafe_wrap(Array, Ptr{Float64}(hp[7]), total); sel = unsafe_wrap(Array, Ptr{Int32}(hp[8]), total)
        elseif hr != C_NULL
            # offsets, status and the six record arrays into the library's block, copied behind the front of its page faults;
            # unsafe_wrap(own = false): Julia never frees the memory — rt_result_free does, in the `finally` below, after the
            # Segments have been built from it
            hp = Vector{Ptr{Cvoid}}(undef, 8); tot = Ref{Int64}(0)
            rc = ccall((:rt_result_fetch, LIB), Int32, (Ptr{Cvoid}, Ptr{Cvoid}, Ptr{Ptr{Cvoid}}, Ref{Int64}), ht, hr, hp, tot)
            rc != 0 && error("rt_result_fetch: " * lasterror())
            offs = unsafe_wrap(Array, Ptr{Int64}(hp[1]), n + 1; own=false)
            spx = unsafe_wrap(Array, Ptr{Float64}(hp[3]), total; own=false); spy = unsafe_wrap(Array, Ptr{Float64}(hp[4]), total; own=false)
            sqx = unsafe_wrap(Array, Ptr{Float64}(hp[5]), total; own=false); sqy = unsafe_wrap(Array, Ptr{Float64}(hp[6]), total; own=false)
            sℓ = unsafe_wrap(Array, Ptr{Float64}(hp[7]), total; own=false); sel = unsafe_wrap(Array, Ptr{Int32}(hp[8]), total; own=false)
        else
            offs = Vector{Int64}(undef, n + 1)
            spx = Vector{Float64}(undef, total); spy = similar(spx); sqx = similar(spx); sqy = similar(spx); sℓ = similar(spx)
            sel = Vector{Int32}(undef, total)
            rc = ccall((:rt_fetch_offsets, LIB), Int32, (Ptr{Cvoid}, Ptr{Int64}, Ptr{Int32}), ht, offs, C_NULL)
            rc != 0 && error("rt_fetch_offsets: " * lasterror())
            rc = ccall((:rt_fetch_segments, LIB), Int32,
                       (Ptr{Cvoid}, Ptr{Float64}, Ptr{Float64}, Ptr{Float64}, Ptr{Float64}, Ptr{Float64}, Ptr{Int32}),
                       ht, spx, spy, sqx, sqy, sℓ, sel)
            rc != 0 && error("rt_fetch_segments: " * lasterror())
        end
        if materialize
            # eager rebuild: real Vector{Segment}s, the reference's layout (src/segment.jl:23-33); one `τ` per segment
            Threads.@threads for u in 1:n
                segs = tracks[u].segments
                cnt = Int(offs[u+1] - offs[u])
                resize!(segs, cnt)                       # one growth per track instead of a push! per segment
                base = Int(offs[u])
                @inbounds for i in 1:cnt
                    s = base + i
                    segs[i] = Segment(Point2D(spx[s], spy[s]), Point2D(sqx[s], sqy[s]), sℓ[s], Float64[], sel[s])
                end
            end
        else
            keep = TrackSetHandle(ht, hm)   # the pinned arrays live as long as a view does
            ht = C_NULL
            first = [Int(offs[u]) + 1 for u in 1:n]; len = [Int(offs[u+1] - offs[u]) for u in 1:n]
            views = [SegmentsView(spx, spy, sqx, sqy, sℓ, sel, first[u], len[u], Dict{Int,Vector{Float64}}(), keep) for u in 1:n]
            ccall((:rt_fetch_volumes, LIB), Int32, (Ptr{Cvoid}, Ptr{Float64}), keep.h, t.volumes)
        end
        materialize && ccall((:rt_fetch_volumes, LIB), Int32, (Ptr{Cvoid}, Ptr{Float64}), ht, t.volumes)
    finally
        ht != C_NULL && ccall((:rt_tracks_destroy, LIB), Cvoid, (Ptr{Cvoid},), ht)
        hr != C_NULL && ccall((:rt_result_free, LIB), Cvoid, (Ptr{Cvoid},), hr)   # (the eager rebuild has copied everything out)
    end
    return materialize ? t : (t, views)
end

"""
    segmentize_amd_multi!(t::TrackGenerator{Float64}; devices=[0], k=5, rtol=Base.rtoldefault(Float64))

`segmentize!` over several GPUs from one Julia process: `tracks_by_uid` is cut into contiguous uid ranges of ≈ equal Σℓ,
one per entry of `devices` (tracks are independent, src/trackgenerator.jl:362-364), every device marches its range, and the
results come back as the arrays of an unsharded run (`rt_multi_*`; tests/c_abi_smoke.c makes the same calls from C).
"""
# The rt_multi handle (N mesh replicas with their preprocessing + the uploaded uid ranges) is kept per TrackGenerator and
# device list, like the single-GPU path keeps its rt_mesh: a second segmentize_amd_multi!(t) only marches.  It is rebuilt
# when the tracks changed (another trace!: different count or Σℓ) and released by `release_multi!(t)` or at exit.
const MULTI_HANDLES = IdDict{Any,Any}()   # t => (devices, n_tracks, Σℓ, handle)

function release_multi!(t)
    lock(MESH_LOCK) do
        e = pop!(MULTI_HANDLES, t, nothing)
        e === nothing || ccall((:rt_multi_destroy, LIB), Cvoid, (Ptr{Cvoid},), e[4])
    end
    return nothing
end

function segmentize_amd_multi!(t::TrackGenerator{Float64}; devices::Vector{Int}=[0], k::Int=5,
                               rtol::Real=Base.rtoldefault(Float64))
    tracks = t.tracks_by_uid
    !isassigned(tracks, 1) && error("Segmentation is intended after tracing. Please, " *
                                    "call `trace!` first!")
    n = length(tracks)
    hm = lock(MESH_LOCK) do
        e = get(MULTI_HANDLES, t, nothing)
        if e !== nothing && e[1] == devices && e[2] == n && e[3] == sum(tr.ℓ for tr in tracks)
            return e[4]
        end
        e === nothing || (pop!(MULTI_HANDLES, t); ccall((:rt_multi_destroy, LIB), Cvoid, (Ptr{Cvoid},), e[4]))
        h = multi_create(t, devices)
        MULTI_HANDLES[t] = (copy(devices), n, sum(tr.ℓ for tr in tracks), h)
        h
    end
    total = ccall((:rt_multi_segmentize, LIB), Int64, (Ptr{Cvoid}, Float64, Int32, Float64, Ptr{Float64}, Int32),
                  hm, t.tiny_step, k, rtol, t.azimuthal_quadrature.δs, nazim2(t.azimuthal_quadrature))
    total < 0 && error("rt_multi_segmentize: " * lasterror())
    nfail = Ref{Int64}(0); uid = Ref{Int64}(0); st = Ref{Int32}(0)
    ccall((:rt_multi_failed_tracks, LIB), Int32, (Ptr{Cvoid}, Ref{Int64}, Ref{Int64}, Ref{Int32}), hm, nfail, uid, st)
    if nfail[] > 0
        msg = unsafe_string(ccall((:rt_status_message, LIB), Cstring, (Int32,), st[]))
        error(replace(msg, "%d" => string(uid[])))
    end
    offs = Vector{Int64}(undef, n + 1); status = Vector{Int32}(undef, n)
    ccall((:rt_multi_fetch_offsets, LIB), Int32, (Ptr{Cvoid}, Ptr{Int64}, Ptr{Int32}), hm, offs, status)
    spx = Vector{Float64}(undef, total); spy = similar(spx); sqx = similar(spx); sqy = similar(spx); sℓ = similar(spx)
    sel = Vector{Int32}(undef, total)
    rc = ccall((:rt_multi_fetch_segments, LIB), Int32,
               (Ptr{Cvoid}, Ptr{Float64}, Ptr{Float64}, Ptr{Float64}, Ptr{Float64}, Ptr{Float64}, Ptr{Int32}),
               hm, spx, spy, sqx, sqy, sℓ, sel)
    rc != 0 && error("rt_multi_fetch_segments: " * lasterror())
    Threads.@threads for u in 1:n
        segs = tracks[u].segments
        cnt = Int(offs[u+1] - offs[u])
        resize!(segs, cnt)
        base = Int(offs[u])
        @inbounds for i in 1:cnt
            s = base + i
            segs[i] = Segment(Point2D(spx[s], spy[s]), Point2D(sqx[s], sqy[s]), sℓ[s], Float64[], sel[s])
        end
    end
    ccall((:rt_multi_fetch_volumes, LIB), Int32, (Ptr{Cvoid}, Ptr{Float64}), hm, t.volumes)
    return t
end

# rt_multi_create for the generator's mesh and tracks (flattened as in segmentize_amd!)
function multi_create(t::TrackGenerator{Float64}, devices::Vector{Int})
    tracks = t.tracks_by_uid
    mesh = t.mesh
    coords = get_node_coordinates(get_grid(mesh.model))
    x = Float64[c[1] for c in coords]; y = Float64[c[2] for c in coords]
    cell_nodes = Vector{Int32}(mesh.cell_nodes.data)
    nc_ptrs = Vector{Int32}(mesh.node_cells.ptrs); nc_data = Vector{Int32}(mesh.node_cells.data)
    bb = Float64[mesh.bb_min[1], mesh.bb_min[2], mesh.bb_max[1], mesh.bb_max[2]]
    n = length(tracks)
    px = Float64[tr.p[1] for tr in tracks]; py = Float64[tr.p[2] for tr in tracks]
    ϕ = Float64[tr.ϕ for tr in tracks]; cϕ = cos.(ϕ); sϕ = sin.(ϕ)
    A = Float64[tr.ABC[1] for tr in tracks]; B = Float64[tr.ABC[2] for tr in tracks]; C = Float64[tr.ABC[3] for tr in tracks]
    ℓ = Float64[tr.ℓ for tr in tracks]; azim = Int32[tr.azim_idx for tr in tracks]
    ids = Vector{Int32}(devices)
    hm = ccall((:rt_multi_create, LIB), Ptr{Cvoid},
               (Ptr{Int32}, Int32, Ptr{Float64}, Ptr{Float64}, Int32, Ptr{Int32}, Int32, Ptr{Int32}, Ptr{Int32}, Ptr{Float64},
                Int64, Ptr{Float64}, Ptr{Float64}, Ptr{Float64}, Ptr{Float64}, Ptr{Float64}, Ptr{Float64}, Ptr{Float64},
                Ptr{Float64}, Ptr{Float64}, Ptr{Int32}),
               ids, Int32(length(ids)), x, y, Int32(length(x)), cell_nodes, Int32(length(cell_nodes) ÷ 3), nc_ptrs, nc_data, bb,
               n, px, py, ϕ, cϕ, sϕ, A, B, C, ℓ, azim)
    hm == C_NULL && error("rt_multi_create: " * lasterror())
    return hm
end

# Opt-in replacement of the reference entry point:  RayTracingAMD.install!()
function install!()
    @eval RayTracing segmentize!(t::TrackGenerator{Float64}; k::Int=5, rtol::Real=Base.rtoldefault(Float64)) =
        $(segmentize_amd!)(t; k=k, rtol=rtol)
    return nothing
end

end # module
