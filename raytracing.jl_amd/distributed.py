"""Multi-GPU sharding of ``segmentize!``: one process per GPU, tracks partitioned by uid.

Tracks are independent in the reference (``src/trackgenerator.jl:362-364``: each track only
writes its own ``segments``), so the path shards without any data-path exchange: rank ``r``
marches a contiguous uid range balanced by Σℓ (segments ∝ ℓ).  The only reduction the
reference performs across tracks is ``fill_volumes`` (``src/trackgenerator.jl:378-386``),
which becomes one all-reduce(sum) of ``n_cells`` doubles.  Reassembling the global segment
list on every rank (``allgather_segments``) is optional: a consumer that stays sharded (a
transport sweep over the rank's own tracks) does not need it.

``torch.distributed`` is plumbing here (RCCL when the backend is ``nccl``; ``gloo`` in the
CPU tests).  The compute function is injected so the CPU tests can exercise the partition /
gather logic without a GPU.
"""
from __future__ import annotations

from typing import Callable, Dict, List, Tuple

import numpy as np

__all__ = ["shard_ranges", "shard_arrays", "segmentize_shard", "allreduce_volumes",
           "allgather_segments", "SegmentGather", "DevArray", "TRACK_FIELDS", "PipelinedVolumesAllReduce",
           "SweepExchangePlan", "ShardedSweep"]

TRACK_FIELDS = ("px", "py", "phi", "cos_phi", "sin_phi", "A", "B", "C", "ell", "azim_idx")


def shard_ranges(ell: np.ndarray, world_size: int) -> List[Tuple[int, int]]:
    """Contiguous 0-based uid ranges ``[lo, hi)`` per rank with ≈ equal Σℓ."""
    n = len(ell)
    if world_size <= 0:
        raise ValueError("world_size must be positive")
    cum = np.concatenate(([0.0], np.cumsum(ell, dtype=np.float64)))
    targets = cum[-1] * np.arange(1, world_size) / world_size
    cuts = np.searchsorted(cum, targets, side="left")
    cuts = np.clip(cuts, 0, n)
    bounds = np.concatenate(([0], cuts, [n])).astype(np.int64)
    bounds = np.maximum.accumulate(bounds)
    return [(int(bounds[r]), int(bounds[r + 1])) for r in range(world_size)]


def shard_arrays(tg, lo: int, hi: int) -> Dict[str, np.ndarray]:
    """The per-track inputs of uid range ``[lo, hi)`` (0-based) as contiguous arrays."""
    return {k: np.ascontiguousarray(getattr(tg, k)[lo:hi]) for k in TRACK_FIELDS}


class DevArray:
    """Zero-copy view of library-owned device memory for ``torch.as_tensor`` (exposes
    ``__cuda_array_interface__``; the owner keeps the memory alive)."""

    def __init__(self, ptr: int, n: int, typestr: str, owner):
        self.__cuda_array_interface__ = {
            "shape": (int(n),), "typestr": typestr, "data": (int(ptr), False), "version": 2, "strides": None,
        }
        self._owner = owner


def segmentize_shard(tg, rank: int, world_size: int, *, device: int = 0, k: int = 5, rtol: float | None = None,
                     dmesh=None):
    """Upload this rank's uid range and return ``(DeviceTracks, (lo, hi))`` ready for
    ``DeviceTracks.segmentize`` (HIP path; raises without a GPU)."""
    from . import _capi

    lo, hi = shard_ranges(tg.ell, world_size)[rank]
    a = shard_arrays(tg, lo, hi)
    dm = dmesh if dmesh is not None else _capi.DeviceMesh(tg.mesh, device)
    dt = _capi.DeviceTracks(dm, a["px"], a["py"], a["phi"], a["cos_phi"], a["sin_phi"], a["A"], a["B"], a["C"],
                            a["ell"], a["azim_idx"])
    return dt, (lo, hi)


def allreduce_volumes(volumes, group=None):
    """Sum the per-rank partial ``volumes`` (each already divided by ``n_azim_2``) in place."""
    import torch.distributed as dist

    if dist.is_available() and dist.is_initialized() and dist.get_world_size(group) > 1:
        dist.all_reduce(volumes, op=dist.ReduceOp.SUM, group=group)
    return volumes


class SegmentGather:
    """All-gather-v of the ranks' segment arrays straight into the final buffers.

    Ranks own contiguous uid ranges in rank order, so the global list is the concatenation of the shards in rank
    order: every rank sends each of its six arrays to every peer and receives each peer's into the slice of the
    final buffer where that shard belongs — one batched group of point-to-point operations
    (``batch_isend_irecv``: RCCL send/recv pairs, all issued inside one group), no padding, no staging, no
    concatenation.  xGMI is point to point, so the 7 transfers a rank sends travel over 7 different links at
    once, which a ring all-gather of 8 unequal blocks would not use.  The final buffers are allocated once and
    reused while they are large enough.  Works over ``gloo`` as well (CPU tests)."""

    NAMES = ("px", "py", "qx", "qy", "ell", "element")

    def __init__(self, group=None):
        self.group = group
        self.buf = {}
        self.cap = 0

    def __call__(self, local: dict) -> dict:
        import torch
        import torch.distributed as dist

        on = dist.is_available() and dist.is_initialized()
        world = dist.get_world_size(self.group) if on else 1
        rank = dist.get_rank(self.group) if on else 0
        dev = local["ell"].device
        n_seg, n_trk = int(local["ell"].numel()), int(local["counts"].numel())
        if world == 1:
            counts = local["counts"]
            out = {k: local[k] for k in self.NAMES}
        else:
            sizes = torch.tensor([n_seg, n_trk], dtype=torch.int64, device=dev)
            all_sizes = [torch.zeros_like(sizes) for _ in range(world)]
            dist.all_gather(all_sizes, sizes, group=self.group)
            all_sizes = torch.stack(all_sizes).cpu().numpy()
            seg_begin = np.concatenate(([0], np.cumsum(all_sizes[:, 0])))
            trk_begin = np.concatenate(([0], np.cumsum(all_sizes[:, 1])))
            total, n_tracks = int(seg_begin[-1]), int(trk_begin[-1])
            if total > self.cap or self.buf.get("dev") != dev:
                cap = total + total // 16 + 64
                self.buf = {k: torch.empty(cap, dtype=torch.int32 if k == "element" else torch.float64, device=dev) for k in self.NAMES}
                self.buf["dev"] = dev
                self.cap = cap
            out = {k: self.buf[k][:total] for k in self.NAMES}
            counts = torch.empty(n_tracks, dtype=torch.int64, device=dev)
            ops = []
            for k in self.NAMES + ("counts",):
                dst = counts if k == "counts" else out[k]
                begin = trk_begin if k == "counts" else seg_begin
                dst[int(begin[rank]): int(begin[rank + 1])].copy_(local[k])  # the rank's own shard: a local copy
                for p in range(world):
                    if p == rank:
                        continue
                    if local[k].numel():
                        ops.append(dist.P2POp(dist.isend, local[k], self._peer(p), self.group))
                    if begin[p + 1] > begin[p]:
                        ops.append(dist.P2POp(dist.irecv, dst[int(begin[p]): int(begin[p + 1])], self._peer(p), self.group))
            if ops:
                for w in dist.batch_isend_irecv(ops):
                    w.wait()
        offsets = torch.zeros(counts.numel() + 1, dtype=torch.int64, device=dev)
        torch.cumsum(counts, 0, out=offsets[1:])
        out = dict(out)
        out["offsets"] = offsets
        return out

    def _peer(self, p):
        import torch.distributed as dist

        return p if self.group is None else dist.get_global_rank(self.group, p)


def allgather_segments(local: dict, group=None, gather: "SegmentGather | None" = None) -> dict:
    """Reassemble the global segment list on every rank.

    ``local``: torch tensors of this rank's shard — ``counts`` (int64, per local track),
    ``px, py, qx, qy, ell`` (float64) and ``element`` (int32), all on one device.  Returns ``offsets``
    (int64, n_tracks+1) and the six arrays (views of ``gather``'s buffers, see ``SegmentGather``)."""
    return (gather or SegmentGather(group))(local)


class PipelinedVolumesAllReduce:
    """All-reduce(sum) of every step's per-rank ``volumes``, one step late.

    The library alternates between two ``volumes`` buffers from call to call (``rt_device_pointers``),
    so the buffer of step i stays untouched while step i+1 runs.  ``hook`` — registered as the
    mesh's enqueue hook (``rt_mesh_set_enqueue_hook``: called once a call's kernels are queued,
    before the call waits for them) — issues the all-reduce of step i's buffer during step i+1, on
    a side stream when the tensors live on a GPU, so that neither its launch nor its latency sits
    between two steps.  Protocol per step::

        k = p.before_call()        # buffer k is about to be zeroed: its all-reduce of two steps ago must be over
        ... segmentize (calls p.hook() from inside) ...
        p.after_call(k, volumes_tensor_of_this_call)

    and ``p.drain()`` after the last step (inside any timed region).  The reduced values are left in
    the library's buffers.  ``all_reduce`` is injectable for tests.
    """

    def __init__(self, group=None, device=None, all_reduce=None):
        import torch
        import torch.distributed as dist

        self._dist = dist
        self._group = group
        self._all_reduce = all_reduce or (lambda t: dist.all_reduce(t, op=dist.ReduceOp.SUM, group=group, async_op=True))
        self._side = None
        if device is not None and torch.device(device).type == "cuda":
            self._side = torch.cuda.Stream(device=device)
        self._torch = torch
        self._prev = None
        self._work = {}
        self.n = 0

    def hook(self):
        if self._prev is None:
            return
        k, vol = self._prev
        self._prev = None
        if self._side is not None:
            # the buffer's kernels finished with the previous (host-synchronous) call: nothing to wait for
            with self._torch.cuda.stream(self._side):
                self._work[k] = self._all_reduce(vol)
        else:
            self._work[k] = self._all_reduce(vol)

    def before_call(self) -> int:
        k = self.n & 1
        w = self._work.get(k)
        if w is not None:
            # two steps old: normally long finished, and a completion seen by the host orders everything
            # launched from now on after it — no wait has to go into the stream
            done = getattr(w, "is_completed", None)
            if done is None or not done():
                w.wait()  # NCCL: a stream-level wait; gloo: blocks until done
            self._work[k] = None
        return k

    def after_call(self, k: int, volumes):
        self._prev = (k, volumes)
        self.n += 1

    def drain(self):
        self.hook()
        for k, w in list(self._work.items()):
            if w is not None:
                w.wait()
                self._work[k] = None


class SweepExchangePlan:
    """The one real exchange of the sharded path (pure numpy; no communication): a transport sweep hands the flux a traversal
    ends with to the entry of the linked track (``next_track_fwd / next_track_bwd`` + ``dir_next_track_*``,
    ``src/track.jl:42-77``), and with ``tracks_by_uid`` cut into uid ranges most links leave the rank — reflective links pair an
    azimuthal angle with its supplementary one, i.e. the first ranks with the last.  For rank ``rank`` of ``ranges``:

    * ``local_links``: the link arrays for ``rt_sweep_set_links`` of the shard — local 1-based uids, 0 where the linked track
      lives on another rank (the library then hands nothing on locally);
    * ``send[p]`` = (indices into the shard's ``psi_out`` viewed as [2·n_local, G], 0/1 mask: 0 behind a Vacuum boundary) and
      ``recv[p]`` = indices into the shard's ``psi_in``, for every peer ``p`` with traffic — both in the order of the global
      source key (uid, direction), which every rank derives from the same global arrays, so the two sides agree without talking.
    """

    def __init__(self, next_fwd, next_bwd, dir_fwd, dir_bwd, bc_fwd, bc_bwd, ranges, rank):
        nxt = (np.asarray(next_fwd, np.int64) - 1, np.asarray(next_bwd, np.int64) - 1)
        dirs = (np.asarray(dir_fwd, np.int64), np.asarray(dir_bwd, np.int64))
        bcs = (np.asarray(bc_fwd, np.int64), np.asarray(bc_bwd, np.int64))
        n = len(nxt[0])
        bounds = np.array([r[0] for r in ranges] + [ranges[-1][1]], np.int64)
        owner = lambda u: np.clip(np.searchsorted(bounds, u, side="right") - 1, 0, len(ranges) - 1)
        lo, hi = ranges[rank]
        self.rank, self.lo, self.hi, self.n_local = rank, lo, hi, hi - lo
        own = np.arange(lo, hi)
        loc = {}
        for d, name in ((0, "fwd"), (1, "bwd")):
            v = nxt[d][own]
            here = (v >= lo) & (v < hi)
            loc["next_" + name] = np.where(here, v - lo + 1, 0).astype(np.int64)
            loc["dir_" + name] = dirs[d][own].astype(np.int8)
            loc["bc_" + name] = bcs[d][own].astype(np.int8)
        self.local_links = loc
        # every cross-rank link of the global problem, in the order of its source key
        g = np.repeat(np.arange(n, dtype=np.int64), 2)
        d = np.tile(np.array([0, 1], np.int64), n)
        v = np.where(d == 0, nxt[0][g], nxt[1][g])
        dn = np.where(d == 0, dirs[0][g], dirs[1][g])
        vac = np.where(d == 0, bcs[0][g], bcs[1][g]) == 0
        s_rank, r_rank = owner(g), owner(v)
        cross = s_rank != r_rank
        self.send, self.recv = {}, {}
        for p in range(len(ranges)):
            if p == rank:
                continue
            out = cross & (s_rank == rank) & (r_rank == p)
            if out.any():
                self.send[p] = ((d[out] * self.n_local + (g[out] - lo)).astype(np.int64), (~vac[out]).astype(np.float64))
            inc = cross & (s_rank == p) & (r_rank == rank)
            if inc.any():
                self.recv[p] = (dn[inc] * self.n_local + (v[inc] - lo)).astype(np.int64)


class ShardedSweep:
    """``rt_sweep`` on a uid shard + the exchange of the fluxes that leave it + the all-reduce of the tallies: the consumer of
    the sharded segments that never gathers them.  ``dt``: the rank's ``DeviceTracks`` after ``segmentize``;
    ``tg``: the GLOBAL traced ``TrackGenerator`` (every rank holds it: ``trace!`` stays on the host).  Point-to-point RCCL
    send/recv pairs (``batch_isend_irecv``) straight out of / into the library's ``psi_out`` / ``psi_in``; gloo in the CPU tests
    through the injectable ``tensors`` (a callable returning the three torch tensors phi [n_cells, G], psi_out, psi_in [2·n_local, G])."""

    def __init__(self, tg, dt, rank, world, ranges=None, group=None, device=None, tensors=None):
        self.ranges = ranges if ranges is not None else shard_ranges(tg.ell, world)
        self.plan = SweepExchangePlan(tg.next_fwd_uid, tg.next_bwd_uid, tg.dir_next_fwd, tg.dir_next_bwd, tg.bc_fwd, tg.bc_bwd,
                                      self.ranges, rank)
        self.dt, self.rank, self.world, self.group, self.device = dt, rank, world, group, device
        self._tensors = tensors
        self._idx = None
        if dt is not None:
            dt.sweep_set_links(self.plan.local_links)

    def _views(self, G):
        import torch

        if self._tensors is not None:
            return self._tensors()
        p = self.dt.sweep_pointers()
        n2 = 2 * self.plan.n_local
        mk = lambda ptr, rows: torch.as_tensor(DevArray(ptr, rows * G, "<f8", self.dt), device=self.device).view(rows, G)
        return mk(p["phi"], self.dt.dmesh.n_cells), mk(p["psi_out"], n2), mk(p["psi_in"], n2)

    def exchange(self, G):
        """Hand the fluxes that leave this shard to their owners and take in the ones that enter it; sum the tallies."""
        import torch
        import torch.distributed as dist

        # rt_sweep wrote phi / psi_out / psi_in on the MESH's stream (and, under the option "async", has only queued its
        # kernels): wait for it before torch's stream reads or writes these buffers
        if self.dt is not None:
            self.dt.wait()
        phi, psi_out, psi_in = self._views(G)
        if self._idx is None or self._idx[0] != psi_out.device:
            dev = psi_out.device
            self._idx = (dev, {p: (torch.as_tensor(i, device=dev), torch.as_tensor(m, device=dev)[:, None]) for p, (i, m) in self.plan.send.items()},
                         {p: torch.as_tensor(i, device=dev) for p, i in self.plan.recv.items()})
        _, send, recv = self._idx
        peer = (lambda p: p) if self.group is None else (lambda p: dist.get_global_rank(self.group, p))
        ops, bufs = [], {}
        for p, (idx, mask) in send.items():
            ops.append(dist.P2POp(dist.isend, (psi_out[idx] * mask).contiguous(), peer(p), self.group))
        for p, idx in recv.items():
            bufs[p] = torch.empty((idx.numel(), G), dtype=psi_in.dtype, device=psi_in.device)
            ops.append(dist.P2POp(dist.irecv, bufs[p], peer(p), self.group))
        if ops:
            for w in dist.batch_isend_irecv(ops):
                w.wait()
        for p, idx in recv.items():
            psi_in[idx] = bufs[p]
        if self.world > 1:
            dist.all_reduce(phi, op=dist.ReduceOp.SUM, group=self.group)
        # ... and the next rt_sweep (mesh stream) must see the scattered fluxes: torch's stream is drained here
        if psi_in.is_cuda:
            torch.cuda.current_stream(psi_in.device).synchronize()
        return phi, psi_out, psi_in

    def sweep(self, G, sigma_t=None, source=None, track_weight=None, psi_in=None, input="auto"):
        """One sweep of the global problem: this rank's traversals, then the exchange.  ``track_weight`` / ``psi_in``: this rank's
        slices ([n_local], [2, n_local, G]).  Returns the library's report + the (device) tensors phi (summed over the ranks),
        psi_out, psi_in (the boundary flux of the next sweep, complete)."""
        r = self.dt.sweep(G, sigma_t, source, track_weight, psi_in, input=input, fetch=False)
        r["phi"], r["psi_out"], r["psi_next"] = self.exchange(G)
        return r
