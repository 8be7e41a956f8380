"""Records in completion order (mesh option "record_order"), same box (development; profiles/r06/exp_completion_order.log):
ms per step with the option 0 / 1 / 2, the records of every track through the per-track table against the CSR records of a call
with the option off (bit for bit), and the hash of the CSR layout made on demand against that call's.
usage: python tools/probe_completion.py [mesh nazim delta] [modes, e.g. 0,1,2] [extra option=value ...]"""
import hashlib, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import raytracing_jl_amd as rt
from raytracing_jl_amd import _capi

mesh = sys.argv[1] if len(sys.argv) > 1 else "pincell.msh"
nazim = int(sys.argv[2]) if len(sys.argv) > 2 else 128
delta = float(sys.argv[3]) if len(sys.argv) > 3 else 1e-3
modes = [int(x) for x in (sys.argv[4] if len(sys.argv) > 4 else "0,1,2").split(",")]
extra = dict(kv.split("=") for kv in sys.argv[5:])
check = os.environ.get("PROBE_NOCHECK") is None
path = rt.data_path(mesh)
model = rt.GmshDiscreteModel(path) if mesh.endswith(".msh") else rt.DiscreteModelFromFile(path)
tg = rt.TrackGenerator(model, nazim, delta); rt.trace(tg)
aq = tg.azimuthal_quadrature
KEYS = ("px", "py", "qx", "qy", "ell", "element")


def sha(off, st, recs):
    h = hashlib.sha256()
    for x in (off, st, *[recs[k] for k in KEYS]): h.update(np.ascontiguousarray(x).tobytes())
    return h.hexdigest()[:12]


ref = None
for mode in modes:
    dm = _capi.DeviceMesh(tg.mesh, 0)
    dm.set_option("record_order", mode)
    for k, v in extra.items(): dm.set_option(k, int(v))
    dt = _capi.DeviceTracks(dm, tg.px, tg.py, tg.phi, tg.cos_phi, tg.sin_phi, tg.A, tg.B, tg.C, tg.ell, tg.azim_idx)
    seg = lambda: dt.segmentize(tg.tiny_step, 5, rt.RTOL_DEFAULT, aq.delta_s, aq.n_azim_2)
    for _ in range(4): total = seg()
    best = 1e9
    for _ in range(5):
        t0 = time.perf_counter()
        for _ in range(40): seg()
        best = min(best, (time.perf_counter() - t0) / 40 * 1e3)
    s = dt.stats()
    line = f"{mesh} {nazim} {delta} record_order={mode}: {total} segments, {best:.4f} ms/step, completion {s['completion_order']} kernel {s['record_kernel']}"
    if check:
        order = dt.record_order()
        beg, cnt, st = dt.fetch_table()
        recs = dt.fetch_records()
        vol = dt.fetch_volumes()
        if ref is None and order == 0:
            off = np.concatenate([beg, [total]])
            ref = dict(off=off, st=st.copy(), recs=recs, vol=vol, sha=sha(off, st, recs))
            line += f" | CSR sha {ref['sha']}"
        elif ref is not None:
            # every track's records through the table against the CSR call's
            ok = np.array_equal(cnt, np.diff(ref["off"])) and np.array_equal(st, ref["st"]) and int(cnt.sum()) == total
            idx = np.repeat(beg - ref["off"][:-1], cnt) + np.arange(total)  # record r of the CSR layout lies at idx[r]
            bad = {k: int((recs[k][idx].view(np.int64 if k != "element" else np.int32) != ref["recs"][k].view(np.int64 if k != "element" else np.int32)).sum()) for k in KEYS}
            spans = np.sort(beg[cnt > 0]); ends = np.sort((beg + cnt)[cnt > 0])
            dense = bool(spans[0] == 0 and ends[-1] == total and np.array_equal(spans[1:], ends[:-1]))
            line += f" | order {order}: table ok {ok}, dense {dense}, mismatching records {bad}, volumes max rel {float(np.max(np.abs(vol - ref['vol']) / np.maximum(ref['vol'], 1e-300))):.2e}"
            # the CSR layout on demand
            off2, st2 = dt.fetch_offsets(); recs2 = dt.fetch_segments()
            line += f" | CSR on demand sha {sha(off2, st2, recs2)} ({'equal' if sha(off2, st2, recs2) == ref['sha'] else 'DIFFERENT'}), order now {dt.record_order()}"
            total2 = seg()
            line += f" | next call order {dt.record_order()} total {total2}"
    print(line, flush=True)
    dt.close(); dm.close()
