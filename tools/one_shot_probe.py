"""The one-shot sequence of bench.py (rt_result_alloc, rt_tracks_create, first rt_segmentize, rt_result_fetch) several times in a row
on one box, with the box's CPU share and huge-page state (development: what the sequence's box-to-box spread is made of).
usage (GPU box): python tools/one_shot_probe.py [reps] [option=value ...]"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import bench
import raytracing_jl_amd as rt
from raytracing_jl_amd import _capi


def cat(p):
    try:
        return open(p).read().strip()
    except Exception as e:
        return "n/a (%s)" % type(e).__name__


reps = int(sys.argv[1]) if len(sys.argv) > 1 else 6
extra = dict(kv.split("=") for kv in sys.argv[2:])
print("cpu.max", cat("/sys/fs/cgroup/cpu.max"), "| nproc", os.cpu_count(), "| affinity", len(os.sched_getaffinity(0)),
      "| THP enabled", cat("/sys/kernel/mm/transparent_hugepage/enabled"), "| defrag", cat("/sys/kernel/mm/transparent_hugepage/defrag"), flush=True)
mi = {l.split(":")[0]: l.split(":")[1].strip() for l in open("/proc/meminfo")}
print({k: mi.get(k) for k in ("MemTotal", "MemFree", "MemAvailable", "AnonHugePages", "HugePages_Total")}, "| cpu.stat", cat("/sys/fs/cgroup/cpu.stat").replace("\n", " "), flush=True)
tg = bench.make_tg(rt, bench.WORKLOADS["c3"])
aq = tg.azimuthal_quadrature
dm = _capi.DeviceMesh(tg.mesh, 0)
for k, v in extra.items():
    dm.set_option(k, int(v))
def cpustat():
    d = dict(l.split() for l in cat("/sys/fs/cgroup/cpu.stat").split("\n") if len(l.split()) == 2)
    return int(d.get("usage_usec", 0)), int(d.get("nr_throttled", 0)), int(d.get("throttled_usec", 0))


def faults():
    import resource
    ru = resource.getrusage(resource.RUSAGE_SELF)
    return ru.ru_minflt, ru.ru_majflt, ru.ru_nvcsw, ru.ru_nivcsw


for r in range(reps):
    c0, f0 = cpustat(), faults()
    s = bench.one_shot_sequence(rt, _capi, dm, tg, aq)
    c1, f1 = cpustat(), faults()
    print("rep %d: alloc %.2f  h2d %.2f  segmentize %.2f  fetch %.2f  wall %.2f ms | cgroup cpu %.1f ms, throttled periods +%d (%.1f ms) | minor faults +%d, ctx switches +%d / +%d" %
          (r, s["result_alloc_ms"], s["tracks_h2d_ms"], s["segmentize_first_call_ms"], s["fetch_result_ms"], s["wall_ms"], (c1[0] - c0[0]) / 1e3, c1[1] - c0[1],
           (c1[2] - c0[2]) / 1e3, f1[0] - f0[0], f1[2] - f0[2], f1[3] - f0[3]), flush=True)
    time.sleep(float(os.environ.get("PROBE_SLEEP", "0.2")))
print("cpu.stat", cat("/sys/fs/cgroup/cpu.stat").replace("\n", " "))
