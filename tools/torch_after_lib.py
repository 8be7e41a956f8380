import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import raytracing_jl_amd as rt
from raytracing_jl_amd import _capi
model = rt.DiscreteModelFromFile(rt.data_path("pincell.json"))
tg = rt.TrackGenerator(model, 8, 0.02); rt.trace(tg)
mode = sys.argv[1]
if mode == "a":
    rt.segmentize(tg)
elif mode == "b":
    for i in range(30):
        dm = _capi.DeviceMesh(tg.mesh, 0)
        tg.device_mesh = dm
        rt.segmentize(tg)
elif mode == "c":
    rt.segmentize(tg)
    dm = tg.device_mesh
    dm.set_option("pool_chunks_hint", 8)
    tg2 = rt.TrackGenerator(model, 32, 5e-3); rt.trace(tg2)
    tg2.device_mesh = dm
    rt.segmentize(tg2)
elif mode == "d":
    try:
        _capi.DeviceMesh(tg.mesh, 7)
    except Exception as e:
        print("expected:", e)
    rt.segmentize(tg)
import torch
print(mode, "torch sees", torch.cuda.is_available(), torch.cuda.device_count())
t = torch.zeros(4, device="cuda"); print(t.sum().item())
