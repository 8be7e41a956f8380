"""Compare a dump made by oracle/reference_dump.jl (real RayTracing.jl) with the oracle, fed
the dumped tracks (so libm differences in trace! do not enter).  Prints what matches."""
import argparse
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("dir")
    ap.add_argument("--mesh", default="pincell.json")
    a = ap.parse_args()
    import raytracing_jl_amd as rt
    from oracle import oracle as orc

    tr = np.loadtxt(os.path.join(a.dir, "tracks.csv"), delimiter=",", ndmin=2)
    sg = np.loadtxt(os.path.join(a.dir, "segments.csv"), delimiter=",", ndmin=2)
    model = rt.DiscreteModelFromFile(rt.data_path(a.mesh))
    om = orc.OracleMesh.from_mesh(rt.Mesh(model))
    r = om.segmentize(tr[:, 2], tr[:, 3], tr[:, 6], tr[:, 8], tr[:, 9], tr[:, 10], tr[:, 7])
    counts = np.bincount(sg[:, 0].astype(int) - 1, minlength=len(tr))
    print("tracks:", len(tr), "reference segments:", len(sg), "oracle segments:", r["total"])
    print("per-track counts equal:", np.array_equal(counts, np.diff(r["offsets"])))
    if len(sg) == r["total"]:
        print("element ids equal:", np.array_equal(sg[:, 2].astype(np.int32), r["element"]))
        for j, k in enumerate(("px", "py", "qx", "qy", "ell")):
            d = np.abs(sg[:, 3 + j] - r[k])
            print(k, "max abs diff", d.max(), "bitwise equal:", bool((d == 0).all()))


if __name__ == "__main__":
    main()
