"""Extracts the known-answer values the reference's own test-suite holds for this path
(test/runtests.jl) into tests/golden/reference_kats.json.  Runs in the build container only
(/root/reference is not available on the GPU box); the JSON it writes is the committed
fixture.  Only data is extracted — expected values of `@test` lines — never source text.
"""
import json
import os
import re
import sys

REF = "/root/reference/test/runtests.jl"
OUT = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden", "reference_kats.json")


def main():
    src = open(REF).read().split("\n")
    kats = {"source": "rvignolo/RayTracing.jl test/runtests.jl (values of @test lines only)", "main": {}, "reflection": []}
    text = "\n".join(src)
    m = re.search(r"TrackGenerator\(model, (\d+), ([\d.]+)\)", text)
    kats["main"]["n_azim"], kats["main"]["delta"] = int(m.group(1)), float(m.group(2))
    kats["main"]["n_total_tracks"] = int(re.search(r"n_total_tracks == (\d+)", text).group(1))
    for name in ("n_tracks_x", "n_tracks_y", "n_tracks"):
        kats["main"][name] = json.loads(re.search(name + r" == (\[[\d, ]+\])", text).group(1))
    kats["main"]["delta_s"] = float(re.search(r"δs, ([\d.]+)\)", text).group(1))
    kats["main"]["phis"] = json.loads(re.search(r"ϕs ≈ (\[[\d., ]+\])", text).group(1))
    # reflection cases: each `tg = TrackGenerator(model, n, δ; bcs=bcs)` followed by "Track k" blocks
    case = None
    bcs = None
    cur = None
    for ln in src:
        mb = re.search(r"bcs = BoundaryConditions\((.*)\)", ln)
        if mb:
            bcs = dict(kv.split("=") for kv in mb.group(1).replace(" ", "").split(","))
        mt = re.search(r"tg = TrackGenerator\(model, (\d+), ([\d.]+); bcs=bcs\)", ln)
        if mt:
            case = {"n_azim": int(mt.group(1)), "delta": float(mt.group(2)), "bcs": bcs, "tracks": []}
            kats["reflection"].append(case)
        mk = re.search(r'@testset "Track (\d+)"', ln)
        if mk and case is not None:
            cur = {"uid": int(mk.group(1))}
            case["tracks"].append(cur)
        if cur is not None:
            for key, pat in (("bc_fwd", r"bc_fwd\(track\) == (\w+)"), ("bc_bwd", r"bc_bwd\(track\) == (\w+)"),
                             ("next_fwd_uid", r"next_track_fwd.uid == (\d+)"), ("next_bwd_uid", r"next_track_bwd.uid == (\d+)"),
                             ("dir_fwd", r"DirNextTrackFwd == RayTracing\.(\w+)"), ("dir_bwd", r"DirNextTrackBwd == RayTracing\.(\w+)")):
                mm = re.search(r"@test .*" + pat, ln)
                if mm:
                    v = mm.group(1)
                    cur[key] = int(v) if v.isdigit() else v
    os.makedirs(os.path.dirname(OUT), exist_ok=True)
    with open(OUT, "w") as fh:
        json.dump(kats, fh, indent=1, ensure_ascii=False)
    n = sum(len(c["tracks"]) for c in kats["reflection"])
    print("wrote", OUT, "with", n, "reflection tracks")


if __name__ == "__main__":
    sys.exit(main())
