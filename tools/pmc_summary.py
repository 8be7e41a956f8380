"""Summarise rocprofv3 --pmc passes (tools/pmc.sh): per kernel, mean counter value per dispatch."""
import csv, glob, os, sys, collections, json
root = sys.argv[1]
out = collections.defaultdict(lambda: collections.defaultdict(list))
for f in sorted(glob.glob(os.path.join(root, "pass*", "*", "*_counter_collection.csv"))):
    per = collections.defaultdict(lambda: collections.defaultdict(float))
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"].split("(")[0].replace("void ", "")
        per[(k, r["Dispatch_Id"])][r["Counter_Name"]] += float(r["Counter_Value"])
    for (k, _d), cs in per.items():
        for c, v in cs.items():
            out[k][c].append(v)
res = {}
for k, cs in out.items():
    res[k] = {c: sum(v) / len(v) for c, v in cs.items()}
    res[k]["dispatches"] = max(len(v) for v in cs.values())
for k in sorted(res, key=lambda k: -res[k].get("SQ_WAVE_CYCLES", 0)):
    print(k)
    for c, v in sorted(res[k].items()):
        print("   %-28s %.6g" % (c, v))
if len(sys.argv) > 2:
    import hashlib
    lib = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "raytracing.jl_amd", "csrc", "librt_segmentize.so")
    json.dump({"lib_sha256": hashlib.sha256(open(lib, "rb").read()).hexdigest(),
               "source": "rocprofv3 --pmc passes (tools/pmc.sh, tools/prof.sh); mean per dispatch; FETCH_SIZE / WRITE_SIZE in KiB",
               "kernels": res}, open(sys.argv[2], "w"), indent=1)
