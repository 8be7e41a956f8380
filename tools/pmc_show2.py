"""Mean PMC counter values per kernel (all kernels): python tools/pmc_show2.py gpurun_out/<dir> [...]"""
import csv, glob, collections, sys
for d in sys.argv[1:]:
    for f in glob.glob(f"{d}/pass1/*/*counter_collection.csv"):
        acc = collections.defaultdict(list)
        for r in csv.DictReader(open(f)):
            acc[(r["Kernel_Name"].split("(")[0][-30:], r["Counter_Name"])].append(float(r["Counter_Value"]))
        for k, v in sorted(acc.items()):
            if "rocclr" in k[0]: continue
            print(d.split("/")[-1], k[0], k[1], "%.4g" % (sum(v) / len(v)), len(v))
