"""Mean PMC counter values per kernel from tools/pmc_cmd.sh outputs: python tools/pmc_show.py gpurun_out/<dir> [...]"""
import csv, glob, collections, sys
for d in sys.argv[1:]:
    for f in glob.glob(f"{d}/pass1/*/*counter_collection.csv"):
        acc = collections.defaultdict(list)
        for r in csv.DictReader(open(f)):
            if "rt::" in r["Kernel_Name"]:
                acc[(r["Kernel_Name"].split("(")[0][-28:], r["Counter_Name"])].append(float(r["Counter_Value"]))
        for k, v in sorted(acc.items()):
            print(d, k[0], k[1], "%.4g" % (sum(v) / len(v)), len(v))
