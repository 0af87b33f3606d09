"""Average of one rocprofv3 --pmc counter per kernel name: tools/pmc_by_kernel.py <dir> <COUNTER> [n]"""
import collections, csv, glob, sys
d, counter = sys.argv[1], sys.argv[2]; n = int(sys.argv[3]) if len(sys.argv) > 3 else 12
acc = collections.defaultdict(list)
for f in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if r["Counter_Name"] == counter:
            acc[(r["Kernel_Name"][:60], r.get("Grid_Size", ""))].append(float(r["Counter_Value"]))
for (name, grid), v in sorted(acc.items(), key=lambda kv: -sum(kv[1]))[:n]:
    print("%-62s grid %-8s calls %5d  avg %12.1f" % (name, grid, len(v), sum(v) / len(v)))
