"""Average rocprofv3 --pmc counter values per kernel (and, from the kernel trace of the same run, the average dispatch
duration): pmc_summary.py <dir> [kernel-substring]"""
import csv, glob, sys, collections
d, sub = sys.argv[1], (sys.argv[2] if len(sys.argv) > 2 else "")
acc = collections.defaultdict(lambda: collections.defaultdict(list))
dur = collections.defaultdict(list)
for f in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if sub in r["Kernel_Name"]:
            acc[r["Kernel_Name"][:60]][r["Counter_Name"]].append(float(r["Counter_Value"]))
for f in glob.glob(d + "/**/*kernel_trace.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if sub in r["Kernel_Name"]:
            dur[r["Kernel_Name"][:60]].append((float(r["End_Timestamp"]) - float(r["Start_Timestamp"])) / 1e3)
for k, cs in acc.items():
    print(k)
    if dur.get(k):
        print(f"   {'duration_us (kernel trace, this pass)':32s} {sum(dur[k]) / len(dur[k]):16.1f}  (n={len(dur[k])})")
    for c, v in sorted(cs.items()):
        print(f"   {c:32s} {sum(v) / len(v):16.1f}  (n={len(v)})")
