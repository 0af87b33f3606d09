"""Print the kernel_stats csv of a rocprofv3 --stats run: tools/rocprof_top.py <dir> [n]"""
import csv, glob, sys
d = sys.argv[1]; n = int(sys.argv[2]) if len(sys.argv) > 2 else 25
for f in glob.glob(d + "/**/*kernel_stats.csv", recursive=True):
    rows = list(csv.DictReader(open(f)))
    rows.sort(key=lambda r: -float(r["TotalDurationNs"]))
    tot = sum(float(r["TotalDurationNs"]) for r in rows)
    print(f, "total ms", tot / 1e6)
    for r in rows[:n]:
        print(f'{float(r["TotalDurationNs"]) / tot * 100:5.1f}%  calls {int(r["Calls"]):6d}  avg {float(r["AverageNs"]) / 1e3:8.2f} us  {r["Name"][:110]}')
