"""Per (kernel, grid) durations from a rocprofv3 --kernel-trace csv: tools/rocprof_by_grid.py <dir> [n]
Tells the four projection shapes of a prefill apart (same kernel name, different grids)."""
import csv, glob, sys, collections
d = sys.argv[1]; n = int(sys.argv[2]) if len(sys.argv) > 2 else 30
for f in glob.glob(d + "/**/*kernel_trace.csv", recursive=True):
    agg = collections.defaultdict(list)
    for r in csv.DictReader(open(f)):
        g = "x".join(str(int(r[k]) // max(1, int(r[w]))) for k, w in (("Grid_Size_X", "Workgroup_Size_X"), ("Grid_Size_Y", "Workgroup_Size_Y"), ("Grid_Size_Z", "Workgroup_Size_Z")))
        agg[(r["Kernel_Name"][:70], g)].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
    rows = sorted(agg.items(), key=lambda kv: -sum(kv[1]))
    tot = sum(sum(v) for _, v in rows)
    print(f, "total ms %.3f" % (tot / 1e3))
    for (name, g), v in rows[:n]:
        v.sort()
        print("%5.1f%%  calls %5d  median %7.2f us  min %7.2f  wgs %-12s %s" % (sum(v) / tot * 100, len(v), v[len(v) // 2], v[0], g, name))
