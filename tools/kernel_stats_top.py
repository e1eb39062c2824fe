"""Print the top rows of a rocprofv3 `*_kernel_stats.csv` (share of kernel time, calls, average duration)."""
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
top = int(sys.argv[2]) if len(sys.argv) > 2 else 25
tot = sum(float(r["TotalDurationNs"]) for r in rows)
for r in rows[:top]:
    print("%5.1f%% %7s calls %9.1f us  %s" % (float(r["TotalDurationNs"]) / tot * 100, r["Calls"],
                                              float(r["AverageNs"]) / 1e3, r["Name"][:120]))
print("total kernel time %.1f ms" % (tot / 1e6))
