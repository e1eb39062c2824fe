"""Timeline of the LAST frame in a rocprofv3 --kernel-trace CSV: every dispatch with start offset, duration and the gap
to the previous dispatch's end (one frame = from frame_prep_kernel to frame_finalize_kernel).
usage: python tools/frame_timeline.py <dir with *_kernel_trace.csv> [frame index from the end, default 1]"""
import csv, glob, os, sys
path = sys.argv[1]
files = glob.glob(os.path.join(path, "**", "*kernel_trace.csv"), recursive=True) if os.path.isdir(path) else [path]
rows = list(csv.DictReader(open(files[0])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
short = lambda n: n.split("(")[0].replace("void ", "").replace("ced::", "")[:40]
starts = [i for i, r in enumerate(rows) if "frame_prep" in r["Kernel_Name"]]
k = int(sys.argv[2]) if len(sys.argv) > 2 else 1
a = starts[-k]
b = next(i for i in range(a, len(rows)) if "frame_finalize_kernel" in rows[i]["Kernel_Name"])
t0 = int(rows[a]["Start_Timestamp"]); prev_end = t0
tot = {}
for r in rows[a:b + 1]:
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    nm = short(r["Kernel_Name"])
    print(f"{(s - t0) / 1e3:9.1f} us  +{(e - s) / 1e3:8.1f} us  gap {(s - prev_end) / 1e3:7.1f}  {nm}  grid {r.get('Grid_Size', '?')}")
    tot[nm] = tot.get(nm, 0.0) + (e - s) / 1e3
    prev_end = max(prev_end, e)
print(f"frame: {(prev_end - t0) / 1e3:.1f} us")
for nm, v in sorted(tot.items(), key=lambda x: -x[1]):
    print(f"  {v:9.1f} us  {nm}")
