"""Cross-check of bench.py's roofline timing from a rocprofv3 --kernel-trace CSV of the same command: per field-kernel
dispatch the mean duration (what --stats lists) and the time during which at least one field kernel was executing
divided by the number of dispatches (what `roofline.avg_launch_ms` is).  usage: field_busy_from_trace.py <dir|csv> [skip]
skip = leading dispatches to leave out (the warm-up step's: calls in flight x iterations, default 0)."""
import csv, glob, os, sys
path = sys.argv[1]
files = glob.glob(os.path.join(path, "**", "*kernel_trace.csv"), recursive=True) if os.path.isdir(path) else [path]
rows = [r for r in csv.DictReader(open(files[0])) if "field_kernel" in r["Kernel_Name"] or "field_half_kernel" in r["Kernel_Name"]]
iv = sorted((int(r["Start_Timestamp"]), int(r["End_Timestamp"])) for r in rows)
iv = iv[int(sys.argv[2]) if len(sys.argv) > 2 else 0:]
n_all = len(iv)
iv = [(s, e) for s, e in iv if e - s >= 10000]      # launches without work (enqueued ahead of a finished frame) are not launches of the workload
busy, (cs, ce) = 0, iv[0]
for s, e in iv[1:]:
    if s > ce:
        busy += ce - cs; cs, ce = s, e
    else:
        ce = max(ce, e)
busy += ce - cs
print(f"{n_all} dispatches, {len(iv)} with work: mean duration {sum(e - s for s, e in iv) / len(iv) / 1e3:.1f} us, "
      f"field-busy time per dispatch {busy / len(iv) / 1e3:.1f} us, busy / span {busy / (iv[-1][1] - iv[0][0]):.3f}")
