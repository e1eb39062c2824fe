"""Per-kernel averages of every counter found in rocprofv3 --pmc counter_collection CSVs under the given
directories.  usage: python tools/pmc_generic.py KERNEL_SUBSTRING DIR [DIR ...]"""
import collections, csv, glob, json, sys

pat, dirs = sys.argv[1], sys.argv[2:]
acc = collections.defaultdict(lambda: collections.defaultdict(lambda: [0, 0.0]))
for d in dirs:
    for f in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
        per_dispatch = collections.defaultdict(float)
        names = {}
        for r in csv.DictReader(open(f)):
            if pat not in r["Kernel_Name"]:
                continue
            per_dispatch[(r["Dispatch_Id"], r["Counter_Name"])] += float(r["Counter_Value"])
            names[r["Dispatch_Id"]] = r["Kernel_Name"].split("(")[0]
        for (did, c), v in per_dispatch.items():
            a = acc[names[did]][c]
            a[0] += 1
            a[1] += v
out = {k: {c: v[1] / v[0] for c, v in sorted(cs.items())} for k, cs in acc.items()}
print(json.dumps(out, indent=1))
