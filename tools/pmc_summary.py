"""Summarises rocprofv3 --pmc counter_collection CSVs (one pass per counter, as the gfx950 guide
prescribes) into profiles/<name>.json: per kernel, the average FETCH_SIZE / WRITE_SIZE per launch.

usage: python tools/pmc_summary.py OUT.json FETCH_DIR WRITE_DIR
FETCH_SIZE / WRITE_SIZE are reported by rocprofv3 in KiB.  On gfx950 FETCH_SIZE under-counts wide
coalesced streams by 2x (MI355X_MICROARCH.md, HBM section); the field kernel's reads are 8-byte
gathers, for which the counter is uncalibrated, so the raw value is recorded and flagged."""
import collections, csv, glob, json, sys

def load(d, counter):
    out = collections.defaultdict(lambda: [0, 0.0])
    for f in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            if r["Counter_Name"] != counter:
                continue
            k = r["Kernel_Name"].split("(")[0]
            out[k][0] += 1
            out[k][1] += float(r["Counter_Value"])
    return {k: {"launches": v[0], "avg_kib": v[1] / v[0]} for k, v in out.items()}

out_path, fdir, wdir = sys.argv[1:4]
fetch, write = load(fdir, "FETCH_SIZE"), load(wdir, "WRITE_SIZE")
res = {}
for k in sorted(set(fetch) | set(write)):
    if "ced::" not in k:
        continue
    res[k] = {"fetch_bytes_per_launch": fetch.get(k, {}).get("avg_kib", 0.0) * 1024,
              "write_bytes_per_launch": write.get(k, {}).get("avg_kib", 0.0) * 1024,
              "launches_profiled": fetch.get(k, {}).get("launches", 0)}
res["_note"] = ("rocprofv3 --pmc FETCH_SIZE and --pmc WRITE_SIZE in separate passes over `python bench.py --steps 5 --warmup 1 "
                "--no-cpu-baseline --also= --no-single-frame --min-seconds 0` (the launches profiled are the warm-up step and the 5 "
                "timed steps); raw counter values x 1024 (KiB -> bytes); FETCH_SIZE is uncalibrated for 8-byte gathers on "
                "gfx950 (it under-counts wide streams 2x), WRITE_SIZE is exact for streaming stores")
json.dump(res, open(out_path, "w"), indent=1)
print(json.dumps(res, indent=1))
