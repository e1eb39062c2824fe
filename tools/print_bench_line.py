"""Read bench.py output on stdin and print the headline numbers of its JSON line."""
import json, sys
for l in sys.stdin:
    if l.startswith("{"):
        j = json.loads(l)
        r = j.get("roofline", {})
        print("value %.4g %s  ms_per_step %.3f  roofline.frac %.3f  field_busy/wall %.3f" % (
            j["value"], j["unit"], j["ms_per_step"], r.get("frac", float("nan")), r.get("field_busy_over_wall", float("nan"))))
