"""One-line summary of a bench.py JSON line (gpu_session.sh `one`)."""
import json
import sys

for path in sys.argv[1:]:
    l = json.loads(open(path).read().strip().splitlines()[-1])
    r = l["roofline"]
    print("%-22s value %8.0f  ms/step %.4f  launch %.4f  period %s  frac %.3f  solo %.4f  path %s" % (
        path.split("/")[-1], l["value"], l["ms_per_step"], r.get("launch_ms", 0.0),
        ("%.4f" % r["launch_period_ms"]) if r.get("launch_period_ms") else "-", r["frac"],
        r.get("standalone_launch_ms", 0.0), l.get("step_path")))
