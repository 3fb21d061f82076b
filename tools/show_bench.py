"""Pretty-print the headline and the sub-sections of a bench.py JSON line (file argument)."""
import json, sys
d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
r = d["roofline"]
print("value", round(d["value"]), "it/s  ms_per_step", d["ms_per_step"], " lio/vio us", round(r["lio_pass_us"], 2), round(r["vio_pass_us"], 2), " frac", r["frac"])
for k in ("mode23", "frame", "config4", "config5", "cpu_frame", "cpu_baseline"):
    print(k, json.dumps(d.get(k))[:1000])
print("at_scale_vio", json.dumps(r.get("at_scale_vio"))[:700])
print("at_scale", json.dumps(r.get("at_scale"))[:500])
