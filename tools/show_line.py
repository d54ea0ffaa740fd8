"""the figures of a bench.py JSON line that are quoted in DESIGN.md, on a few lines"""
import json, sys
d = json.load(open(sys.argv[1]))
print("ms_per_step", round(d["ms_per_step"], 3), "best", d.get("best_preconditioner"), "value %.3e" % d["value"])
if d.get("config5"): print("config5", d["config5"]["steps"], "steps", round(d["config5"]["seconds"], 4), "s")
r = d["roofline"]; print("roofline", r["bound"], "achieved %.1f %s frac %.3f avg_launch_us %.1f" % (r["achieved"], r["unit"], r["frac"], r["avg_launch_us"]), "traffic", r.get("traffic"))
o = d.get("roofline_operator")
if o: print("operator avg_launch_us %.1f frac %.3f index-free %.3f traffic" % (o["avg_launch_us"], o["frac"], o["frac_without_index_bytes"]), o.get("traffic"))
print("work_per_step", d["work_per_step"])
if d.get("cpu_baseline"): print("cpu_baseline", {k: d["cpu_baseline"][k] for k in ("value", "unit", "cores", "kind")})
