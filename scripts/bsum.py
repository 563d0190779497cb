"""print one line per bench log: particles/s, ms/step, dominant kernel launch average, roofline fractions"""
import json, sys
for f in sys.argv[1:]:
    for l in open(f):
        if l.startswith("{"):
            d = json.loads(l); r = d["roofline"]
            print(f.split("/")[-1], round(d["value"]), "p/s", round(d["ms_per_step"], 2), "ms/step ", r["kernel"], round(r["avg_launch_ms"], 2), "ms x", r["launches"],
                  " frac %.3f whole %.3f hot share %.3f live %.2f" % (r["frac"], r["whole_path"]["frac"], r["hot_kernels_share_of_step"], d["config"].get("live_shift_fraction", 1.0)))
            for k, o in d.get("other_workloads", {}).items():
                ro = o["roofline"]
                print("   ", k, round(o["value"]), "p/s", round(o["ms_per_step"], 2), "ms/step  frac %.3f whole %.3f hot %.3f" % (ro["frac"], ro["whole_path"]["frac"], ro["hot_kernels_share_of_step"]),
                      "flips", [o.get("parity", {}).get(s, {}).get("tie_flips") for s in ("sigma_0.25", "sigma_1")])
