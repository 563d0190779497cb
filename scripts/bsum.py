"""print one line per bench log: particles/s, ms/step, kernel launch averages"""
import json, sys
for f in sys.argv[1:]:
    for l in open(f):
        if l.startswith("{"):
            d = json.loads(l); r = d["roofline"]
            print(f.split("/")[-1], round(d["value"]), "p/s", round(d["ms_per_step"], 1), "ms/step  ccf", round(r["avg_launch_ms"], 2),
                  "polar", round(r["polar_fft_kernel"]["avg_launch_ms"], 2))
