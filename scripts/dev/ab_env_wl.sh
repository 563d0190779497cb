#!/bin/bash
# same-box comparison of environment settings on another workload: scripts/dev/ab_env_wl.sh <workload> "A=1 B=2" "A=3" ...
cd "$(dirname "$0")/../.."
wl=$1; shift
for rep in 1 2; do
  for setting in "$@"; do
    env $setting python bench.py --workload $wl --no-cpu-baseline --no-pcie --no-parity --no-others > gpurun_out/ab_tmp.json 2>gpurun_out/ab_tmp.err || exit 1
    python - "$setting" <<'PY'
import json, sys
d = json.loads(open("gpurun_out/ab_tmp.json").read().strip().splitlines()[-1])
k = d["roofline"]["kernels"]
print("%-32s %10.1f particles/s  %.2f ms/step  " % (sys.argv[1], d["value"], d["ms_per_step"]) + "  ".join("%s %.2f ms" % (n, v["avg_launch_ms"]) for n, v in k.items()))
PY
  done
done
