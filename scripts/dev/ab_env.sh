#!/bin/bash
# same-box comparison of environment settings: scripts/dev/ab_env.sh "A=1 B=2" "A=3" ... (each measured twice, interleaved)
cd "$(dirname "$0")/../.."
for rep in 1 2; do
  for setting in "$@"; do
    env $setting python bench.py --no-cpu-baseline --no-pcie --no-parity > gpurun_out/ab_tmp.json 2>gpurun_out/ab_tmp.err || exit 1
    python - "$setting" <<'PY'
import json, sys
d = json.loads(open("gpurun_out/ab_tmp.json").read().strip().splitlines()[-1])
print("%-40s %10.0f particles/s  search launch %.3f ms" % (sys.argv[1], d["value"], d["roofline"]["avg_launch_ms"]))
PY
  done
done
