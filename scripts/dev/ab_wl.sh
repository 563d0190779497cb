#!/bin/bash
# same-box A/B of engine builds on another workload: scripts/dev/ab_wl.sh <workload> ab_libs/a.so ab_libs/b.so ...
cd "$(dirname "$0")/../.."
wl=$1; shift
for rep in 1 2; do
  for lib in "$@"; do
    RALIGN_LIB=$PWD/$lib python bench.py --workload $wl --no-cpu-baseline --no-pcie --no-parity > gpurun_out/ab_tmp.json 2>gpurun_out/ab_tmp.err || exit 1
    python - "$lib" <<'PY'
import json, sys
d = json.loads(open("gpurun_out/ab_tmp.json").read().strip().splitlines()[-1])
k = d["roofline"]["kernels"]
print("%-20s %10.1f particles/s  " % (sys.argv[1], d["value"]) + "  ".join("%s %.2f ms" % (n, v["avg_launch_ms"]) for n, v in k.items()))
PY
  done
done
