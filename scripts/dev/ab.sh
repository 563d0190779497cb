#!/bin/bash
# same-box A/B of engine builds: scripts/dev/ab.sh ab_libs/a.so ab_libs/b.so ...  (each measured twice, interleaved)
cd "$(dirname "$0")/../.."
for rep in 1 2; do
  for lib in "$@"; do
    RALIGN_LIB=$PWD/$lib python bench.py --no-cpu-baseline --no-pcie --no-parity > gpurun_out/ab_tmp.json 2>gpurun_out/ab_tmp.err || exit 1
    python - "$lib" <<'PY'
import json, sys
d = json.loads(open("gpurun_out/ab_tmp.json").read().strip().splitlines()[-1])
print("%-24s %10.0f particles/s  search launch %.3f ms" % (sys.argv[1], d["value"], d["roofline"]["avg_launch_ms"]))
PY
  done
done
