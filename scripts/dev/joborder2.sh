#!/bin/bash
# targeted job orders after the wave timeline of round 4's second half: waves 4 - 7 (one inverse-FFT call + a 256-sample job) end
# the ring-job phase, waves 0 - 3 and 10, 11 have 4 - 5 k cycles of slack
cd "$(dirname "$0")/../.."
for rep in 1 2; do
for ord in "0,1,2,3,4,5,6,7,8,9,10,11,12,13,14,15" "0,1,2,3,10,11,6,7,8,9,4,5,12,13,14,15" "0,1,2,3,10,11,8,9,6,7,4,5,12,13,14,15" "4,5,6,7,0,1,2,3,8,9,10,11,12,13,14,15" "0,1,2,3,8,9,10,11,4,5,6,7,12,13,14,15" "0,1,2,3,10,11,12,7,8,9,4,5,6,13,14,15"; do
RALIGN_JOB_ORDER=$ord python bench.py --no-cpu-baseline --no-pcie --no-parity --no-others > gpurun_out/ab_tmp.json 2>gpurun_out/ab_tmp.err || exit 1
python -c "
import json,sys
d=json.loads(open('gpurun_out/ab_tmp.json').read().strip().splitlines()[-1]); r=d['roofline']; print('%-42s %9.0f particles/s  search %.3f ms' % (sys.argv[1], d['value'], r['avg_launch_ms']))" $ord
done
done
