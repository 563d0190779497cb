for ord in "0,1,2,3,4,5,6,7,8,9,10,11,12,13,14,15" "15,14,13,12,11,10,9,8,7,6,5,4,3,2,1,0" "8,9,10,11,12,13,14,15,0,1,2,3,4,5,6,7" "0,1,2,3,8,9,10,11,4,5,6,7,12,13,14,15" "12,13,14,15,0,1,2,3,4,5,6,7,8,9,10,11" "0,1,2,3,12,13,14,15,4,5,6,7,8,9,10,11"; do
RALIGN_JOB_ORDER=$ord python bench.py --no-cpu-baseline --no-pcie --no-parity --no-others > gpurun_out/ab_tmp.json 2>gpurun_out/ab_tmp.err || exit 1
python -c "
import json,sys
d=json.loads(open('gpurun_out/ab_tmp.json').read().strip().splitlines()[-1]); r=d['roofline']; print(sys.argv[1], d['value'], r['avg_launch_ms'])" $ord
done
