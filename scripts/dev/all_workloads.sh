#!/bin/bash
# one bench line per workload (no CPU baseline, no PCIe leg): value, ms per step, roofline fraction of the dominant kernel
for w in ${@:-mref reffree mref50 nb00 box128 box100}; do
  python bench.py --workload $w --no-cpu-baseline --no-pcie --no-others 2>/dev/null | python -c "
import sys,json; d=json.loads(sys.stdin.readlines()[-1]); r=d['roofline']; p=d.get('parity') or {}
print('$w', round(d['value']), round(d['ms_per_step'],2), r['kernel'], round(r['frac'],4), 'flips', [p[k]['tie_flips'] for k in p if k.startswith('sigma')])"
done
