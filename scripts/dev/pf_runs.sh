#!/bin/bash
# search_duo_kernel: operand buffers of the contraction (RALIGN_DUO_PF) x slice width (RALIGN_DUO_NQT) on the two maxrin-512 workloads
for w in nb00 box128; do
for cfg in "2 14" "1 14" "2 16" "1 16"; do
  set -- $cfg
  echo "== $w PF=$1 NQT=$2"
  RALIGN_DUO_PF=$1 RALIGN_DUO_NQT=$2 python bench.py --workload $w --no-cpu-baseline --no-pcie --no-others 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.readlines()[-1]); print(d['value'], d['ms_per_step'], d['roofline']['frac'])"
done; done
