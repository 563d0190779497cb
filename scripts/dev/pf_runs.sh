#!/bin/bash
# search_duo_kernel: slice width (RALIGN_DUO_NQT = 14 | 16) on the two maxrin-512 workloads.  The first version of this script also
# crossed it with the number of operand buffers of the contraction (RALIGN_DUO_PF = 1 | 2, commit 51c8a0f's parent tree): two buffers
# were 1 - 4 % slower on both workloads and the variant was removed; the numbers are in DESIGN.md section 4.1c.
for w in nb00 box128; do
for q in 14 16; do
  echo "== $w NQT=$q"
  RALIGN_DUO_NQT=$q python bench.py --workload $w --no-cpu-baseline --no-pcie --no-others 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.readlines()[-1]); print(d['value'], d['ms_per_step'], d['roofline']['frac'])"
done; done
