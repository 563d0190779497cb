#!/bin/bash
# large-box contraction with 8 / 4 waves per workgroup (profiling build: RALIGN_GCCF_WAVES), transforms unchanged: how much does the
# contraction lose at one wave per SIMD -- the occupancy that would leave room for a transform workgroup on the same CU
set -e
cd "$(dirname "$0")/../.."
mkdir -p gpurun_out
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -shared -DRALIGN_PROFILE_SWITCHES -Iinclude -o gpurun_out/libralign_prof.so cryo_ralib_amd/csrc/ralign_engine.hip
for w in 8 4 6; do
  for m in 0 1; do       # RALIGN_DEBUG=1: no inverse transforms (contraction alone)
    echo "waves $w, RALIGN_DEBUG=$m"
    RALIGN_GCCF_WAVES=$w RALIGN_LIB=$PWD/gpurun_out/libralign_prof.so RALIGN_DEBUG=$m python bench.py --workload largebox --steps 1 --warmup 1 --particles 2640 --no-cpu-baseline --no-parity --no-pcie --function none 2>&1 | tail -1 |
        python -c "import sys,json; d=json.loads(sys.stdin.read()); r=d['roofline']; print('  particles/s %.0f ' % d['value'], {k: round(v['avg_launch_ms'],2) for k,v in r['kernels'].items()})"
  done
done
