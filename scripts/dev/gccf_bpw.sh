#!/bin/bash
# large box with live-offset lists: blocks per contraction workgroup and slice (profiling build: RALIGN_GCCF_BPW) -- a partially filled
# last slice costs a whole slice time
set -e
cd "$(dirname "$0")/../.."
mkdir -p gpurun_out
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -shared -DRALIGN_PROFILE_SWITCHES -Iinclude -o gpurun_out/libralign_prof.so cryo_ralib_amd/csrc/ralign_engine.hip
for b in 2 1 3; do
    echo "blocks per workgroup $b"
    RALIGN_GCCF_BPW=$b RALIGN_LIB=$PWD/gpurun_out/libralign_prof.so python bench.py --workload largebox --steps 3 --warmup 1 --no-cpu-baseline --no-parity --no-pcie 2>&1 | tail -1 |
        python -c "import sys,json; d=json.loads(sys.stdin.read()); r=d['roofline']; print('  particles/s %.0f ' % d['value'], {k: round(v['avg_launch_ms'],2) for k,v in r['kernels'].items()}, 'live', round(d['config']['live_shift_fraction'],3))"
done
