#!/bin/bash
# polar_zone_kernel: offset chunks per (particle, zone) -- workgroups = particles x zones x chunks (profiling build: RALIGN_ZONE_CHUNKS)
set -e
cd "$(dirname "$0")/../.."
mkdir -p gpurun_out
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -shared -DRALIGN_PROFILE_SWITCHES -Iinclude -o gpurun_out/libralign_prof.so cryo_ralib_amd/csrc/ralign_engine.hip
for c in ${CHUNKS:-4 2 1 8 16}; do
    echo "chunks $c"
    RALIGN_ZONE_CHUNKS=$c RALIGN_LIB=$PWD/gpurun_out/libralign_prof.so python bench.py --workload largebox --steps 2 --warmup 1 --particles 2640 --no-cpu-baseline --no-parity --no-pcie --function none 2>&1 | tail -1 |
        python -c "import sys,json; d=json.loads(sys.stdin.read()); r=d['roofline']; print('  particles/s %.0f ' % d['value'], {k: round(v['avg_launch_ms'],2) for k,v in r['kernels'].items()})"
done
