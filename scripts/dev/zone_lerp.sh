#!/bin/bash
# polar_zone_kernel: Util::bilinear's own operation order (shipped) against the two-lerp form of the particle-resident kernels
cd "$(dirname "$0")/../.."
mkdir -p gpurun_out
for opt in "" "-DRA_ZONE_LERP"; do
    /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -shared $opt -Iinclude -o gpurun_out/libralign_v.so cryo_ralib_amd/csrc/ralign_engine.hip || exit 1
    echo "variant '$opt'"
    RALIGN_LIB=$PWD/gpurun_out/libralign_v.so python bench.py --workload largebox --steps 2 --warmup 1 --particles 2640 --no-cpu-baseline --no-pcie --function none 2>/dev/null | tail -1 |
        python -c "import sys,json; d=json.loads(sys.stdin.read()); r=d['roofline']; print('  particles/s %.0f ' % d['value'], {k: round(v['avg_launch_ms'],2) for k,v in r['kernels'].items()}, 'flips', d['parity']['sigma_1']['tie_flips'], d['parity']['sigma_0.25']['tie_flips'], 'peak', d['parity']['sigma_1']['max_rel_peak'])"
done
