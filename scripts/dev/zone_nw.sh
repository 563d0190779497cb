#!/bin/bash
# polar_zone_kernel variants, same box: waves per workgroup (register budget 256 / 170 / 128 VGPRs; the zones shrink with the
# wave buffers).  (The half-quad variant of profiles/r06_zone_variants_experiment.txt -- 8-byte panel pieces -- was removed again.)
cd "$(dirname "$0")/../.."
mkdir -p gpurun_out
for v in ${VARIANTS:-"8 4" "12 4" "16 4"}; do
    set -- $v
    /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -shared -DRA_ZONE_NW=$1 -Iinclude -o gpurun_out/libralign_v.so cryo_ralib_amd/csrc/ralign_engine.hip || exit 1
    echo "NW=$1"
    RALIGN_INFO=1 RALIGN_LIB=$PWD/gpurun_out/libralign_v.so python bench.py --workload largebox --steps 2 --warmup 1 --particles 2640 --no-cpu-baseline --no-pcie --function none 2>gpurun_out/v.err | tail -1 |
        python -c "import sys,json; d=json.loads(sys.stdin.read()); r=d['roofline']; print('  particles/s %.0f ' % d['value'], {k: round(v['avg_launch_ms'],2) for k,v in r['kernels'].items()}, 'flips', d['parity']['sigma_1']['tie_flips'], d['parity']['sigma_0.25']['tie_flips'])"
    grep "zone plan" gpurun_out/v.err | head -1 | cut -c1-120
done
