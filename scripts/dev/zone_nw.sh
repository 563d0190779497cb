#!/bin/bash
# polar_zone_kernel with 8 / 12 / 16 waves per workgroup (register budget 256 / 170 / 128, zones shrink with the wave buffers)
cd "$(dirname "$0")/../.."
mkdir -p gpurun_out
for nw in ${NWS:-8 16}; do
    /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -shared -DRA_ZONE_NW=$nw -Iinclude -o gpurun_out/libralign_nw$nw.so cryo_ralib_amd/csrc/ralign_engine.hip || exit 1
    echo "NW=$nw"
    RALIGN_INFO=1 RALIGN_LIB=$PWD/gpurun_out/libralign_nw$nw.so python bench.py --workload largebox --steps 2 --warmup 1 --particles 2640 --no-cpu-baseline --no-parity --no-pcie --function none 2>gpurun_out/nw$nw.err | tail -1 |
        python -c "import sys,json; d=json.loads(sys.stdin.read()); r=d['roofline']; print('  particles/s %.0f ' % d['value'], {k: round(v['avg_launch_ms'],2) for k,v in r['kernels'].items()})"
    grep "zone plan" gpurun_out/nw$nw.err | head -1
done
