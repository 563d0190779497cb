#!/bin/bash
# in-situ phase timing of polar_zone_kernel (large box): profile build (-DRALIGN_PROFILE_SWITCHES) run through
# bench.py --workload largebox with RALIGN_DEBUG masks: 512 = no image taps, 256 = no ring FFT, 2048 = no split step, 1024 = no panel stores
set -e
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -shared -DRALIGN_PROFILE_SWITCHES -Iinclude \
    -o gpurun_out/libralign_prof.so cryo_ralib_amd/csrc/ralign_engine.hip
for m in ${MASKS:-0 512 256 2048 1024 768 2816 3840}; do
    echo "RALIGN_DEBUG=$m $EXTRA"
    env $EXTRA RALIGN_LIB=$PWD/gpurun_out/libralign_prof.so RALIGN_DEBUG=$m python bench.py --workload largebox --steps 1 --warmup 1 --particles 2640 --no-cpu-baseline --no-parity --no-pcie --function none 2>&1 | tail -1 |
        python -c "import sys,json; d=json.loads(sys.stdin.read()); r=d['roofline']; print('  particles/s %.0f ' % d['value'], {k: round(v['avg_launch_ms'],2) for k,v in r['kernels'].items()})"
done
