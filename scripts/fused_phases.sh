#!/bin/bash
# in-situ phase timing of the fused search kernel: a separate build with -DRALIGN_PROFILE_SWITCHES (the shipped
# library has no such switches), run through bench.py with RALIGN_DEBUG phase-skip masks:
#   16 = no ring jobs (sampling + ring FFT), 2 = no contraction, 1 = no inverse FFT / argmax, 4 = no spectra rounds at all,
#   contraction only: 32 = every B request hits one L1 line, 64 = every A read hits one LDS word, 128 = no matrix instructions
set -e
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -shared -DRALIGN_PROFILE_SWITCHES -Iinclude \
    -o gpurun_out/libralign_prof.so cryo_ralib_amd/csrc/ralign_engine.hip
for m in ${MASKS:-0 16 2 18 1 4 22}; do
    echo "RALIGN_DEBUG=$m"
    RALIGN_LIB=$PWD/gpurun_out/libralign_prof.so RALIGN_DEBUG=$m python bench.py --steps 4 --warmup 1 --no-cpu-baseline --function none 2>&1 | tail -1 |
        python -c "import sys,json; d=json.loads(sys.stdin.read()); r=d['roofline']; print('  particles/s %.0f  search launch %.3f ms' % (d['value'], r.get('avg_launch_ms', 0)))"
done
