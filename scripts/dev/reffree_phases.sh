#!/bin/bash
# reference-free workload (BASELINE configs[2]) through the profiling build with phase-skip masks: how much of a pass is the
# one-reference tail (contraction, spectra store, inverse FFT) -- the ceiling of any scheme that hides it behind the ring jobs
#   16 = no ring jobs, 2 = no contraction, 1 = no inverse FFT / argmax, 4 = no spectra rounds at all
set -e
cd "$(dirname "$0")/../.."
mkdir -p gpurun_out
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -shared -DRALIGN_PROFILE_SWITCHES -Iinclude -o gpurun_out/libralign_prof.so cryo_ralib_amd/csrc/ralign_engine.hip
for m in ${MASKS:-0 2 1 4 6 7 16}; do
    echo "RALIGN_DEBUG=$m"
    RALIGN_LIB=$PWD/gpurun_out/libralign_prof.so RALIGN_DEBUG=$m python bench.py --workload reffree --steps 4 --warmup 1 --no-cpu-baseline --no-parity --no-pcie --function none 2>&1 | tail -1 |
        python -c "import sys,json; d=json.loads(sys.stdin.read()); r=d['roofline']; print('  particles/s %.0f  search launch %.3f ms' % (d['value'], r.get('avg_launch_ms', 0)))"
done
