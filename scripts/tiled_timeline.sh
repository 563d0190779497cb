#!/bin/bash
# wave timeline of search_tiled_kernel (as scripts/fused_timeline.sh): profiling build, bench workload mref50 by default
set -e
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -shared -DRALIGN_PROFILE_SWITCHES -Iinclude \
    -o gpurun_out/libralign_prof.so cryo_ralib_amd/csrc/ralign_engine.hip
RALIGN_LIB=$PWD/gpurun_out/libralign_prof.so RALIGN_TIMELINE=$PWD/gpurun_out/tiled_timeline.bin \
    python bench.py --workload ${1:-mref50} --steps 1 --warmup 0 --particles 7000 --no-cpu-baseline --no-parity --no-pcie --function none > gpurun_out/tiled_timeline.log 2>&1
python scripts/tiled_timeline.py gpurun_out/tiled_timeline.bin
