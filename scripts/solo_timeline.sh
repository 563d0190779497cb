#!/bin/bash
# wave timeline of search_solo_kernel (maxrin 512): profiling build (-DRALIGN_PROFILE_SWITCHES) with RALIGN_TIMELINE=<file>; workgroup
# 0 stamps clock64() per wave at the phase boundaries of its first 64 passes; scripts/solo_timeline.py prints the per-wave phase times
#   bash scripts/solo_timeline.sh [box128 | nb00]
set -e
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -shared -DRALIGN_PROFILE_SWITCHES -Iinclude \
    -o gpurun_out/libralign_prof.so cryo_ralib_amd/csrc/ralign_engine.hip
RALIGN_LIB=$PWD/gpurun_out/libralign_prof.so RALIGN_TIMELINE=$PWD/gpurun_out/solo_timeline.bin \
    python bench.py --workload ${1:-box128} --steps 1 --warmup 0 --particles 2048 --no-cpu-baseline --no-parity --no-pcie --function none > gpurun_out/solo_timeline.log 2>&1
python scripts/solo_timeline.py gpurun_out/solo_timeline.bin
