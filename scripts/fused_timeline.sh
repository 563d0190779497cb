#!/bin/bash
# wave timeline of the fused search kernel: profiling build (-DRALIGN_PROFILE_SWITCHES) with RALIGN_TIMELINE=<file>; workgroup
# 0 stamps clock64() per wave at the phase boundaries of every pass of its first particle; scripts/fused_timeline.py prints
# when each wave finished each phase (how long it then waited at the barrier)
set -e
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -shared -DRALIGN_PROFILE_SWITCHES -Iinclude \
    -o gpurun_out/libralign_prof.so cryo_ralib_amd/csrc/ralign_engine.hip
RALIGN_LIB=$PWD/gpurun_out/libralign_prof.so RALIGN_TIMELINE=$PWD/gpurun_out/fused_timeline.bin \
    python bench.py --workload ${1:-mref} --steps 1 --warmup 0 --particles 7000 --no-cpu-baseline --no-parity --no-pcie --function none > gpurun_out/fused_timeline.log 2>&1
python scripts/fused_timeline.py gpurun_out/fused_timeline.bin
