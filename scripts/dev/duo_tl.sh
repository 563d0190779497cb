#!/bin/bash
# wave timeline of search_duo_kernel on one workload (profiling build)
set -e
cd "$(dirname "$0")/../.."
mkdir -p gpurun_out
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -shared -DRALIGN_PROFILE_SWITCHES -Iinclude \
    -o gpurun_out/libralign_prof.so cryo_ralib_amd/csrc/ralign_engine.hip
for w in ${@:-box128}; do
RALIGN_LIB=$PWD/gpurun_out/libralign_prof.so RALIGN_TIMELINE=$PWD/gpurun_out/duo_tl_$w.bin \
    python bench.py --workload $w --steps 1 --warmup 0 --particles 2048 --no-cpu-baseline --no-parity --no-pcie --no-others --function none > gpurun_out/duo_tl_$w.log 2>&1
echo "== $w"; python scripts/duo_timeline.py gpurun_out/duo_tl_$w.bin
done
