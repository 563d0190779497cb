#!/bin/bash
# wave timeline of search_duo_kernel under two settings of the slice width (RALIGN_DUO_NQT) -- which phase the narrower slice speeds up
set -e
cd "$(dirname "$0")/../.."
mkdir -p gpurun_out
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -shared -DRALIGN_PROFILE_SWITCHES -Iinclude \
    -o gpurun_out/libralign_prof.so cryo_ralib_amd/csrc/ralign_engine.hip
for q in 14 16; do
RALIGN_DUO_PF=1 RALIGN_DUO_NQT=$q RALIGN_LIB=$PWD/gpurun_out/libralign_prof.so RALIGN_TIMELINE=$PWD/gpurun_out/duo_tl_$q.bin \
    python bench.py --workload ${1:-nb00} --steps 1 --warmup 0 --particles 2048 --no-cpu-baseline --no-parity --no-pcie --no-others --function none > gpurun_out/duo_tl_$q.log 2>&1
echo "== NQT $q"; python scripts/duo_timeline.py gpurun_out/duo_tl_$q.bin | tail -3
done
