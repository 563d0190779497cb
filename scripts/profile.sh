#!/bin/bash
# Run ON THE GPU BOX (through gpurun): rocprofv3 kernel-trace statistics and PMC passes of bench.py.
#   bash scripts/profile.sh <tag> [bench args...]
# Results land in gpurun_out/prof_<tag>/ ; scripts/pmc_summary.py condenses them into profiles/.
set -o pipefail
tag=$1; shift
root=$(pwd)
out=$root/gpurun_out/prof_$tag
mkdir -p $out
export TMPDIR=/tmp
cd /tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $out/trace -o run -- python3 $root/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-parity --no-pcie --no-others "$@" > $out/trace.log 2>&1 || exit 1
# counters: one group per pass, smaller run (2 launches of each hot kernel)
for grp in "FETCH_SIZE" "WRITE_SIZE" "TCC_HIT_sum TCC_MISS_sum" \
           "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS" \
           "SQ_VALU_MFMA_BUSY_CYCLES SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR GRBM_GUI_ACTIVE"; do
    name=$(echo $grp | cut -c1-10 | tr ' ' '_')
    rocprofv3 --pmc $grp --output-format csv -d $out/pmc_$name -o run -- python3 $root/bench.py --steps 1 --warmup 0 --particles 14000 --no-cpu-baseline --no-parity --no-pcie --no-others "$@" > $out/pmc_$name.log 2>&1 || exit 1
done
# condense on the box (the raw counter CSVs exceed what gpurun copies back) and drop the raw files
mkdir -p $root/gpurun_out/profiles_$tag
python3 $root/scripts/pmc_summary.py $tag $tag $root/gpurun_out/profiles_$tag "$*"
find $out -name "*counter_collection.csv" -delete
find $out -name "*kernel_trace.csv" -delete
echo done
