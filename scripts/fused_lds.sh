#!/bin/bash
# LDS counters of the fused search kernel per phase: PMC passes of the -DRALIGN_PROFILE_SWITCHES build with phase-skip masks
set -e
cd "$(dirname "$0")/.."
root=$PWD
mkdir -p gpurun_out
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -shared -DRALIGN_PROFILE_SWITCHES -Iinclude \
    -o gpurun_out/libralign_prof.so cryo_ralib_amd/csrc/ralign_engine.hip
export TMPDIR=/tmp
export RALIGN_LIB=$root/gpurun_out/libralign_prof.so
cd /tmp
for m in ${MASKS:-0 16 2 4}; do
    export RALIGN_DEBUG=$m
    out=$root/gpurun_out/lds_$m
    rm -rf $out
    rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS SQ_INSTS_VALU SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY \
        --output-format csv -d $out -o run -- python3 $root/bench.py --steps 1 --warmup 0 --particles 14000 --no-cpu-baseline --no-parity --no-pcie --function none > $out.log 2>&1
    python3 - $out $m <<'PY'
import csv, glob, sys
acc = {}
for f in glob.glob(sys.argv[1] + "/**/*counter_collection.csv", recursive=True):
    for row in csv.DictReader(open(f)):
        if "search_fused" not in row["Kernel_Name"]:
            continue
        acc.setdefault(row["Counter_Name"], {}).setdefault(row["Dispatch_Id"], 0.0)
        acc[row["Counter_Name"]][row["Dispatch_Id"]] += float(row["Counter_Value"])
print("mask", sys.argv[2], {k: round(sum(v.values()) / len(v) / 7000) for k, v in sorted(acc.items())}, "(per particle)")
PY
    find $out -name "*counter_collection.csv" -delete
done
