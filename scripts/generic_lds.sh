#!/bin/bash
# counters of the generic contraction kernel per phase: PMC passes of the -DRALIGN_PROFILE_SWITCHES build with phase-skip
# masks (2 = no contraction, 1 = no inverse FFT / argmax), bench.py --workload largebox
set -e
cd "$(dirname "$0")/.."
root=$PWD
mkdir -p gpurun_out
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -shared -DRALIGN_PROFILE_SWITCHES -Iinclude \
    -o gpurun_out/libralign_prof.so cryo_ralib_amd/csrc/ralign_engine.hip
export TMPDIR=/tmp
export RALIGN_LIB=$root/gpurun_out/libralign_prof.so
cd /tmp
for m in ${MASKS:-0 2 1}; do
    export RALIGN_DEBUG=$m
    for grp in "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS SQ_INSTS_VALU SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY" "FETCH_SIZE" "WRITE_SIZE" "SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR GRBM_GUI_ACTIVE SQ_INSTS_MFMA"; do
        out=$root/gpurun_out/glds_$m
        rm -rf $out
        rocprofv3 --pmc $grp --output-format csv -d $out -o run -- python3 $root/bench.py --workload largebox --steps 1 --warmup 0 --no-cpu-baseline --no-parity --no-pcie --function none > $out.log 2>&1
        python3 - $out $m <<'PY'
import csv, glob, sys
acc = {}
for f in glob.glob(sys.argv[1] + "/**/*counter_collection.csv", recursive=True):
    for row in csv.DictReader(open(f)):
        if "ccf_generic" not in row["Kernel_Name"]:
            continue
        acc.setdefault(row["Counter_Name"], {}).setdefault(row["Dispatch_Id"], 0.0)
        acc[row["Counter_Name"]][row["Dispatch_Id"]] += float(row["Counter_Value"])
print("mask", sys.argv[2], {k: round(sum(v.values()) / len(v) / 1e6, 1) for k, v in sorted(acc.items())}, "(millions per launch)")
PY
        find $out -name "*counter_collection.csv" -delete
    done
done
