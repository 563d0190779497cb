#!/bin/bash
# instruction counts of the fused search kernel by phase: profiling build (-DRALIGN_PROFILE_SWITCHES), one rocprofv3 --pmc run
# per RALIGN_DEBUG mask (0 full kernel, 16 no ring jobs, 2 no contraction, 4 no spectra store / inverse FFT); the difference
# to the full kernel is the phase.  Run ON THE GPU BOX: bash scripts/fused_inst_phases.sh > gpurun_out/fused_inst_phases.txt
set -e
root=$(cd "$(dirname "$0")/.." && pwd)
mkdir -p $root/gpurun_out
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -shared -DRALIGN_PROFILE_SWITCHES -I$root/include \
    -o $root/gpurun_out/libralign_prof.so $root/cryo_ralib_amd/csrc/ralign_engine.hip
export TMPDIR=/tmp RALIGN_LIB=$root/gpurun_out/libralign_prof.so
cd /tmp
for m in 0 16 2 4; do
    export RALIGN_DEBUG=$m
    out=$root/gpurun_out/instph_$m
    rm -rf $out
    rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_ACTIVE_INST_ANY SQ_BUSY_CYCLES GRBM_GUI_ACTIVE --output-format csv -d $out -o run -- \
        python3 $root/bench.py --steps 1 --warmup 0 --particles 7000 --no-cpu-baseline --no-parity --no-pcie --function none > $out.log 2>&1
    python3 - $out $m <<'PY'
import csv, glob, sys
acc = {}
for f in glob.glob(sys.argv[1] + "/**/*counter_collection.csv", recursive=True):
    for row in csv.DictReader(open(f)):
        if "search_fused" in row["Kernel_Name"]:
            acc[row["Counter_Name"]] = acc.get(row["Counter_Name"], 0.0) + float(row["Counter_Value"])
print("mask %-3s " % sys.argv[2] + "  ".join("%s %.0f" % (k.replace("SQ_INSTS_", "").replace("SQ_", ""), v / 7000) for k, v in sorted(acc.items())), "(per particle)")
PY
    find $out -name "*counter_collection.csv" -delete
done
