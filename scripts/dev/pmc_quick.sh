#!/bin/bash
# quick counter comparison of the search kernel under environment switches: bash scripts/dev/pmc_quick.sh TAG [ENV=VAL ...]
# (run ON THE GPU BOX; two PMC passes of bench.py --steps 1 --particles 14000, summary to gpurun_out/pmcq_TAG.txt)
tag=$1; shift
root=$(pwd)
for kv in "$@"; do export "$kv"; done
export TMPDIR=/tmp
cd /tmp
for grp in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS" \
           "SQ_VALU_MFMA_BUSY_CYCLES SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR GRBM_GUI_ACTIVE"; do
    name=$(echo $grp | cut -c1-10 | tr ' ' '_')
    out=$root/gpurun_out/pmcq_${tag}_$name
    rm -rf $out
    rocprofv3 --pmc $grp --output-format csv -d $out -o run -- python3 $root/bench.py --steps 1 --warmup 0 --particles 14000 --no-cpu-baseline --no-parity --no-pcie --no-others ${BENCH_ARGS} > $out.log 2>&1 || exit 1
done
python3 - $root/gpurun_out $tag <<'PY' > $root/gpurun_out/pmcq_$tag.txt
import csv, glob, sys
acc = {}
for f in glob.glob(sys.argv[1] + "/pmcq_" + sys.argv[2] + "_*/**/*counter_collection.csv", recursive=True):
    per = {}
    for row in csv.DictReader(open(f)):
        k = row["Kernel_Name"].split("(")[0]
        if "search_" not in k and "generic" not in k and "gccf" not in k: continue
        per.setdefault((k, row["Counter_Name"]), {}).setdefault(row["Dispatch_Id"], 0.0)
        per[(k, row["Counter_Name"])][row["Dispatch_Id"]] += float(row["Counter_Value"])
    for (k, c), d in per.items():
        acc.setdefault(k, {})[c] = sum(d.values()) / len(d)
for k, e in acc.items():
    print(k)
    for c in sorted(e): print("   %-28s %14.0f   per particle %10.1f" % (c, e[c], e[c] / 7000))
PY
find $root/gpurun_out -path "*pmcq_${tag}_*" -name "*.csv" -delete
cat $root/gpurun_out/pmcq_$tag.txt
