# Run ON THE GPU BOX: VALU / LDS instruction and LDS-conflict counters of ccf_kernel with phases switched off
# (RALIGN_DEBUG 1: no inverse FFT phase, 2: no contraction phase), to attribute instructions to phases.
export TMPDIR=/tmp; root=$(pwd); cd /tmp
for d in 0 1 2; do
  RALIGN_DEBUG=$d rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS SQ_INSTS_VALU SQ_INSTS_SALU SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_WAVE_CYCLES --output-format csv -d /tmp/ccfp_$d -o run -- python3 $root/bench.py --steps 1 --warmup 0 --particles 14000 --no-cpu-baseline --function none > /tmp/ccfp_$d.log 2>&1
  python3 - <<PY
import csv,glob,collections
acc=collections.defaultdict(float); n=collections.defaultdict(set)
for f in glob.glob("/tmp/ccfp_$d/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if "ccf_kernel" in r["Kernel_Name"]:
            acc[r["Counter_Name"]]+=float(r["Counter_Value"]); n[r["Counter_Name"]].add(r["Dispatch_Id"])
print("dbg $d", {k: round(v/len(n[k])/1e6,1) for k,v in acc.items()})
PY
done
