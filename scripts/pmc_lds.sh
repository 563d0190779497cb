export TMPDIR=/tmp; root=$(pwd); cd /tmp
for d in ${DBGS:-0 16 64}; do
  RALIGN_DEBUG=$d rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS SQ_INSTS_VALU --output-format csv -d /tmp/lds_$d -o run -- python3 $root/bench.py --steps 1 --warmup 0 --particles 14000 --no-cpu-baseline --function none > /tmp/lds_$d.log 2>&1
  python3 - <<PY
import csv,glob,collections
acc=collections.defaultdict(float); n=collections.defaultdict(set)
for f in glob.glob("/tmp/lds_$d/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if "polar_fft_kernel" in r["Kernel_Name"] and "ref_" not in r["Kernel_Name"]:
            acc[r["Counter_Name"]]+=float(r["Counter_Value"]); n[r["Counter_Name"]].add(r["Dispatch_Id"])
print("dbg $d", {k: round(v/len(n[k])/1e6,1) for k,v in acc.items()})
PY
done
