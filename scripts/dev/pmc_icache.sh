#!/bin/bash
# instruction-cache counters of the search kernels (run ON THE GPU BOX): requests, hits, misses per workload
set -o pipefail
root=$(pwd)
export TMPDIR=/tmp
cd /tmp
for w in ${@:-mref nb00 box128}; do
  out=$root/gpurun_out/ic_$w
  rm -rf $out; mkdir -p $out
  rocprofv3 --pmc SQC_ICACHE_REQ SQC_ICACHE_HITS SQC_ICACHE_MISSES SQ_WAVE_CYCLES SQ_IFETCH --output-format csv -d $out -o run -- python3 $root/bench.py --workload $w --steps 1 --warmup 0 --particles 4096 --no-cpu-baseline --no-parity --no-pcie --no-others > $out/log.txt 2>&1 || { tail -5 $out/log.txt; continue; }
  python3 - $out $w <<'PY'
import sys, csv, glob, collections
out, w = sys.argv[1], sys.argv[2]
acc = collections.defaultdict(lambda: collections.defaultdict(float))
for f in glob.glob(out + "/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"]
        if "search_" not in k: continue
        acc[k.split("(")[0][:60]][r["Counter_Name"]] += float(r["Counter_Value"])
for k, v in acc.items():
    print(w, k, {a: int(b) for a, b in v.items()})
PY
  find $out -name "*counter_collection.csv" -delete
done
