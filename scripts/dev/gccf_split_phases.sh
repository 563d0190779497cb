#!/bin/bash
# what bounds the two halves of the split large-box contraction: profile build (-DRALIGN_PROFILE_SWITCHES), bench.py --workload
# largebox under rocprofv3 --kernel-trace --stats with RALIGN_DEBUG masks: 4 = the contraction's stores go to a cache-resident
# piece of the scratch, 8 = no stores, 16 = the transforms read a cache-resident piece (results are wrong, times are the point)
cd "$(dirname "$0")/../.."
root=$(pwd)
export TMPDIR=/tmp
mkdir -p gpurun_out
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -shared -DRALIGN_PROFILE_SWITCHES -Iinclude \
    -o gpurun_out/libralign_prof.so cryo_ralib_amd/csrc/ralign_engine.hip || exit 1
export RALIGN_LIB=$PWD/gpurun_out/libralign_prof.so
for m in ${MASKS:-0 4 8 16 20}; do
    export RALIGN_DEBUG=$m
    rm -rf gpurun_out/gsp_$m
    (cd /tmp && rocprofv3 --kernel-trace --stats --output-format csv -d $root/gpurun_out/gsp_$m -o p -- python3 $root/bench.py --workload largebox --steps 1 --warmup 1 --particles 4000 \
        --no-cpu-baseline --no-parity --no-pcie --no-others --function none > $root/gpurun_out/gsp_$m.log 2>&1) || { tail -5 gpurun_out/gsp_$m.log; exit 1; }
    python - $m <<'PY'
import csv, glob, sys
f = glob.glob("gpurun_out/gsp_%s/**/*kernel_stats.csv" % sys.argv[1], recursive=True)[0]
out = []
for r in csv.DictReader(open(f)):
    if "ccf_generic" in r["Name"] or "gccf_ifft" in r["Name"] or "polar_generic_kernel<false>" in r["Name"]:
        out.append("%s %d x %.3f ms" % (r["Name"].split("(")[0].replace("void ralign::", ""), int(r["Calls"]), float(r["AverageNs"]) / 1e6))
print("RALIGN_DEBUG=%-3s " % sys.argv[1] + "   ".join(out))
PY
    find gpurun_out/gsp_$m -name '*kernel_trace.csv' -delete
done
