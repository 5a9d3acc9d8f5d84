#!/bin/bash
# Memory-side traffic of one command, one rocprofv3 run per counter group (--kernel-trace --pmc only; the TCC block
# has four counter slots and FETCH_SIZE alone takes three: MI355X_MICROARCH.md, rocprofv3 PMC slots):
#   tools/profile_traffic.sh <tag> <command...>     -> gpurun_out/<tag>/<group>/..., gpurun_out/<tag>_traffic_raw.txt
# tools/traffic_summary.py turns the raw table into per-kernel bytes (profiles/*_hbm_traffic.md, hbm_traffic.json).
tag=$1; shift
root=$(cd "$(dirname "$0")/.." && pwd)
export TMPDIR=/tmp
cd /tmp
raw=$root/gpurun_out/${tag}_traffic_raw.txt
: > "$raw"
i=0
for grp in "FETCH_SIZE" "WRITE_SIZE" \
           "TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum TCC_EA0_RDREQ_64B_sum TCC_EA0_RDREQ_128B_sum" \
           "TCC_EA0_WRREQ_sum TCC_EA0_WRREQ_64B_sum"; do
    i=$((i+1))
    out=$root/gpurun_out/$tag/g$i
    mkdir -p "$out"
    (cd "$root" && rocprofv3 --kernel-trace --pmc $grp -d "$out" -o run --output-format csv -- "$@" > "$out/log.txt" 2>&1)
    csv=$(find "$out" -name '*counter_collection.csv' | head -1)
    echo "== $grp" >> "$raw"
    if [ -n "$csv" ]; then python "$root/tools/pmc_summary.py" "$csv" "" >> "$raw"; else tail -3 "$out/log.txt" >> "$raw"; fi
done
find "$root/gpurun_out/$tag" -type f -size +8M -delete
