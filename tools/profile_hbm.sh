#!/bin/bash
# HBM-side traffic counters of one bench step, one rocprofv3 run per counter (MI355X_MICROARCH.md, HBM).
#   tools/profile_hbm.sh <tag> -> gpurun_out/<tag>_hbm.txt
tag=${1:-hbm}
root=$(cd "$(dirname "$0")/.." && pwd)
export TMPDIR=/tmp
cd /tmp
: > "$root/gpurun_out/${tag}_hbm.txt"
for c in FETCH_SIZE WRITE_SIZE; do
    out=$root/gpurun_out/$tag/$c
    mkdir -p "$out"
    rocprofv3 --kernel-trace --pmc $c -d "$out" -o run --output-format csv -- python "$root/bench.py" --steps 1 --warmup 1 --cpu-channels 0 --no-extras --no-pcie --placement-sets 1 > "$out/log.txt" 2>&1
    csv=$(find "$out" -name '*counter_collection.csv' | head -1)
    echo "== $c" >> "$root/gpurun_out/${tag}_hbm.txt"
    [ -n "$csv" ] && python "$root/tools/pmc_summary.py" "$csv" k_ >> "$root/gpurun_out/${tag}_hbm.txt" || tail -3 "$out/log.txt" >> "$root/gpurun_out/${tag}_hbm.txt"
done
find "$root/gpurun_out/$tag" -type f -size +8M -delete
