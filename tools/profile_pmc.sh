#!/bin/bash
# SQ counter passes over one bench run (one rocprofv3 run per counter group, --kernel-trace only).
#   tools/profile_pmc.sh <tag> [bench.py arguments...] -> gpurun_out/<tag>_pmc.txt
tag=${1:-pmc}; shift || true
root=$(cd "$(dirname "$0")/.." && pwd)
export TMPDIR=/tmp
cd /tmp
: > "$root/gpurun_out/${tag}_pmc.txt"
i=0
for grp in "SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY" \
           "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA" \
           "SQ_INSTS_VALU SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE" \
           "SQ_BUSY_CYCLES SQ_WAVES SQ_WAIT_INST_LDS GRBM_GUI_ACTIVE"; do
    i=$((i+1))
    out=$root/gpurun_out/$tag/g$i
    mkdir -p "$out"
    rocprofv3 --kernel-trace --pmc $grp -d "$out" -o run --output-format csv -- python "$root/bench.py" --steps 1 --warmup 1 --cpu-channels 0 --no-extras --no-pcie --placement-sets 1 "$@" > "$out/log.txt" 2>&1
    csv=$(find "$out" -name '*counter_collection.csv' | head -1)
    echo "== $grp" >> "$root/gpurun_out/${tag}_pmc.txt"
    [ -n "$csv" ] && python "$root/tools/pmc_summary.py" "$csv" k_ >> "$root/gpurun_out/${tag}_pmc.txt" || tail -3 "$out/log.txt" >> "$root/gpurun_out/${tag}_pmc.txt"
done
find "$root/gpurun_out/$tag" -type f -size +8M -delete
