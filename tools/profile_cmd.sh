#!/bin/bash
# Kernel durations and SQ / memory-side counters of ONE command (one rocprofv3 run per counter group, --kernel-trace only):
#   tools/profile_cmd.sh <tag> <kernel-name filter> <command...>   -> gpurun_out/<tag>_cmd_pmc.txt
tag=$1; filt=$2; shift 2
root=$(cd "$(dirname "$0")/.." && pwd)
export TMPDIR=/tmp
cd /tmp
res=$root/gpurun_out/${tag}_cmd_pmc.txt
: > "$res"
i=0
for grp in "SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY" \
           "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_BUSY_CYCLES" \
           "SQ_INSTS_VALU SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE" \
           "TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_128B_sum TCC_EA0_WRREQ_sum TCC_EA0_WRREQ_64B_sum" \
           "TCC_HIT_sum TCC_MISS_sum TCP_TCC_READ_REQ_sum TCP_TCC_WRITE_REQ_sum"; do
    i=$((i+1))
    out=$root/gpurun_out/$tag/g$i
    mkdir -p "$out"
    (cd "$root" && rocprofv3 --kernel-trace --pmc $grp -d "$out" -o run --output-format csv -- "$@" > "$out/log.txt" 2>&1)
    csv=$(find "$out" -name '*counter_collection.csv' | head -1)
    echo "== $grp" >> "$res"
    if [ -n "$csv" ]; then python "$root/tools/pmc_summary.py" "$csv" "$filt" >> "$res"; else tail -3 "$out/log.txt" >> "$res"; fi
done
out=$root/gpurun_out/$tag/trace
mkdir -p "$out"
(cd "$root" && rocprofv3 --kernel-trace --stats -d "$out" -o run --output-format csv -- "$@" > "$out/log.txt" 2>&1)
csv=$(find "$out" -name '*kernel_stats.csv' | head -1)
echo "== kernel trace (name, calls, total ns, average ns, ...)" >> "$res"
[ -n "$csv" ] && grep -E "Name|$filt" "$csv" | cut -c1-260 >> "$res"
find "$root/gpurun_out/$tag" -type f -size +8M -delete
