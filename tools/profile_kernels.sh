#!/bin/bash
# Per-kernel summary of one bench run: rocprofv3 --kernel-trace --stats, then the top-kernel table.
#   tools/profile_kernels.sh <tag> [bench.py arguments...]   -> gpurun_out/<tag>_kernel_stats.md
set -e
tag=${1:-prof}; shift || true
root=$(cd "$(dirname "$0")/.." && pwd)
out=$root/gpurun_out/$tag
mkdir -p "$out"
export TMPDIR=/tmp
cd /tmp
# the traced process runs ONLY the headline path (--no-extras, no CPU baseline): 2 warm-up + 1 profile pass + 5 timed steps
rocprofv3 --kernel-trace --stats -d "$out" -o run -- python "$root/bench.py" --steps 5 --warmup 2 --cpu-channels 0 --no-extras --no-pcie --placement-sets 1 "$@" > "$out/bench.log" 2>&1 || true
db=$(find "$out" -name '*.db' | head -1)
if [ -n "$db" ]; then
    python "$root/tools/rocpd_summary.py" "$db" --per-step 8 > "$root/gpurun_out/${tag}_kernel_stats.md"
    echo >> "$root/gpurun_out/${tag}_kernel_stats.md"
    echo "Every kernel of the traced process (input synthesis included):" >> "$root/gpurun_out/${tag}_kernel_stats.md"
    echo >> "$root/gpurun_out/${tag}_kernel_stats.md"
    python "$root/tools/rocpd_summary.py" "$db" 40 >> "$root/gpurun_out/${tag}_kernel_stats.md"
else
    csv=$(find "$out" -name '*kernel_stats.csv' | head -1)
    cp "$csv" "$root/gpurun_out/${tag}_kernel_stats.csv"
fi
grep "^{\"metric\"" "$out/bench.log" | tail -1 > "$root/gpurun_out/${tag}_profiled_bench.json"
find "$out" -type f -size +8M -delete
