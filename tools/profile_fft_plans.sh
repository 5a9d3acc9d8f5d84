#!/bin/bash
# Per-pass kernel times of the wideband FFT under several plans: rocprofv3 --kernel-trace --stats of tools/bench_fft.py,
# one traced process per plan.   tools/profile_fft_plans.sh <tag> <n> "<bench_fft args of plan 1>" "<... plan 2>" ...
# -> gpurun_out/<tag>_fft_passes.md
tag=$1; n=$2; shift 2
root=$(cd "$(dirname "$0")/.." && pwd)
export TMPDIR=/tmp
out=$root/gpurun_out/${tag}_fft_passes.md
: > "$out"
i=0
for plan in "$@"; do
    i=$((i + 1))
    d=$root/gpurun_out/${tag}_fftprof_$i
    mkdir -p "$d"
    (cd /tmp && rocprofv3 --kernel-trace --stats -d "$d" -o run -- python "$root/tools/bench_fft.py" "$n" $plan > "$d/log.txt" 2>&1)
    echo "### bench_fft.py $n $plan" >> "$out"
    grep "engine" "$d/log.txt" >> "$out"
    echo >> "$out"
    db=$(find "$d" -name '*.db' | head -1)
    if [ -n "$db" ]; then python "$root/tools/rocpd_summary.py" "$db" 12 rcfm >> "$out"; fi
    echo >> "$out"
    rm -rf "$d"
done
cat "$out"
