#!/bin/bash
# Same-box A/B of two builds of librcfm.so: alternates them `reps` times on the cfg4 bench and prints the sorted
# ms_per_step of each (the pool's boxes differ by several percent, single runs by ~1 %: only alternation on one box
# resolves sub-percent changes).   tools/ab.sh build_ab/librcfm_a.so build_ab/librcfm_b.so [reps] [bench args...]
a=$1; b=$2; reps=${3:-6}; shift 3 || true
for i in $(seq "$reps"); do
    for v in "$a" "$b"; do
        ms=$(RCFM_LIB=$v python bench.py --steps 30 --warmup 5 --cpu-channels 0 --no-extras --no-pcie "$@" 2>/dev/null | grep -E -o '"ms_per_step": [0-9.]*' | head -1 | cut -d' ' -f2)
        echo "$(basename "$v") $ms"
    done
done | sort | awk '{a[$1]=a[$1]" "$2; s[$1]+=$2; n[$1]++} END {for (k in a) printf "%s mean %.4f :%s\n", k, s[k]/n[k], a[k]}'
