#!/bin/bash
# alternates two settings of bench.py's arguments on one box (kernel options are per-handle arguments now, not
# environment variables):   tools/ab_args.sh "<argsA>" "<argsB>" reps -- common bench args
#   tools/ab_args.sh "--opt pilot_chain=0" "" 4 -- --config cfg4
a=$1; b=$2; reps=$3; shift 3; [ "$1" = "--" ] && shift
for i in $(seq "$reps"); do
  for v in "$a" "$b"; do
    ms=$(python bench.py --steps 30 --warmup 5 --cpu-channels 0 --no-extras --no-pcie --placement-sets 1 $v "$@" 2>/dev/null | grep -E -o '"ms_per_step": [0-9.]*' | head -1 | cut -d' ' -f2)
    echo "${v:-default}|$ms"
  done
done | sort | awk -F'|' '{a[$1]=a[$1]" "$2; s[$1]+=$2; n[$1]++} END {for (k in a) printf "[%s] mean %.4f :%s\n", k, s[k]/n[k], a[k]}'
