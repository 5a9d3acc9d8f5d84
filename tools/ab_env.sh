#!/bin/bash
# alternates env settings on one box: tools/ab_env.sh "<envA>" "<envB>" reps -- bench args
a=$1; b=$2; reps=$3; shift 4
for i in $(seq "$reps"); do
  for v in "$a" "$b"; do
    ms=$(env $v python bench.py --steps 30 --warmup 5 --cpu-channels 0 --no-extras "$@" 2>/dev/null | grep -E -o '"ms_per_step": [0-9.]*' | head -1 | cut -d' ' -f2)
    echo "[$v] $ms"
  done
done | sort | awk '{k=$1; a[k]=a[k]" "$2; s[k]+=$2; n[k]++} END {for (k in a) printf "%s mean %.4f :%s\n", k, s[k]/n[k], a[k]}'
