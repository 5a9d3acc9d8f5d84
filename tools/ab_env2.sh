#!/bin/bash
# like ab_env.sh, for settings of SEVERAL variables: tools/ab_env2.sh reps "<label>=<env settings>" ... -- bench args
#   tools/ab_env2.sh 4 "off=RCFM_BUF_PHASE=0 RCFM_FFT_PINGPONG=0" "on=RCFM_BUF_PHASE=1" -- --config cfg5
reps=$1; shift
sets=()
while [ "$1" != "--" ] && [ $# -gt 0 ]; do sets+=("$1"); shift; done
shift
for i in $(seq "$reps"); do
  for s in "${sets[@]}"; do
    label=${s%%=*}; envs=${s#*=}
    ms=$(env $envs python bench.py --steps 30 --warmup 5 --cpu-channels 0 --no-extras "$@" 2>/dev/null | grep -E -o '"ms_per_step": [0-9.]*' | head -1 | cut -d' ' -f2)
    echo "$label $ms"
  done
done | sort | awk '{k=$1; a[k]=a[k]" "$2; s[k]+=$2; n[k]++} END {for (k in a) printf "%-10s mean %.4f :%s\n", k, s[k]/n[k], a[k]}'
