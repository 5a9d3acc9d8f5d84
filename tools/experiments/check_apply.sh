#!/bin/bash
# Every patch under tools/experiments/ is a measured experiment that was not adopted, kept as a diff against the commit
# it was measured on (BASES below).  This checks that each one still applies THERE (git apply --check in a throw-away
# worktree of that commit) and reports whether it also applies to HEAD.   tools/experiments/check_apply.sh
# Needs the git history (build container); exit status 1 when a patch no longer applies to its base.
set -u
root=$(cd "$(dirname "$0")/../.." && pwd)
cd "$root"
declare -A BASES=(
  [r02_pair_interleave.patch]=dd89fbdb3579
  [r03_alternating_tile_order.patch]=4ab2b8c5d7c0
  [r03_decim_persist_prefetch.patch]=f87835340afc
  [r04_lds_dma_persist_ablate.patch]=e9a9a929e69b
  [r04_narrow_tiles_per_stage.patch]=a2261f836d1f
  [r05_quad_plan.patch]=bf38dd2dfbd0
  [r05_wave_owned_mask_tile.patch]=f9b41baa5aab
  [r06_big_tile_probes.patch]=c2c60cbfe553
  [r06_twiddle_table.patch]=7118f5dd8967
  [r06_fir_state_merge.patch]=dd4b3e5a84e6
)
bad=0
wt=$(mktemp -d /tmp/rcfm_patch_check.XXXXXX)
for p in tools/experiments/*.patch; do
  name=$(basename "$p")
  base=${BASES[$name]:-}
  if [ -z "$base" ]; then echo "NO BASE  $name (add it to BASES)"; bad=1; continue; fi
  rm -rf "$wt"; git worktree add -q --detach "$wt" "$base" 2>/dev/null || { echo "NO COMMIT $name base $base"; bad=1; continue; }
  if (cd "$wt" && git apply --check "$root/$p" 2>/dev/null); then at_base=ok; else at_base=FAILS; bad=1; fi
  git worktree remove --force "$wt"
  if git apply --check "$p" 2>/dev/null; then at_head="applies to HEAD too"; else at_head="HEAD has moved on"; fi
  echo "$at_base  $name  at $base  ($at_head)"
done
rm -rf "$wt"; git worktree prune
exit $bad
