#!/bin/bash
# The measurement set of a round at the final code (GPU box): tests, the driver-style bench line, per-kernel traces, the
# memory-side traffic passes, fuzz + soak logs, the reference's harness shapes.   tools/measure_set.sh <tag>
# Everything lands under gpurun_out/<tag>*; copy what is to be judged into profiles/ (and run traffic_summary.py --json
# in the checkout the counters were collected on, so that the provenance stamp matches).
tag=${1:-set}; R=$(cd "$(dirname "$0")/.." && pwd); cd "$R"; O=gpurun_out; mkdir -p $O
rm -f $O/oracle_samples.log
(python -m pytest tests -m gpu -q 2>&1 | tail -8) > $O/${tag}_gputests.log
cat $O/oracle_samples.log >> $O/${tag}_gputests.log 2>/dev/null   # which channels met the oracle this time (tests/test_hip_whole_output.py)
python bench.py --steps 20 --warmup 3 --profile-all > $O/${tag}_bench.json 2> $O/${tag}_stages.txt
tools/profile_kernels.sh ${tag}
tools/profile_kernels.sh ${tag}_cfg3 --config cfg3
tools/profile_kernels.sh ${tag}_cfg5 --config cfg5
tools/profile_traffic.sh ${tag} python bench.py --steps 1 --warmup 1 --cpu-channels 0 --no-extras --no-pcie --placement-sets 1
{ echo "== fuzz_fft 200 seed 51"; python tools/fuzz_fft.py 200 51 2>&1 | tail -4; echo "== fuzz_pipeline 150 seed 52"; python tools/fuzz_pipeline.py 150 52 2>&1 | tail -6;
  echo "== soak_determinism 30 cfg4"; python tools/soak_determinism.py 30 cfg4 2>&1 | tail -4; echo "== soak_determinism 50 cfg5"; python tools/soak_determinism.py 50 cfg5 2>&1 | tail -4;
  echo "== soak_lanes cfg3 60 2"; python tools/soak_lanes.py cfg3 60 2 2>&1 | tail -3; echo "== soak_lanes cfg4 12 2"; python tools/soak_lanes.py cfg4 12 2 2>&1 | tail -3; } > $O/${tag}_fuzz_and_soak.txt
python bench/reference_shapes.py > $O/${tag}_reference_shapes.txt 2>&1
find $O/${tag}* -type f -size +8M -delete
grep -E "passed|failed" $O/${tag}_gputests.log; head -c 600 $O/${tag}_bench.json; echo; tail -12 $O/${tag}_fuzz_and_soak.txt
