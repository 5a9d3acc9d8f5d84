mkdir -p gpurun_out/r05c; O=gpurun_out/r05c
(python -m pytest tests -m gpu -q 2>&1 | tail -15) > $O/gputests.log
python tools/bench_fft.py 10000000 > $O/fft_1e7.txt 2>&1
python tools/bench_fft.py 10000000 --plan 3125,3200 >> $O/fft_1e7.txt 2>&1
python tools/bench_fft.py 10000000 --plan 3200,3125 >> $O/fft_1e7.txt 2>&1
for i in 1 2 3 4 5; do python bench.py --no-extras --cpu-channels 0 --no-pcie 2>/dev/null | python -c "import sys,json; r=json.loads(sys.stdin.readline()); print('default', r['ms_per_step'], r['placement']['ms_per_step_by_set'])"; done > $O/placement_runs.txt
for i in 1 2 3 4 5; do python bench.py --no-extras --cpu-channels 0 --no-pcie --arena 1 2>/dev/null | python -c "import sys,json; r=json.loads(sys.stdin.readline()); print('arena  ', r['ms_per_step'], r['placement']['ms_per_step_by_set'])"; done >> $O/placement_runs.txt
tools/profile_pmc.sh r05c_cfg5_pmc --config cfg5 > /dev/null 2>&1
tail -3 $O/gputests.log; cat $O/fft_1e7.txt | grep -v amdgpu; cat $O/placement_runs.txt
