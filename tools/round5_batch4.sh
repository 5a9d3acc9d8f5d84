# per-kernel times and SQ counters of the two-pass 4-line plan of N = 1e7 against the default three-pass plan
O=$PWD/gpurun_out/r05d; mkdir -p $O; export TMPDIR=/tmp; R=$PWD; cd /tmp
for plan in default 3125,3200; do
  tag=${plan/,/x}; arg=""; [ "$plan" != default ] && arg="--plan $plan"
  rocprofv3 --kernel-trace --stats -d $O/trace_$tag -o run --output-format csv -- python $R/tools/bench_fft.py 10000000 $arg > $O/trace_$tag.log 2>&1
  f=$(find $O/trace_$tag -name '*kernel_stats.csv' | head -1); [ -n "$f" ] && { echo "== $plan"; grep -i "k_fft" $f | cut -c1-260; } >> $O/quad_kernel_stats.txt
  for grp in "SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY" "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT" "SQ_WAIT_INST_LDS SQ_LDS_IDX_ACTIVE SQ_BUSY_CYCLES SQ_WAVES"; do
    d=$O/pmc_${tag}_$(echo $grp | cut -d' ' -f1); mkdir -p $d
    rocprofv3 --kernel-trace --pmc $grp -d $d -o run --output-format csv -- python $R/tools/bench_fft.py 10000000 $arg > $d/log.txt 2>&1
    csv=$(find $d -name '*counter_collection.csv' | head -1)
    echo "== $plan : $grp" >> $O/quad_pmc.txt
    [ -n "$csv" ] && python $R/tools/pmc_summary.py $csv k_fft >> $O/quad_pmc.txt
  done
done
find $O -type f -size +4M -delete
cat $O/quad_kernel_stats.txt; head -60 $O/quad_pmc.txt
