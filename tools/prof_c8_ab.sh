cd "$GRAFT_REPO_ROOT"; export TMPDIR=/tmp
for m in nhwc c8; do
  OUT=gpurun_out/prof_$m; rm -rf $OUT; mkdir -p $OUT
  if [ $m = nhwc ]; then export AFLDM_NO_C8=1; else unset AFLDM_NO_C8; fi
  rocprofv3 --kernel-trace --stats --output-format csv -d $OUT -o trace -- python bench.py --steps 10 --warmup 2 --regions 1 --no-cpu-baseline --no-roofline --no-extras > $OUT/bench.log 2>&1
  F=$(find $OUT -name "*kernel_trace.csv" | head -1)
  python tools/trace_breakdown.py $F 60 > gpurun_out/c8ab_${m}_breakdown.txt 2>&1
done
