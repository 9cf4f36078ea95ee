cd "$GRAFT_REPO_ROOT"; export TMPDIR=/tmp
TAG=r02m
bash profiles/run_pmc_conv3x3.sh $TAG > gpurun_out/pmc_$TAG.log 2>&1
bash profiles/run_profile.sh $TAG > gpurun_out/prof_$TAG.log 2>&1
F=$(find gpurun_out/prof_$TAG -name "*kernel_trace.csv" | head -1)
python tools/trace_breakdown.py $F 60 > gpurun_out/${TAG}_step_breakdown.txt 2>&1
python tools/trace_timeline.py $F > gpurun_out/${TAG}_step_timeline.txt 2>&1
bash tools/prof_batches.sh $TAG > gpurun_out/batches_$TAG.log 2>&1
bash profiles/run_pmc_step.sh $TAG > gpurun_out/pmc_step_$TAG.log 2>&1
tail -3 gpurun_out/pmc_$TAG.log
