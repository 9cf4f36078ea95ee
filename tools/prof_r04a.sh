cd "$GRAFT_REPO_ROOT"; export TMPDIR=/tmp
bash profiles/run_profile.sh r04a > gpurun_out/prof_r04a.log 2>&1
F=$(find gpurun_out/prof_r04a -name "*kernel_trace.csv" | head -1)
python tools/trace_breakdown.py $F 60 > gpurun_out/r04a_step_breakdown.txt 2>&1
python tools/trace_timeline.py $F > gpurun_out/r04a_step_timeline.txt 2>&1
head -30 gpurun_out/r04a_step_breakdown.txt
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d gpurun_out/fcal -o fc -- python tools/proto/run_fetch_calib.py > gpurun_out/fcal.log 2>&1
python tools/proto/run_fetch_calib.py report gpurun_out/fcal
