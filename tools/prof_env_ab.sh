#!/bin/bash
# per-kernel breakdown of one step under an environment setting: tools/prof_env_ab.sh <name> [ENV=V ...]
cd "$GRAFT_REPO_ROOT"; export TMPDIR=/tmp
name=$1; shift
OUT=gpurun_out/prof_$name; rm -rf $OUT; mkdir -p $OUT
env "$@" rocprofv3 --kernel-trace --stats --output-format csv -d $OUT -o trace -- python bench.py --steps 10 --warmup 2 --regions 1 --no-cpu-baseline --no-roofline --no-extras > $OUT/bench.log 2>&1
F=$(find $OUT -name "*kernel_trace.csv" | head -1)
python tools/trace_breakdown.py $F 60 > gpurun_out/${name}_breakdown.txt 2>&1
python tools/trace_timeline.py $F > gpurun_out/${name}_timeline.txt 2>&1
