#!/bin/bash
# Kernel-trace breakdown of one denoise step at batch 1 and 8 (run on the GPU box via gpurun).  Usage: tools/prof_batches.sh <tag>
TAG=${1:-r02}
cd "$GRAFT_REPO_ROOT"; export TMPDIR=/tmp
for B in 1 8; do
OUT=gpurun_out/prof_${TAG}_b$B; rm -rf $OUT; mkdir -p $OUT
rocprofv3 --kernel-trace --output-format csv -d $OUT -o trace -- python bench.py --batch $B --steps 10 --warmup 2 --regions 1 --no-cpu-baseline --no-roofline --no-extras > $OUT/bench.log 2>&1
F=$(find $OUT -name "*kernel_trace.csv" | head -1)
python tools/trace_breakdown.py $F 45 > gpurun_out/${TAG}_b${B}_breakdown.txt 2>&1
python tools/trace_timeline.py $F > gpurun_out/${TAG}_b${B}_timeline.txt 2>&1
done
