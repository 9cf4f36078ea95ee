#!/bin/bash
# Per-kernel profile of the bench command (run on the GPU box via gpurun).  Usage: profiles/run_profile.sh <tag>
set -e
TAG=${1:-r01}
cd "$GRAFT_REPO_ROOT"
export TMPDIR=/tmp
OUT=gpurun_out/prof_$TAG
rm -rf $OUT && mkdir -p $OUT
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT -o trace -- python bench.py --steps 10 --warmup 2 --regions 1 --no-cpu-baseline --no-roofline --no-extras > $OUT/bench.log 2>&1 || true
find $OUT -type f | head -20
F=$(find $OUT -name "*kernel_stats.csv" | head -1)
[ -n "$F" ] && cp "$F" gpurun_out/${TAG}_kernel_stats.csv
tail -2 $OUT/bench.log
