#!/bin/bash
# Per-kernel profile of the AF-VAE workload (BASELINE configs[3]: encode + decode 256^2 x 128, bf16).
# Usage (on the GPU box): profiles/run_vae_profile.sh <tag>  ->  gpurun_out/<tag>_vae_kernel_stats.csv
TAG=${1:-r03}
cd "$GRAFT_REPO_ROOT"; export TMPDIR=/tmp
OUT=gpurun_out/prof_vae_$TAG; rm -rf $OUT; mkdir -p $OUT
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT -o trace -- python bench.py --workload vae > $OUT/bench.log 2>&1 || true
F=$(find $OUT -name "*kernel_stats.csv" | head -1)
[ -n "$F" ] && cp "$F" gpurun_out/${TAG}_vae_kernel_stats.csv && head -25 "$F"
tail -1 $OUT/bench.log
