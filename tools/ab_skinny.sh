#!/bin/bash
# A/B of skinny.hip (operands straight to registers for few-row 1x1 / dense layers): parity, then in-step timings
cd "$GRAFT_REPO_ROOT"
timeout 900 python -m pytest tests/test_gpu_ops.py tests/test_gpu_unet.py -x -q -m gpu 2>&1 | tail -3
for B in 64 32 16 8 1; do for m in 0 1024; do echo "B=$B MAXM=$m"; AFLDM_SKINNY_MAXM=$m timeout 300 python bench.py --no-cpu-baseline --no-extras --no-roofline --steps 50 --regions 3 --batch $B 2>&1 | tail -1 | cut -c100-215; done; done
