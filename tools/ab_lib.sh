#!/bin/bash
# Same-box A/B of two library builds in the step: tools/ab_lib.sh <libA.so> <libB.so> [batches]   (paths relative to the repo)
cd "$GRAFT_REPO_ROOT"
for B in ${3:-64 8 1}; do for l in $1 $2 $1 $2; do echo "B=$B $l"; AFLDM_LIB=$l timeout 300 python bench.py --no-cpu-baseline --no-extras --no-roofline --steps 50 --regions 3 --batch $B 2>&1 | tail -1 | cut -c150-215; done; done
