#!/bin/bash
# Same-box A/B of an environment switch in the step: tools/ab_env.sh <VAR> [batches]   (VAR=1 vs unset, alternating)
cd "$GRAFT_REPO_ROOT"
for B in ${2:-64 8 1}; do for f in 1 0 1 0; do echo "B=$B $1=$f"; if [ $f = 1 ]; then export $1=1; else unset $1; fi; timeout 300 python bench.py --no-cpu-baseline --no-extras --no-roofline --steps 50 --regions 3 --batch $B 2>&1 | tail -1 | cut -c150-215; done; done
