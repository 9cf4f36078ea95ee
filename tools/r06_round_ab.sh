#!/bin/bash
# Same-box A/B of the round's step changes as a whole: the r06 library with its three switches off (plane-constant 2x2 level,
# non-temporal weight streams, fused conv1 -> norm2 -> activation launch) against the defaults, alternating, batch 64 / 8 / 1.
cd "$GRAFT_REPO_ROOT"; export TMPDIR=/tmp
run() { timeout 300 python bench.py --no-cpu-baseline --no-extras --no-roofline --steps 50 --regions 3 --batch $1 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['ms_per_step'], d['config']['regions_ms_per_step'])"; }
for B in 64 8 1; do
  for rep in 1 2; do
    echo -n "B=$B round-6 step changes ON  : "; run $B
    echo -n "B=$B round-6 step changes OFF : "; AFLDM_NO_CONST2=1 AFLDM_NT_WEIGHTS=0 AFLDM_DENSE2_MIN_B=999 run $B
  done
done
