#!/bin/bash
# in-situ variant check of the 16^2 short-K GEMMs (to_out 384->384, shortcuts 768 / 576 -> 384, D1 shortcut 192 -> 384) after the round's changes
cd "$GRAFT_REPO_ROOT"; export TMPDIR=/tmp
run() { timeout 300 python bench.py --no-cpu-baseline --no-extras --no-roofline --steps 50 --regions 3 --batch 64 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['ms_per_step'], d['config']['regions_ms_per_step'])"; }
echo -n "default                : "; run
for v in 29 30 33 37 38 39; do
  ov="16384:384:1:384=$v/1;16384:384:1:768=$v/1;16384:384:1:576=$v/1;16384:384:1:192=$v/1"
  echo -n "16^2 1x1 GEMMs on variant $v : "; AFLDM_CONV_OVERRIDE="$ov" run
done
echo -n "default                : "; run
for v in 29 31 37 39; do
  ov="4096:384:1:384=$v/1;4096:384:1:768=$v/1;4096:384:1:1152=$v/1"
  echo -n "8^2 1x1 GEMMs on variant $v : "; AFLDM_CONV_OVERRIDE="$ov" run
done
echo -n "default                : "; run
