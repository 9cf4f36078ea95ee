#!/bin/bash
# Round 6, fifth GPU session: dense2 with the XCD-aware order: tests, A/B at batch 64 / 32 / 16, kernel times.
cd "$GRAFT_REPO_ROOT"; export TMPDIR=/tmp
O=gpurun_out/r06e; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_r06.py -q -m gpu > $O/tests_r06.log 2>&1; echo "r06 tests rc=$?" ; tail -5 $O/tests_r06.log
run() { timeout 300 python bench.py --no-cpu-baseline --no-extras --no-roofline --steps 50 --regions 3 --batch $1 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['ms_per_step'], d['config']['regions_ms_per_step'])"; }
{
for rep in 1 2; do
  echo -n "B=64 default (fused)  : "; run 64
  echo -n "B=64 DENSE2_MIN_B=999 : "; AFLDM_DENSE2_MIN_B=999 run 64
  echo -n "B=32 default (fused)  : "; run 32
  echo -n "B=32 DENSE2_MIN_B=999 : "; AFLDM_DENSE2_MIN_B=999 run 32
  echo -n "B=16 DENSE2_MIN_B=16  : "; AFLDM_DENSE2_MIN_B=16 run 16
  echo -n "B=16 default (off)    : "; run 16
done
} > $O/ab.log 2>&1
cat $O/ab.log
bash profiles/run_profile.sh r06e > $O/prof.log 2>&1
F=$(find gpurun_out/prof_r06e -name "*kernel_trace.csv" | head -1)
python tools/trace_timeline.py $F > $O/r06e_step_timeline.txt 2>&1
grep -n "dense2" $O/r06e_step_timeline.txt | head -10
