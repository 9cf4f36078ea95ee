#!/bin/bash
# Round 6, fourth GPU session: r06 tests, skinny policy A/Bs on the new 2x2-level shapes, kernel trace of the step.
cd "$GRAFT_REPO_ROOT"; export TMPDIR=/tmp
O=gpurun_out/r06d; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_r06.py -q -m gpu > $O/tests_r06.log 2>&1; echo "r06 tests rc=$?" ; tail -5 $O/tests_r06.log
run() { timeout 300 python bench.py --no-cpu-baseline --no-extras --no-roofline --steps 50 --regions 3 --batch $1 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['ms_per_step'], d['config']['regions_ms_per_step'])"; }
{
for rep in 1 2; do
  echo -n "B=64 default            : "; run 64
  echo -n "B=64 SKINNY_MAXX=131072 : "; AFLDM_SKINNY_MAXX=131072 run 64
  echo -n "B=64 SKINNY_MAXX=262144 : "; AFLDM_SKINNY_MAXX=262144 run 64
  echo -n "B=64 DENSE2_MIN_B=999   : "; AFLDM_DENSE2_MIN_B=999 run 64
done
echo -n "B=32 default          : "; run 32
echo -n "B=32 DENSE2_MIN_B=999 : "; AFLDM_DENSE2_MIN_B=999 run 32
echo -n "B=16 default          : "; run 16
echo -n "B=16 DENSE2_MIN_B=16  : "; AFLDM_DENSE2_MIN_B=16 run 16
} > $O/ab.log 2>&1
cat $O/ab.log
bash profiles/run_profile.sh r06d > $O/prof.log 2>&1
F=$(find gpurun_out/prof_r06d -name "*kernel_trace.csv" | head -1)
python tools/trace_breakdown.py $F 60 > $O/r06d_step_breakdown.txt 2>&1
python tools/trace_timeline.py $F > $O/r06d_step_timeline.txt 2>&1
grep -n "dense2\|skinny" $O/r06d_step_timeline.txt | head -30
