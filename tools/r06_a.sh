#!/bin/bash
# Round 6, first GPU session: tests of the round's changes, then same-box A/Bs (ms/step of the bench step, graph replay).
cd "$GRAFT_REPO_ROOT"; export TMPDIR=/tmp
O=gpurun_out/r06a; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_r06.py -x -q -m gpu > $O/tests_r06.log 2>&1; echo "r06 tests rc=$?" ; tail -3 $O/tests_r06.log
timeout 1500 python -m pytest tests -x -q -m gpu --deselect tests/test_gpu_r06.py > $O/tests_all.log 2>&1; echo "all tests rc=$?"; tail -3 $O/tests_all.log
run() { timeout 300 python bench.py --no-cpu-baseline --no-extras --no-roofline --steps 50 --regions 3 --batch $1 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['ms_per_step'], d['config']['regions_ms_per_step'])"; }
{
for B in 64 1 8; do
  for rep in 1 2; do
    echo -n "B=$B const2 on : "; run $B
    echo -n "B=$B const2 off: "; AFLDM_NO_CONST2=1 run $B
  done
done
for rep in 1 2; do
  echo -n "B=64 sites none     : "; run 64
  echo -n "B=64 sites 16:576   : "; AFLDM_ACTCONV_SITES=16:576 run 64
  echo -n "B=64 sites 16:576,16:768 : "; AFLDM_ACTCONV_SITES=16:576,16:768 run 64
done
} > $O/ab.log 2>&1
cat $O/ab.log
bash profiles/run_profile.sh r06a > $O/prof.log 2>&1
F=$(find gpurun_out/prof_r06a -name "*kernel_trace.csv" | head -1)
python tools/trace_breakdown.py $F 60 > $O/r06a_step_breakdown.txt 2>&1
python tools/trace_timeline.py $F > $O/r06a_step_timeline.txt 2>&1
head -30 $O/r06a_step_breakdown.txt
timeout 600 python bench.py --workload harness > $O/bench_harness.json 2> $O/bench_harness.err; tail -c 1500 $O/bench_harness.json; tail -3 $O/bench_harness.err
