#!/bin/bash
# Round 6, third GPU session: the fused conv1 -> norm2 -> activation launch of the 2x2 level (csrc/dense2.hip): tests, A/B, harness.
cd "$GRAFT_REPO_ROOT"; export TMPDIR=/tmp
O=gpurun_out/r06c; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_r06.py -q -m gpu > $O/tests_r06.log 2>&1; echo "r06 tests rc=$?" ; tail -5 $O/tests_r06.log
run() { timeout 300 python bench.py --no-cpu-baseline --no-extras --no-roofline --steps 50 --regions 3 --batch $1 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['ms_per_step'], d['config']['regions_ms_per_step'])"; }
{
for B in 64 1 8; do
  for rep in 1 2; do
    echo -n "B=$B dense2 fused on : "; run $B
    echo -n "B=$B dense2 fused off: "; AFLDM_NO_DENSE2_FUSED=1 run $B
  done
done
} > $O/ab.log 2>&1
cat $O/ab.log
timeout 600 python bench.py --workload harness > $O/bench_harness.json 2> $O/bench_harness.err; tail -c 1500 $O/bench_harness.json; tail -3 $O/bench_harness.err
timeout 1500 python -m pytest tests -x -q -m gpu --deselect tests/test_gpu_r06.py > $O/tests_all.log 2>&1; echo "all tests rc=$?"; tail -3 $O/tests_all.log
