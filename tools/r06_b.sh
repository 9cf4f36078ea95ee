#!/bin/bash
# Round 6, second GPU session: r06 tests, nt-weights / fused split-K A/Bs at small batches, harness timing with finer laps.
cd "$GRAFT_REPO_ROOT"; export TMPDIR=/tmp
O=gpurun_out/r06b; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_r06.py -q -m gpu > $O/tests_r06.log 2>&1; echo "r06 tests rc=$?" ; tail -5 $O/tests_r06.log
run() { timeout 300 python bench.py --no-cpu-baseline --no-extras --no-roofline --steps 50 --regions 3 --batch $1 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['ms_per_step'], d['config']['regions_ms_per_step'])"; }
{
for B in 1 8; do
  for rep in 1 2; do
    echo -n "B=$B nt on : "; run $B
    echo -n "B=$B nt off: "; AFLDM_NT_WEIGHTS=0 run $B
  done
  echo -n "B=$B fused splitk: "; AFLDM_FUSED_SPLITK=1 run $B
  echo -n "B=$B default     : "; run $B
done
echo -n "B=64 nt on : "; run 64
echo -n "B=64 nt off: "; AFLDM_NT_WEIGHTS=0 run 64
} > $O/ab.log 2>&1
cat $O/ab.log
timeout 600 python bench.py --workload harness > $O/bench_harness.json 2> $O/bench_harness.err; tail -c 1800 $O/bench_harness.json; tail -3 $O/bench_harness.err
