#!/bin/bash
# same-box A/B: phase C of k_attn_fused reading the siblings' o rows with sc1 loads (library) vs plain loads (libafldm_plaino.so)
cd "$GRAFT_REPO_ROOT"; export TMPDIR=/tmp
O=gpurun_out/r06g; mkdir -p $O
run() { timeout 300 python bench.py --no-cpu-baseline --no-extras --no-roofline --steps 50 --regions 3 --batch 64 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['ms_per_step'], d['config']['regions_ms_per_step'])"; }
{
for rep in 1 2 3; do
  echo -n "sc1 loads   : "; run
  echo -n "plain loads : "; AFLDM_LIB=afldm_amd/lib/libafldm_plaino.so run
done
SHAPES=1024x192x8 GRAPH=1 timeout 300 python tools/bench_attnfo.py 2>&1 | tail -8
echo "--- plain"
AFLDM_LIB=afldm_amd/lib/libafldm_plaino.so SHAPES=1024x192x8 GRAPH=1 timeout 300 python tools/bench_attnfo.py 2>&1 | tail -8
} > $O/ab.log 2>&1
cat $O/ab.log
