#!/bin/bash
# End-of-round evidence run (on the GPU box via gpurun): PMC traffic of the conv3x3 family, per-kernel profile + step
# breakdown / timeline at batch 64, breakdowns at batch 1 / 8, whole-step PMC traffic, the default bench line.
# Usage: tools/final_prof.sh <tag>     (results under gpurun_out/; copy what is to be judged into profiles/)
TAG=${1:-r02}
cd "$GRAFT_REPO_ROOT"; export TMPDIR=/tmp
bash profiles/run_pmc_conv3x3.sh $TAG > gpurun_out/pmc_$TAG.log 2>&1
cp gpurun_out/conv3x3_traffic_$TAG.json profiles/conv3x3_traffic.json      # so that the bench below stamps the traffic of THIS build
bash profiles/run_profile.sh $TAG > gpurun_out/prof_$TAG.log 2>&1
F=$(find gpurun_out/prof_$TAG -name "*kernel_trace.csv" | head -1)
python tools/trace_breakdown.py $F 60 > gpurun_out/${TAG}_step_breakdown.txt 2>&1
python tools/trace_timeline.py $F > gpurun_out/${TAG}_step_timeline.txt 2>&1
bash tools/prof_batches.sh $TAG > gpurun_out/batches_$TAG.log 2>&1
bash profiles/run_pmc_step.sh $TAG > gpurun_out/pmc_step_$TAG.log 2>&1
bash profiles/run_pmc_sq.sh $TAG > gpurun_out/pmc_sq_$TAG.log 2>&1
bash profiles/run_vae_profile.sh $TAG > gpurun_out/prof_vae_$TAG.log 2>&1
python bench.py > gpurun_out/bench_$TAG.json 2> gpurun_out/bench_$TAG.err
python bench.py --workload vae --no-cpu-baseline > gpurun_out/bench_vae_$TAG.json 2> gpurun_out/bench_vae_$TAG.err
python bench.py --workload harness > gpurun_out/bench_harness_$TAG.json 2> gpurun_out/bench_harness_$TAG.err
timeout 1500 python -m pytest tests -q -m gpu > gpurun_out/gputests_$TAG.log 2>&1; tail -3 gpurun_out/gputests_$TAG.log
tail -c 600 gpurun_out/pmc_$TAG.log; tail -c 300 gpurun_out/bench_$TAG.json
