#!/bin/bash
# in-step A/B of environment switches: ms/step of the bench step (graph replay).  usage: ab_step_env.sh OUTDIR BATCH name:ENV=1 name2:ENV2=x ...
out=$1; shift; batch=$1; shift
mkdir -p $out
for cfg in "$@"; do
  name=${cfg%%:*}; envs=${cfg#*:}
  env $envs python bench.py --batch $batch --steps 30 --warmup 5 --regions 3 --no-cpu-baseline --no-roofline --no-extras > $out/step_${name}_b$batch.json 2>/dev/null
  python - <<PY
import json
d=json.loads(open("$out/step_${name}_b$batch.json").read().strip().splitlines()[-1])
print("batch $batch $name", d["ms_per_step"], d["config"].get("regions_ms_per_step"))
PY
done
