# in-step A/B of the merged launches: ms/step of the batch-64 bf16 step, graph replay
for cfg in "none:AFLDM_NO_ACTCONV=1" "n16:AFLDM_ACTCONV_N=16" "n16_32:AFLDM_ACTCONV_N=16,32" "none2:AFLDM_NO_ACTCONV=1" "n16b:AFLDM_ACTCONV_N=16"; do
  name=${cfg%%:*}; envs=${cfg#*:}
  env $envs python bench.py --steps 30 --warmup 5 --regions 3 --no-cpu-baseline --no-roofline --no-extras > gpurun_out/r05c/step_$name.json 2>/dev/null
  python - <<PY
import json
d=json.loads(open("gpurun_out/r05c/step_$name.json").read().strip().splitlines()[-1])
print("$name", d["ms_per_step"], d["config"].get("regions_ms_per_step"), d["config"].get("kernels_per_step"))
PY
done
