# same-box A/B: bench step with / without the fused attention front end + standalone kernel timings
import os, sys, time, json, subprocess
root = os.environ.get("GRAFT_REPO_ROOT", "/root/repo")
def run(env_extra, tag):
    env = dict(os.environ, **env_extra)
    r = subprocess.run([sys.executable, "bench.py", "--steps", "40", "--warmup", "10", "--no-cpu-baseline", "--no-roofline", "--no-extras"],
                       capture_output=True, text=True, env=env, cwd=root)
    line = [l for l in r.stdout.splitlines() if l.startswith("{")]
    if not line:
        print(tag, "FAILED", r.stdout[-2000:], r.stderr[-3000:]); return
    d = json.loads(line[-1])
    print(tag, d["ms_per_step"], d["config"]["regions_ms_per_step"], flush=True)
for rep in range(2):
    run({"AFLDM_NO_FUSED_ATTN": "1"}, "three-launch      ")
    run({"AFLDM_FUSED_ATTN_MIN_T": "1024"}, "fused 32^2        ")
    run({"AFLDM_FUSED_ATTN_MIN_T": "256"}, "fused 32^2 + 16^2 ")
