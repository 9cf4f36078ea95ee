# generic same-box in-step A/B: python tools/ab_env.py "label=ENV=V,ENV2=V" "label2=" ...
import os, sys, json, subprocess
root = os.environ.get("GRAFT_REPO_ROOT", "/root/repo")
def run(env_extra, tag):
    env = dict(os.environ, **env_extra)
    extra = os.environ.get("AB_ARGS", "").split()          # e.g. AB_ARGS="--batch 8"
    r = subprocess.run([sys.executable, "bench.py", "--steps", "40", "--warmup", "10", "--no-cpu-baseline", "--no-roofline", "--no-extras"] + extra,
                       capture_output=True, text=True, env=env, cwd=root)
    line = [l for l in r.stdout.splitlines() if l.startswith("{")]
    if not line:
        print(tag, "FAILED", r.stdout[-1500:], r.stderr[-2500:]); return
    d = json.loads(line[-1])
    print(f"{tag:28s}", d["ms_per_step"], d["config"]["regions_ms_per_step"], flush=True)
cfgs = []
for a in sys.argv[1:]:
    label, _, envs = a.partition("=")
    cfgs.append((label, dict(kv.split("=", 1) for kv in envs.split(",") if kv)))
for rep in range(2):
    for label, env in cfgs:
        run(env, label)
