# same-box A/B of the AF-VAE workload (BASELINE configs[3]) across conv3h sub-plane variants
import os, sys, json, subprocess
root = os.environ.get("GRAFT_REPO_ROOT", "/root/repo")
def run(env_extra, tag):
    env = dict(os.environ, **env_extra)
    r = subprocess.run([sys.executable, "bench.py", "--workload", "vae", "--no-cpu-baseline"], capture_output=True, text=True, env=env, cwd=root)
    line = [l for l in r.stdout.splitlines() if l.startswith("{")]
    if not line:
        print(tag, "FAILED", r.stdout[-1500:], r.stderr[-2500:]); return
    d = json.loads(line[-1])
    c = d["config"]
    print(tag, d["value"], "img/s", {k: c[k] for k in c if "ms" in k or "img" in k}, flush=True)
for rep in range(2):
    run({"AFLDM_CONV3H_SUBV": "58"}, "one tile per workgroup (58)")
    run({"AFLDM_CONV3H_SUBV": "63"}, "persistent tiles (63)      ")
