import json, sys
d = json.load(open(sys.argv[1]))
print(d["value"], d["ms_per_step"], d["config"]["regions_ms_per_step"])
print("box", d["box"])
print("rccl", d.get("rccl"))
print("roofline", {k: d["roofline"].get(k) for k in ("kernel", "achieved", "frac", "traffic", "family_ms_per_step", "traffic_note")})
for k, v in d["kernel_families"].items():
    print(" ", k, v["launches_per_step"], v["ms_per_step"], v.get("tflops"), v.get("gbs"))
print("other", [(o.get("batch"), o.get("dtype"), o.get("ms_per_step")) for o in d["other_configs"]])
print("vae", d["vae_c4"]["value"])
for c in d.get("concurrent_jobs", []):
    print("concurrent", c["jobs"], "x", c["batch_per_job"], ":", c["ms_per_64_samples"], "ms per 64 samples,", c["value"], "steps/s")
print("cpu", d["cpu_baseline"])
