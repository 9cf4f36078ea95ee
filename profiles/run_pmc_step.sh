#!/bin/bash
# HBM traffic (PMC) of EVERY kernel of a denoise step (batch 64, bf16, eager launches so that each dispatch is
# visible to the counters).  Separate passes for FETCH_SIZE and WRITE_SIZE; bytes = (2 * FETCH_SIZE +
# WRITE_SIZE) * 1024 (MI355X_MICROARCH.md, "HBM" / rocprofv3 sections).  Usage: profiles/run_pmc_step.sh <tag>
TAG=${1:-r01}
cd "$GRAFT_REPO_ROOT"; export TMPDIR=/tmp
OUT=gpurun_out/pmc_step_$TAG; rm -rf $OUT; mkdir -p $OUT
CMD="python bench.py --no-graph --steps 4 --warmup 2 --regions 1 --no-cpu-baseline --no-roofline --no-extras"
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $OUT -o fetch -- $CMD > $OUT/fetch.log 2>&1
rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $OUT -o write -- $CMD > $OUT/write.log 2>&1
python - "$OUT" "$TAG" <<'PY'
import csv, glob, json, re, sys
out, tag = sys.argv[1], sys.argv[2]


def fam(n):
    for k in ("k_conv3h", "k_igemm3", "k_igemm2", "k_igemm", "k_lin_wreg", "k_attn_fused", "k_attn", "k_af_act_plane", "k_af_act_kron", "k_af_act_small",
              "k_resample_plane", "k_axis_contract", "k_splitk", "k_gn_apply", "k_gn_partial", "k_conv_cin4", "k_conv_small"):
        if k in n:
            return k
    return "other"


tot = {}
nsteps = 0
for name, ctr in (("fetch", "FETCH_SIZE"), ("write", "WRITE_SIZE")):
    f = glob.glob(out + "/**/" + name + "_counter_collection.csv", recursive=True)[0]
    rows = [r for r in csv.DictReader(open(f)) if r["Counter_Name"] == ctr]
    rows.sort(key=lambda r: int(r["Dispatch_Id"]))
    marks = [i for i, r in enumerate(rows) if ("k_select_step_row" in r["Kernel_Name"] or "k_select_timestep" in r["Kernel_Name"])]
    lo, hi = marks[2], marks[-1]           # steady state: from the third step's first kernel to the last step's
    nsteps = len(marks) - 3
    for r in rows[lo:hi]:
        d = tot.setdefault(fam(r["Kernel_Name"]), dict(fetch=0.0, write=0.0, n=0))
        d[name] += float(r["Counter_Value"])
        if name == "fetch":
            d["n"] += 1
STEPS = float(nsteps)
res = {}
# Round 4 calibration (profiles/r04/fetch_calib.txt): FETCH_SIZE tallies 64 bytes per fabric read request WHATEVER its size - it is
# exact for requests of <= 64 bytes (kernels that read 16 / 32 / 64-byte runs per row) and HALF the truth for 128-byte requests
# (wide coalesced rows, LDS-DMA of 128-byte rows).  read_MB_low = FETCH_SIZE * 1024 (all requests <= 64 B), read_MB_high = 2 x
# that (all requests 128 B); `read_MB_per_step` takes the one that matches how the family reads its operands.
NARROW = ("k_af_act_plane", "k_af_act_kron", "k_af_act_small", "k_axis_contract", "k_resample_plane")      # channel pieces of <= 64 bytes per pixel
for k, d in sorted(tot.items(), key=lambda kv: -(2 * kv[1]["fetch"] + kv[1]["write"])):
    lo = d["fetch"] * 1024 / STEPS / 1e6
    res[k] = dict(launches_per_step=round(d["n"] / STEPS, 1), read_MB_low=round(lo, 1), read_MB_high=round(2 * lo, 1),
                  read_MB_per_step=round(lo if k in NARROW else 2 * lo, 1), reads="<= 64-byte runs" if k in NARROW else "128-byte requests",
                  write_MB_per_step=round(d["write"] * 1024 / STEPS / 1e6, 1))
total = dict(read_MB_per_step=round(sum(v["read_MB_per_step"] for v in res.values()), 1),
             write_MB_per_step=round(sum(v["write_MB_per_step"] for v in res.values()), 1))
doc = dict(tag=tag, note="HBM bytes per denoise step by kernel family from FETCH_SIZE / WRITE_SIZE (separate --pmc passes, eager step); "
                         "FETCH_SIZE counts 64 B per request: x1 for families reading <= 64-byte runs, x2 for 128-byte requests "
                         "(profiles/r04/fetch_calib.txt)",
           total=total, families=res)
print(json.dumps(doc, indent=1))
open("gpurun_out/step_traffic_%s.json" % tag, "w").write(json.dumps(doc, indent=1) + "\n")
PY
