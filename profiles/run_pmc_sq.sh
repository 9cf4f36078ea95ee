#!/bin/bash
# SQ counters (MFMA busy, wait states, instruction mix) of EVERY kernel of a denoise step (batch 64, bf16, eager launches so
# that each dispatch is visible), aggregated per kernel family.  Two separate --pmc passes (8 SQ slots each,
# MI355X_MICROARCH.md "rocprofv3 PMC slots"), --kernel-trace only (no other trace domains).  MFMA utilisation of a family =
# SQ_VALU_MFMA_BUSY_CYCLES / (1024 SIMDs x cycles), cycles = GRBM_GUI_ACTIVE / 8: rocprofv3 reports GRBM_GUI_ACTIVE summed over
# the 8 XCDs (877 038 for a 43.9 us dispatch = 8 x 2.5 GHz), the SQ counters summed over all SIMDs.
# Usage (on the GPU box, via gpurun): profiles/run_pmc_sq.sh <tag>   ->  gpurun_out/pmc_sq_<tag>.{json,txt}
TAG=${1:-r03}
cd "$GRAFT_REPO_ROOT"; export TMPDIR=/tmp
OUT=gpurun_out/pmc_sq_$TAG; rm -rf $OUT; mkdir -p $OUT
CMD="python bench.py --no-graph --steps 4 --warmup 2 --regions 1 --no-cpu-baseline --no-roofline --no-extras"
rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE --output-format csv -d $OUT -o a -- $CMD > $OUT/a.log 2>&1
rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_INSTS_VMEM SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE --output-format csv -d $OUT -o b -- $CMD > $OUT/b.log 2>&1
python - "$OUT" "$TAG" <<'PY'
import csv, glob, json, sys
out, tag = sys.argv[1], sys.argv[2]


def fam(n):
    for k in ("k_conv3h", "k_igemm3", "k_igemm2", "k_igemm", "k_skinny", "k_dense2_gn_act", "k_lin_wreg", "k_attn_fused", "k_attn", "k_af_act_plane", "k_af_act_kron",
              "k_af_act_small", "k_af_act_slabs", "k_resample_plane", "k_axis_contract", "k_splitk", "k_gn_apply", "k_gn_partial",
              "k_conv_cin4", "k_conv_out_fused"):
        if k in n:
            return k
    return "other"


# conv3h by plane size (template args <T, BM, W_, ...>): the 32^2 / 16^2 launches separately from 8^2 / 4^2
def fam2(n):
    f = fam(n)
    if f == "k_conv3h":
        import re
        m = re.search(r"k_conv3h(?:I\w+?b|<[^,]+,)\s*(?:Li)?(\d+)E?,?\s*(?:Li)?(\d+)", n)      # mangled (IDF16bLi256ELi32E...) or demangled
        return f"k_conv3h_W{m.group(2)}" if m else f
    return f


tot = {}
nsteps = 0
for name in ("a", "b"):
    fs = glob.glob(out + "/**/" + name + "_counter_collection.csv", recursive=True)
    if not fs:
        continue
    rows = list(csv.DictReader(open(fs[0])))
    ids = sorted({int(r["Dispatch_Id"]) for r in rows})
    first = {}
    for r in rows:
        first.setdefault(int(r["Dispatch_Id"]), r["Kernel_Name"])
    marks = [i for i in ids if "k_select_step_row" in first[i] or "k_select_timestep" in first[i]]
    lo, hi = marks[2], marks[-1]
    nsteps = len(marks) - 3
    for r in rows:
        d = int(r["Dispatch_Id"])
        if d < lo or d >= hi:
            continue
        t = tot.setdefault(fam2(r["Kernel_Name"]), {})
        t[r["Counter_Name"]] = t.get(r["Counter_Name"], 0.0) + float(r["Counter_Value"])
        if name == "a" and r["Counter_Name"] == "GRBM_GUI_ACTIVE":
            t["_n"] = t.get("_n", 0) + 1
res = {}
for k, t in tot.items():
    g = t.get("GRBM_GUI_ACTIVE", 0.0)
    wc = t.get("SQ_WAVE_CYCLES", 0.0)
    d = dict(launches_per_step=round(t.get("_n", 0) / nsteps, 1), gui_active_cycles_per_step=round(g / 8.0 / nsteps))
    if g > 0:
        d["mfma_busy_frac"] = round(t.get("SQ_VALU_MFMA_BUSY_CYCLES", 0.0) / (1024.0 * g / 8.0), 4)
    if wc > 0:
        d["wait_any_frac_of_wave_cycles"] = round(t.get("SQ_WAIT_ANY", 0.0) / wc, 4)
        d["wait_inst_any_frac"] = round(t.get("SQ_WAIT_INST_ANY", 0.0) / wc, 4)
        d["active_inst_any_frac"] = round(t.get("SQ_ACTIVE_INST_ANY", 0.0) / wc, 4)
        d["wait_inst_lds_frac"] = round(t.get("SQ_WAIT_INST_LDS", 0.0) / wc, 4)
    for c in ("SQ_INSTS_VALU", "SQ_INSTS_MFMA", "SQ_INSTS_LDS", "SQ_INSTS_VMEM", "SQ_LDS_BANK_CONFLICT", "SQ_LDS_IDX_ACTIVE"):
        if c in t:
            d[c + "_per_step"] = round(t[c] / nsteps)
    res[k] = d
res = dict(sorted(res.items(), key=lambda kv: -kv[1]["gui_active_cycles_per_step"]))
doc = dict(tag=tag, steps_averaged=nsteps,
           note="eager denoise step, batch 64 bf16; SQ counters summed over the family's dispatches of the steady-state steps; "
                "mfma_busy_frac = SQ_VALU_MFMA_BUSY_CYCLES / (1024 SIMDs x GRBM_GUI_ACTIVE / 8 XCDs); wait/active fractions are of SQ_WAVE_CYCLES",
           families=res)
open(f"gpurun_out/pmc_sq_{tag}.json", "w").write(json.dumps(doc, indent=1) + "\n")
with open(f"gpurun_out/pmc_sq_{tag}.txt", "w") as f:
    f.write(f"{'family':22s} {'n/step':>6s} {'Mcyc/step':>9s} {'MFMAbusy':>8s} {'waitAny':>8s} {'waitInst':>8s} {'active':>7s} {'waitLDS':>8s}\n")
    for k, d in res.items():
        f.write(f"{k:22s} {d['launches_per_step']:6.1f} {d['gui_active_cycles_per_step'] / 1e6:9.3f} {d.get('mfma_busy_frac', 0):8.3f} "
                f"{d.get('wait_any_frac_of_wave_cycles', 0):8.3f} {d.get('wait_inst_any_frac', 0):8.3f} {d.get('active_inst_any_frac', 0):7.3f} "
                f"{d.get('wait_inst_lds_frac', 0):8.3f}\n")
print(open(f"gpurun_out/pmc_sq_{tag}.txt").read())
PY
