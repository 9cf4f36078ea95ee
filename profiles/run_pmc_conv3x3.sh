#!/bin/bash
# HBM traffic (PMC) of the bench's dominant kernel family: the 64 conv3x3 launches of one denoise
# step (batch 64, bf16), replayed alone.  Separate passes for FETCH_SIZE and WRITE_SIZE (TCC slot
# limit, MI355X_MICROARCH.md "rocprofv3 PMC slots"); bytes = (2 * FETCH_SIZE + WRITE_SIZE) * 1024
# (gfx950: FETCH_SIZE counts 64 B per 128-B request of a wide coalesced read, "HBM" section).
# Usage (on the GPU box, via gpurun): profiles/run_pmc_conv3x3.sh <tag>
TAG=${1:-r01}
cd "$GRAFT_REPO_ROOT"; export TMPDIR=/tmp
OUT=gpurun_out/pmc_conv3x3_$TAG; rm -rf $OUT; mkdir -p $OUT
python tools/replay_conv3x3.py record > $OUT/record.json 2> $OUT/record.log
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $OUT -o fetch -- python tools/replay_conv3x3.py > $OUT/fetch.log 2>&1
rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $OUT -o write -- python tools/replay_conv3x3.py > $OUT/write.log 2>&1
python - "$OUT" "$TAG" <<'PY'
import csv, glob, json, sys
out, tag = sys.argv[1], sys.argv[2]
rec = json.loads(open(out + "/record.json").read().strip().splitlines()[-1])
tot = {}
n = {}
for name in ("fetch", "write"):
    f = glob.glob(out + "/**/" + name + "_counter_collection.csv", recursive=True)[0]
    s, k = 0.0, 0
    for r in csv.DictReader(open(f)):
        kn = r["Kernel_Name"]
        if ("k_igemm" in kn or "k_splitk" in kn or "k_conv_" in kn or "k_conv3h" in kn) and r["Counter_Name"] in ("FETCH_SIZE", "WRITE_SIZE"):
            s += float(r["Counter_Value"]); k += 1
    tot[name], n[name] = s, k
reps = 3
launches = rec["launches"]
fetch_b = 2.0 * tot["fetch"] * 1024 / reps
write_b = tot["write"] * 1024 / reps
import hashlib
sha = hashlib.sha256(open("afldm_amd/lib/libafldm_hip.so", "rb").read()).hexdigest()
res = dict(tag=tag, family="conv3x3", lib_sha256=sha, launches_per_step=launches, kernel_dispatches_counted=n["fetch"] // reps,
           fetch_bytes_per_step=fetch_b, write_bytes_per_step=write_b,
           hbm_bytes_per_launch=(fetch_b + write_b) / launches,
           algorithmic_bytes_per_launch=rec["algorithmic_bytes_per_step"] / launches,
           note="bytes = (2*FETCH_SIZE + WRITE_SIZE)*1024, separate --pmc passes, 3 replays of the conv3x3 launches of one step averaged; lib_sha256 = the library measured (bench.py refuses another build)")
print(json.dumps(res))
open("gpurun_out/conv3x3_traffic_%s.json" % tag, "w").write(json.dumps(res, indent=1) + "\n")
PY
