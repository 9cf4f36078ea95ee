#!/bin/bash
# kernel-only durations (rocprofv3) of tools/probe_skinny.py for the libraries / settings given: prof_skinny.sh "<lib stems>"
cd "$GRAFT_REPO_ROOT"; export TMPDIR=/tmp
for l in $1; do
  OUT=gpurun_out/prof_skp; rm -rf $OUT; mkdir -p $OUT
  AFLDM_LIB=afldm_amd/lib/$l.so rocprofv3 --kernel-trace --output-format csv -d $OUT -o t -- python tools/probe_skinny.py > $OUT/log.txt 2>&1
  F=$(find $OUT -name "*kernel_trace.csv" | head -1)
  echo "## $l (AFLDM_SKINNY_XT=$AFLDM_SKINNY_XT)"
  python - $F <<'PY'
import csv, sys
rows = [r for r in csv.DictReader(open(sys.argv[1]))]
seq = [(r["Kernel_Name"], (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3) for r in rows if "afldm" in r["Kernel_Name"]]
# one probe configuration = 53 calls; group consecutive identical kernel sequences
per = {}
i = 0
calls = [s for s in seq if ("skinny" in s[0] or "igemm" in s[0] or "splitk" in s[0])]
names = []
for n, d in calls:
    key = n.split("(")[0][:60]
    per.setdefault(key, []).append(d)
for k, v in per.items():
    n = len(v) // 4 if len(v) >= 4 else 1
    meds = []
    for c in range(0, len(v), max(n, 1)):
        seg = sorted(v[c:c + n]); meds.append(seg[len(seg) // 2])
    print("   %-60s %s" % (k, " ".join("%6.1f" % m for m in meds[:4])))
PY
done
