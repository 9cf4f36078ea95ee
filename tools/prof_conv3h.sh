#!/bin/bash
# kernel-only durations (rocprofv3) of tools/bench_conv3h.py under a list of AFLDM_CONV_DBG values: prof_conv3h.sh "0 3 7 11 27" "L32 192" 41
cd "$GRAFT_REPO_ROOT"; export TMPDIR=/tmp
for d in $1; do
  OUT=gpurun_out/prof_c3h_$d; rm -rf $OUT; mkdir -p $OUT
  AFLDM_CONV_DBG=$d SHAPES="$2" VARIANTS=$3 ROUNDS=1 ITERS=40 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT -o t -- python tools/bench_conv3h.py > $OUT/log.txt 2>&1
  python - "$OUT" "$d" <<'PY'
import csv, glob, sys
f = glob.glob(sys.argv[1] + "/**/*kernel_stats.csv", recursive=True)[0]
for r in csv.DictReader(open(f)):
    n = r["Name"]
    if "conv3h" in n or "igemm" in n:
        print(f"dbg {sys.argv[2]:>3s}  {n[20:80]:60s} avg {float(r['AverageNs'])/1e3:7.1f} us  min {float(r['MinNs'])/1e3:7.1f}")
PY
done
