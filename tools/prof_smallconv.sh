#!/bin/bash
# kernel-only durations (rocprofv3) of tools/bench_smallconv.py for a list of FORCE settings: prof_smallconv.sh "auto 32/8 32/16" "L2d 3072"
cd "$GRAFT_REPO_ROOT"; export TMPDIR=/tmp
for f in $1; do
  OUT=gpurun_out/prof_small_$(echo $f | tr '/' '_'); rm -rf $OUT; mkdir -p $OUT
  if [ "$f" = "auto" ]; then unset FORCE; else export FORCE=$f; fi
  ONLY="$2" ITERS=20 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT -o t -- python tools/bench_smallconv.py > $OUT/log.txt 2>&1
  echo "== $f"
  python - "$OUT" <<'PY'
import csv, glob, sys
f = glob.glob(sys.argv[1] + "/**/*kernel_stats.csv", recursive=True)[0]
for r in csv.DictReader(open(f)):
    n = r["Name"]
    if "afldm" in n and ("igemm" in n or "splitk" in n or "conv3h" in n):
        print(f"   {n[:110]:110s} calls {r['Calls']:>5s} avg {float(r['AverageNs'])/1e3:7.1f} us")
PY
done
