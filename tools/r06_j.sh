#!/bin/bash
# Round 6: the I2SB ODE bridge / SR harness on replayed graphs: tests + timing, and the batch-64 forward's distance to the oracle.
cd "$GRAFT_REPO_ROOT"; export TMPDIR=/tmp
O=gpurun_out/r06j; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_r06.py -q -m gpu -k "i2sb or sr" > $O/tests_r06.log 2>&1; echo "r06 i2sb tests rc=$?"; tail -5 $O/tests_r06.log
timeout 1500 python -m pytest tests -q -m gpu -k "i2sb or bridge or sr or bench_path or full_50" -s 2>&1 | grep -E "passed|failed|\[C|rel-RMS|bench path|\[equiv" | tail -20
timeout 600 python - <<'PY' 2>&1 | tail -3
import json, sys
sys.path.insert(0, ".")
import bench
import torch
torch.cuda.set_device(0)
print(json.dumps(bench.i2sb_c5()))
PY
