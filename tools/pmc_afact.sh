#!/bin/bash
# PMC counters for the fused alias-free activation kernel (run on the GPU box via gpurun)
cd "$GRAFT_REPO_ROOT"; export TMPDIR=/tmp
OUT=gpurun_out/pmc_afact; rm -rf $OUT; mkdir -p $OUT
cat > /tmp/one_afact.py <<'PY'
import sys; sys.path.insert(0, '.')
import torch
from afldm_amd import ops
x = torch.randn(64, 32, 32, 576, device='cuda').to(torch.bfloat16)
g = torch.ones(576, device='cuda'); b = torch.zeros(576, device='cuda')
st = ops.gn_stats(x, 32); y = torch.empty_like(x)
for _ in range(3): ops.af_act(x, None, st, g, b, 32, 1e-5, out=y)
torch.cuda.synchronize()
PY
rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE --output-format csv -d $OUT -o p1 -- python /tmp/one_afact.py > $OUT/log1.txt 2>&1
rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_INSTS_SALU SQ_VALU_MFMA_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_INST_CYCLES_VMEM --output-format csv -d $OUT -o p2 -- python /tmp/one_afact.py > $OUT/log2.txt 2>&1
ls $OUT
python - <<'PY'
import csv, glob, collections
for f in sorted(glob.glob('gpurun_out/pmc_afact/*counter_collection.csv')):
    rows = list(csv.DictReader(open(f)))
    agg = collections.defaultdict(list)
    for r in rows:
        if 'af_act_mfma' in r['Kernel_Name']:
            agg[r['Counter_Name']].append(float(r['Counter_Value']))
    for k, v in agg.items():
        print(f, k, 'n=%d' % len(v), 'last=%.3e' % v[-1])
PY
tail -3 $OUT/log1.txt
