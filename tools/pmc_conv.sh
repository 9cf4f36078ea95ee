#!/bin/bash
# PMC counters for the implicit-GEMM conv kernel on the FFHQ level-32 shape (run on the GPU box)
cd "$GRAFT_REPO_ROOT"; export TMPDIR=/tmp
OUT=gpurun_out/pmc_conv; rm -rf $OUT; mkdir -p $OUT
cat > /tmp/one_conv.py <<'PY'
import sys; sys.path.insert(0, '.')
import torch
from afldm_amd import ops
B, H, C1, Cout = 64, 32, 576, 192
x = torch.randn(B, H, H, C1, device='cuda').to(torch.bfloat16)
w = (torch.randn(Cout, 3, 3, C1, device='cuda') / 72).to(torch.bfloat16)
bias = torch.zeros(Cout, device='cuda'); y = torch.empty(B, H, H, Cout, device='cuda', dtype=torch.bfloat16)
for _ in range(3): ops.conv2d(x, w, bias, out=y)
torch.cuda.synchronize()
PY
rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE --output-format csv -d $OUT -o p1 -- python /tmp/one_conv.py > $OUT/log1.txt 2>&1
rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_INSTS_SALU SQ_VALU_MFMA_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_INSTS_VMEM --output-format csv -d $OUT -o p2 -- python /tmp/one_conv.py > $OUT/log2.txt 2>&1
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $OUT -o p3 -- python /tmp/one_conv.py > $OUT/log3.txt 2>&1
rocprofv3 --kernel-trace --pmc WRITE_SIZE TCC_HIT_sum TCC_MISS_sum --output-format csv -d $OUT -o p4 -- python /tmp/one_conv.py > $OUT/log4.txt 2>&1
python - <<'PY'
import csv, glob, collections
for f in sorted(glob.glob('gpurun_out/pmc_conv/*counter_collection.csv')):
    rows = list(csv.DictReader(open(f)))
    agg = collections.defaultdict(list)
    for r in rows:
        if 'igemm' in r['Kernel_Name']:
            agg[r['Counter_Name']].append(float(r['Counter_Value']))
    for k, v in agg.items():
        print(f.split('/')[-1], k, 'n=%d' % len(v), 'last=%.4e' % v[-1])
rows = list(csv.DictReader(open('gpurun_out/pmc_conv/p1_kernel_trace.csv')))
for r in rows:
    if 'igemm' in r['Kernel_Name']:
        print('dur us', (int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e3, r['Kernel_Name'][:60], 'vgpr', r['VGPR_Count'], 'agpr', r['Accum_VGPR_Count'], 'lds', r['LDS_Block_Size'])
PY
