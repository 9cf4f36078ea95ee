"""List the GroupNorm-statistics passes that still run as stand-alone kernels in one denoise step
(every other consumer takes the partial sums its producer emitted).  GPU box only."""
import collections
import os
import sys
import traceback

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from afldm_amd import ops  # noqa: E402
import bench  # noqa: E402

cnt = collections.Counter()
_orig = ops._tensor_stats


def spy(x, out=None):
    if getattr(x, "gn_partial", None) is None or out is not None:
        fr = traceback.extract_stack(limit=7)[:-1]
        cnt[(tuple(x.shape), " <- ".join(f"{os.path.basename(f.filename)}:{f.name}:{f.lineno}" for f in fr[-4:]))] += 1
    return _orig(x, out)


ops._tensor_stats = spy
from afldm_amd.engine import DenoiseEngine  # noqa: E402
from afldm_amd.schedulers.ddim import ffhq_ddim_scheduler  # noqa: E402

unet = bench.build_unet(torch.bfloat16, "cuda")
B = int(os.environ.get("B", 64))
eng = DenoiseEngine(unet, ffhq_ddim_scheduler(), B, 50, use_graph=False)
eng.reset(torch.randn(B, 4, 32, 32, generator=torch.Generator().manual_seed(1)))
eng.step(1)
cnt.clear()
eng.step(1)
for (shape, where), n in sorted(cnt.items(), key=lambda kv: -kv[1]):
    print(n, shape, where)
