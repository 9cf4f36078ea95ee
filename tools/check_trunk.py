import sys, time, torch
import os; R = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, R); sys.path.insert(0, os.path.join(R, "tests"))
from test_gpu_r02 import build_unet, rel_rms
from afldm_amd import trunk, ops
unet, _, _ = build_unet("ffhq", torch.bfloat16)
for B in (2, 64, 1, 8):
    x = torch.randn(B, 4, 32, 32, generator=torch.Generator().manual_seed(B)).cuda()
    trunk._ENABLED = False
    y0 = unet(x, 501).sample
    trunk._ENABLED = True
    y1 = unet(x, 501).sample
    y2 = unet(x, 501).sample
    torch.cuda.synchronize()
    print(B, "trunk vs separate launches rel-RMS", rel_rms(y1.float(), y0.float().cpu()), "rerun identical", torch.equal(y1, y2), "err", ops.actconv_error(), flush=True)
