#!/usr/bin/env python
"""AF-VAE at the reference's full size (configs/vae/model_afvae.json topology, seeded weights):
encode + decode throughput at 256x256 and the fractional-shift equivariance PSNR of the decoder
(BASELINE.json configs[3]).  python tools/bench_vae.py [batch]"""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

from afldm_amd.models.af_vae import AliasFreeAutoencoderKL  # noqa: E402
from afldm_amd.shift_utils.metrics import mask_psnr  # noqa: E402
from afldm_amd.shift_utils.shifters import ImageShifter  # noqa: E402

B = int(sys.argv[1]) if len(sys.argv) > 1 else 16
torch.manual_seed(0)
vae = AliasFreeAutoencoderKL(in_channels=3, out_channels=3, down_block_types=["DownEncoderBlock2D"] * 4,
                             up_block_types=["UpDecoderBlock2D"] * 4, block_out_channels=[128, 256, 512, 512],
                             layers_per_block=2, latent_channels=4, scaling_factor=0.6,
                             down_filtered_act=[False, True, True, True], up_filtered_act=[True, True, True, False],
                             up_rescale=[True, True, True]).cuda().to(torch.bfloat16)
x = (torch.rand(B, 3, 256, 256, generator=torch.Generator().manual_seed(7)) * 2 - 1).cuda()
for name, fn in (("encode", lambda: vae.encode(x).latent_dist.mode()), ):
    fn(); torch.cuda.synchronize()
    t0 = time.perf_counter(); z = fn(); torch.cuda.synchronize()
    print(f"{name}: B={B} {1e3*(time.perf_counter()-t0):.1f} ms  ({B/(time.perf_counter()-t0):.1f} img/s)  finite={bool(torch.isfinite(z).all())}")
z = z.float() * 0.6
vae.decode_scale(z); torch.cuda.synchronize()
t0 = time.perf_counter(); img = vae.decode_scale(z); torch.cuda.synchronize()
print(f"decode: B={B} {1e3*(time.perf_counter()-t0):.1f} ms  ({B/(time.perf_counter()-t0):.1f} img/s)  finite={bool(torch.isfinite(img).all())}")
for tj in (0.125, 0.5, 1.0):
    zs, m = ImageShifter("ideal_crop", 8).shift(z[:2].contiguous(), 0, tj)
    img_s = vae.decode_scale(zs)
    ref, m_img = ImageShifter("ideal_crop", 1).shift(img[:2].float().contiguous(), 0, 8 * tj) if float(8 * tj).is_integer() else (None, None)
    if ref is not None:
        mask = torch.ones_like(ref); mask[..., :32] = 0; mask[..., -32:] = 0
        print(f"decoder shift-equivariance tj={tj}: masked PSNR {float(mask_psnr(img_s.float(), ref, mask)):.2f} dB")
