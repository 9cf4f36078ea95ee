#!/usr/bin/env python
"""x4 super-resolution fractional-shift equivariance demo on MI355X - same CLI as the reference's
scripts/shift_ldm_sr.py (--num_inference_steps --shift_steps --output_path --input_path).

No network on the target machines: pass --ckpt /path/to/alias_free_ldm_sr (diffusers-format
directory with unet/, scheduler/, vae/) or --random-init for seeded random weights of the same
architecture and a synthetic input image (demonstrates the full flow; the picture is noise).
Multi-GPU: launch with torch.distributed.run; the shift offsets are sharded across ranks."""
import argparse
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402


def parse_args():
    p = argparse.ArgumentParser()
    p.add_argument("--num_inference_steps", type=int, default=50)
    p.add_argument("--shift_steps", type=int, default=16)
    p.add_argument("--output_path", type=str, default="results/shift_ldm_sr.gif")
    p.add_argument("--input_path", type=str, default="assets/bear_hr.jpg")
    p.add_argument("--ckpt", type=str, default=os.environ.get("AFLDM_SR_CKPT"))
    p.add_argument("--random-init", action="store_true")
    p.add_argument("--dtype", default="fp32", choices=["fp32", "bf16"])
    p.add_argument("--seed", type=int, default=1234)
    p.add_argument("--sequential", action="store_true",
                   help="one B = 1 sampler run per offset like the reference loop (default: all offsets of a rank in one batch)")
    p.add_argument("--eager", action="store_true",
                   help="run the STORE / LOAD passes as the eager per-step loop instead of replayed HIP graphs")
    return p.parse_args()


def main():
    args = parse_args()
    from afldm_amd import compat, parallel
    compat.install()
    from afldm.af_modules.af_api import make_af_unet, make_af_vae_from_config
    from afldm.pipelines.i2sb_pipeline import I2SBLDMPipeline
    from afldm_amd.harness import shift_ldm_sr
    rank, world, local = parallel.init_distributed()
    dtype = torch.bfloat16 if args.dtype == "bf16" else torch.float32
    image = None
    if args.ckpt:
        pipe = I2SBLDMPipeline.from_pretrained(args.ckpt)
    elif args.random_init:
        from afldm_amd.configs import FFHQ_UNET_CONFIG
        from afldm_amd.models.unet_2d import UNet2DModel
        from afldm_amd.models.vae import AutoencoderKL
        from afldm_amd.schedulers.i2sb import I2SBScheduler
        torch.manual_seed(0)
        unet = UNet2DModel.from_config(FFHQ_UNET_CONFIG)
        with torch.no_grad():
            unet.conv_out.weight.mul_(0.1)
            unet.conv_out.bias.mul_(0.1)
        vae = AutoencoderKL(in_channels=3, out_channels=3, down_block_types=["DownEncoderBlock2D"] * 4,
                            up_block_types=["UpDecoderBlock2D"] * 4, block_out_channels=[128, 256, 512, 512],
                            layers_per_block=2, latent_channels=4, scaling_factor=0.6, mid_act=True,
                            down_filtered_act=[False, True, True, True], up_filtered_act=[True, True, True, False],
                            up_rescale=[True, True, True])
        pipe = I2SBLDMPipeline(vae, unet, I2SBScheduler(clip_sample=False))
        if not os.path.exists(args.input_path):
            g = torch.Generator().manual_seed(args.seed)     # smooth synthetic image in [-1, 1]
            image = torch.nn.functional.interpolate(torch.rand(1, 3, 16, 16, generator=g) * 2 - 1, size=(256, 256),
                                                    mode="bicubic", align_corners=False).clamp(-1, 1)
    else:
        raise SystemExit("pass --ckpt DIR (or AFLDM_SR_CKPT) or --random-init; checkpoints cannot be downloaded here")
    pipe = pipe.to(f"cuda:{local}").to(dtype)
    pipe.set_progress_bar_config(disable=True)
    make_af_unet(pipe.unet)
    make_af_vae_from_config(pipe.vae)
    frames, errs = shift_ldm_sr(pipe, args.num_inference_steps, args.shift_steps, args.output_path, args.input_path,
                                image=image, rank=rank, world=world, batch_offsets=not args.sequential, use_graph=not args.eager)
    if rank == 0:
        print(f"wrote {args.output_path}: {len(frames)} frames; latent equivariance mask-MSE per offset:",
              " ".join(f"{e:.3e}" for e in errs))


if __name__ == "__main__":
    main()
