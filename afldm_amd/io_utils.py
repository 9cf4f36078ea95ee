"""PIL-only counterparts of reference afldm/io_utils.py:10-61 (torchvision / imageio are not
available on the target image)."""
import os

import numpy as np
import torch


def image_to_tensor(files, resolution=(512, 512)):
    from PIL import Image
    files = files if isinstance(files, list) else [files]
    out = []
    for f in files:
        img = Image.open(f).convert("RGB")
        if resolution is not None:
            img = img.resize((resolution[1], resolution[0]), Image.BILINEAR)
        t = torch.from_numpy(np.asarray(img, dtype=np.float32) / 255.0).permute(2, 0, 1)
        out.append(((t - 0.5) / 0.5).unsqueeze(0))
    return torch.cat(out)


def save_gif_from_tensors(tensors, output_gif_path, duration=0.5, denorm=False):
    from PIL import Image
    frames = []
    for t in tensors:
        t = t.detach().float().cpu()
        if denorm:
            t = (t + 1) / 2
        if t.ndim == 4:                       # 'n c h w -> c h (n w)'
            t = torch.cat(list(t), dim=-1)
        if t.shape[0] == 4:
            t = t[:3]
        arr = (torch.clamp(t, 0, 1) * 255).round().to(torch.uint8).permute(1, 2, 0).numpy()
        frames.append(Image.fromarray(arr.squeeze() if arr.shape[-1] == 1 else arr))
    d = os.path.dirname(output_gif_path)
    if d:
        os.makedirs(d, exist_ok=True)
    frames[0].save(output_gif_path, save_all=True, append_images=frames[1:], duration=int(duration * 1000), loop=0)
