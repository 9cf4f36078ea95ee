"""Small host helpers with diffusers-compatible names."""
import torch


def randn_tensor(shape, generator=None, device=None, dtype=None, layout=None):
    """diffusers.utils.torch_utils.randn_tensor: with a CPU generator and a non-CPU target the
    draw happens on CPU and is then moved (this is what makes seeded runs device-independent,
    reference ldm_pipeline.py:82-88)."""
    device = torch.device(device) if device is not None else torch.device("cpu")
    layout = layout or torch.strided
    rand_device = device
    if generator is not None:
        gen_device = (generator[0] if isinstance(generator, list) else generator).device.type
        if gen_device != device.type and gen_device == "cpu":
            rand_device = torch.device("cpu")
        elif gen_device != device.type and gen_device == "cuda":
            raise ValueError(f"Cannot generate a {device} tensor from a generator of type {gen_device}.")
    if isinstance(generator, list) and len(generator) == 1:
        generator = generator[0]
    if isinstance(generator, list):
        shape_1 = (1,) + tuple(shape[1:])
        latents = torch.cat([torch.randn(shape_1, generator=generator[i], device=rand_device, dtype=dtype,
                                         layout=layout) for i in range(shape[0])], dim=0).to(device)
    else:
        latents = torch.randn(shape, generator=generator, device=rand_device, dtype=dtype, layout=layout).to(device)
    return latents
