"""Minimal `diffusers` surface for the reference's scripts when the real package is absent
(it is not installed on the target image and there is no network): exactly the names the
AF-LDM hot path imports, bound to afldm_amd's implementations."""
import importlib.util
import sys
import types


def install_diffusers_shim(force=False):
    """Register shim modules under `diffusers.*` unless the real diffusers is importable."""
    if not force and "diffusers" not in sys.modules and importlib.util.find_spec("diffusers") is not None:
        return False
    from .models import blocks
    from .models.unet_2d import UNet2DModel, UNet2DOutput
    from .pipelines.pipeline_utils import DiffusionPipeline, ImagePipelineOutput
    from .schedulers.ddim import DDIMScheduler, DDIMSchedulerOutput
    from .utils import randn_tensor

    def mod(name, **attrs):
        m = types.ModuleType(name)
        m.__dict__.update(attrs)
        m.__afldm_shim__ = True
        sys.modules[name] = m
        return m

    from .models.vae import AutoencoderKL

    root = mod("diffusers", UNet2DModel=UNet2DModel, DDIMScheduler=DDIMScheduler, AutoencoderKL=AutoencoderKL,
               DiffusionPipeline=DiffusionPipeline, __version__="0.32.1+afldm_amd_shim", __path__=[])
    root.models = mod("diffusers.models", UNet2DModel=UNet2DModel, AutoencoderKL=AutoencoderKL, __path__=[])
    root.models.unets = mod("diffusers.models.unets", __path__=[])
    mod("diffusers.models.unets.unet_2d", UNet2DModel=UNet2DModel, UNet2DOutput=UNet2DOutput)
    mod("diffusers.models.attention_processor", Attention=blocks.Attention, AttnProcessor2_0=blocks.AttnProcessor2_0)
    mod("diffusers.models.downsampling", Downsample2D=blocks.Downsample2D)
    mod("diffusers.models.upsampling", Upsample2D=blocks.Upsample2D)
    mod("diffusers.models.resnet", ResnetBlock2D=blocks.ResnetBlock2D)
    root.schedulers = mod("diffusers.schedulers", DDIMScheduler=DDIMScheduler, __path__=[])
    mod("diffusers.schedulers.scheduling_ddim", DDIMScheduler=DDIMScheduler, DDIMSchedulerOutput=DDIMSchedulerOutput)
    root.utils = mod("diffusers.utils", __path__=[])
    mod("diffusers.utils.torch_utils", randn_tensor=randn_tensor)
    root.pipelines = mod("diffusers.pipelines", DiffusionPipeline=DiffusionPipeline, __path__=[])
    mod("diffusers.pipelines.pipeline_utils", DiffusionPipeline=DiffusionPipeline,
        ImagePipelineOutput=ImagePipelineOutput)
    return True


def install():
    """Make `afldm.*` and (if needed) `diffusers.*` importable, backed by afldm_amd."""
    import afldm  # noqa: F401  (registers the alias finder)
    return install_diffusers_shim()
