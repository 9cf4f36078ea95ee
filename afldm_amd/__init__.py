"""afldm_amd — MI355X-native alias-free latent-diffusion denoising path.

The compute path is libafldm_hip.so (hand-written HIP for gfx950, C ABI in include/afldm_hip.h);
this package is the host side: diffusers-compatible modules, the reference's `afldm` API surface
and the graph-replayed sampler.  Importing the compute modules requires the built library —
there is no CPU fallback (the CPU restatement in oracle/ is test infrastructure only)."""
__version__ = "0.1.0"
