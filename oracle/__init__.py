"""CPU oracle for the AF-LDM denoising hot path.  TEST INFRASTRUCTURE ONLY.

This package is a plain-PyTorch (CPU, fp32, eager) restatement of the
reference algorithm on the hot path of SingleZombie/AFLDM:

  * ideal FFT-domain low-pass / reconstruction filters and the x-up zero-stuff
    upsampler          (reference afldm/af_libs/ideal_lpf.py:12-172)
  * WarpedNonlinearity / AliasFreeUpsample2D / AliasFreeDownsample2D bodies
                       (reference afldm/af_modules/af_blocks.py:12-152)
  * the diffusers UNet2DModel forward after make_af_unet surgery
                       (reference afldm/af_modules/af_api.py:70-83; diffusers
                        0.32.1 semantics restated from SURVEY.md Appendix A)
  * the cross-frame attention STORE/LOAD hook
                       (reference afldm/pipelines/cross_frame_attn.py:66-130)
  * DDIMScheduler      (diffusers 0.25/0.32 semantics; reference config
                        configs/ldm/noise_scheduler.json:1-14)
  * the DDIM sampling loop and the fractional-shift harness
                       (reference afldm/pipelines/ldm_pipeline.py:80-112,
                        scripts/shift_ldm_ffhq.py:49-159)
  * ImageShifter ('ideal', 'ideal_crop', bilinear) and the masked metrics
                       (reference afldm/shift_utils/shifters.py:31-206,
                        afldm/shift_utils/metrics.py:5-20)
  * the AF-VAE (AutoencoderKL after make_af_vae_from_config), the I2SB scheduler
    and the x4 super-resolution degrade operator build_sr4x
                       (reference afldm/models/af_vae.py:8-55, af_api.py:34-67,
                        afldm/schedulers/i2sb_scheduler.py:131-197,382-459,
                        afldm/af_libs/superresolution.py:160-320)

Pinning status
--------------
* The alias-free pieces, shifters, metrics and the super-resolution degrade
  operator (g9_sr4x) are pinned bit-for-bit / to 1e-6 (fp32)
  against the *imported* reference modules by ``oracle/gen_golden.py`` (run in
  the build container, where /root/reference exists); its outputs are committed
  under ``tests/golden/`` and re-checked by ``tests/test_oracle_golden.py``.
* The diffusers blocks (UNet2DModel, Attention, ResnetBlock2D, DDIMScheduler)
  live in a third-party package (``diffusers``, unpinned in the reference's
  requirements.txt:1; configs record 0.32.1 / 0.25.0) that is NOT vendored in the
  reference and NOT installed here, and the reference ships no tests or golden
  vectors for them.  Those parts of the oracle are therefore **parity unpinned**:
  they restate the published diffusers algorithm and are anchored only on the
  scheduler/embedding known-answer values of SURVEY.md Appendix C and on the
  reference's own call sites.

Only ``tests/``, ``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline``
leg may import this package.  The product (``afldm_amd``) never does.
"""
