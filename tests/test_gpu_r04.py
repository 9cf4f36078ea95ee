"""Round-4 parity cases on MI355X (VERDICT r03 items 3c, 3d, 5, 6 + the fused attention front-end):

* the product's I2SBScheduler.step - stochastic branch (is_ode=False) and clip_sample=True - and the AF block modules
  against fixtures recorded from the REFERENCE's own files (tests/golden/g15_r04_refpins.npz, oracle/gen_golden.py part i);
* the I2SB bridge's mid-point (evaluation 50) and a bf16 bound DERIVED from the oracle's own noise floor
  (tests/golden/g16_r04_floor.npz, part j);
* stochastic DDIM (eta > 0) against the oracle; WarpedNonlinearity around a module other than SiLU;
* the product's CrossFrameAttnProcessor against the reference file's control flow.

Tolerances: fp32 per-op max-abs <= 2e-5 max|ref|; bf16 per-op rel-RMS <= 2e-2."""
import numpy as np
import pytest
import torch

from test_gpu_r02 import build_unet, rel_rms

pytestmark = pytest.mark.gpu


def t(a):
    return torch.from_numpy(np.asarray(a))


def close32(got, ref, scale=2e-5):
    ref = torch.as_tensor(ref)
    got = got.float().cpu()
    assert got.shape == ref.shape
    return float((got - ref).abs().max()) <= scale * max(float(ref.abs().max()), 1e-6)


# ------------------------------------------------------------------------------------------------ item 3c
@pytest.mark.parametrize("clip", [False, True])
def test_i2sb_step_stochastic_and_clip_vs_reference_file(golden, clip):
    """afldm_amd.schedulers.i2sb.I2SBScheduler.step on the GPU (afldm_ddim_step_flat) against the reference file's own
    outputs: ODE and stochastic (seeded CPU generator, as diffusers' randn_tensor draws it) steps, clip_sample off / on
    (i2sb_scheduler.py:382-459), including pred_original_sample."""
    from afldm_amd.configs import FFHQ_DDIM_CONFIG
    from afldm_amd.schedulers.i2sb import I2SBScheduler
    g = golden("g15_r04_refpins.npz")
    cfg = {k: v for k, v in FFHQ_DDIM_CONFIG.items() if k != "set_alpha_to_one"}
    s = I2SBScheduler.from_config(dict(cfg, clip_sample=clip))
    s.set_timesteps(100)
    tag = "clip" if clip else "noclip"
    x, e = t(g["i2sb_step_x"]).cuda(), t(g["i2sb_step_eps"]).cuda()
    for ts in (991, 501, 11):
        out = s.step(e, ts, x, is_ode=True)
        assert close32(out.prev_sample, g[f"i2sb_step_ode_{tag}_{ts}"]), (tag, ts)
        if clip:
            assert close32(out.pred_original_sample, g[f"i2sb_step_x0_{tag}_{ts}"])
        sde = s.step(e, ts, x, is_ode=False, generator=torch.Generator().manual_seed(1000 + ts)).prev_sample
        assert close32(sde, g[f"i2sb_step_sde_{tag}_{ts}"]), (tag, ts)
        assert not torch.equal(sde, out.prev_sample)
    # the pipeline default is the stochastic bridge (reference i2sb_pipeline.py:27: is_ode=False)
    import inspect
    from afldm_amd.pipelines.i2sb_pipeline import I2SBLDMPipeline
    assert inspect.signature(I2SBLDMPipeline.__call__).parameters["is_ode"].default is False


def test_i2sb_stochastic_bridge_tiny_vs_oracle():
    """I2SBLDMPipeline._bridge with is_ode=False and clip_sample=True (the reference scheduler's default,
    i2sb_scheduler.py:152) on the tiny UNet: the seeded noise stream of the product equals the oracle's draw for draw."""
    from afldm_amd.pipelines.i2sb_pipeline import I2SBLDMPipeline
    from afldm_amd.schedulers.i2sb import I2SBScheduler
    from afldm_amd.configs import FFHQ_DDIM_CONFIG
    from oracle import i2sb as oi, unet as ou
    cfg = {k: v for k, v in FFHQ_DDIM_CONFIG.items() if k != "set_alpha_to_one"}
    unet, ucfg, sd = build_unet("tiny", torch.float32)
    start = 0.8 * torch.randn(2, 4, 16, 16, generator=torch.Generator().manual_seed(5))
    o = oi.I2SB(clip_sample=True)
    o.set_timesteps(10)
    lat, gen = start.clone(), torch.Generator().manual_seed(99)
    for ts in o.timesteps[:9]:
        lat = o.step(ou.unet_forward(sd, ucfg, lat, ts), ts, lat, is_ode=False, generator=gen)
    pipe = I2SBLDMPipeline(None, unet, I2SBScheduler.from_config(dict(cfg, clip_sample=True)))
    pipe.set_progress_bar_config(disable=True)
    got = pipe._bridge(start.cuda(), 10, False, torch.Generator().manual_seed(99))
    r = rel_rms(got, lat)
    print(f"[I2SB stochastic bridge, clip_sample] rel-RMS vs oracle {r:.3e}")
    assert r <= 1e-3


# ------------------------------------------------------------------------------------------------ item 3d
@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
def test_ffhq_i2sb_bridge_midpoint_and_derived_bf16_bound(golden, dtype):
    """The 99-evaluation bridge of BASELINE configs[4] at FFHQ size (batch 1) against the fp32 oracle at evaluation 50 AND
    99 (g14_r03.npz).  fp32: 1e-3.  bf16: the bounds are DERIVED from the oracle's own noise floors (part j,
    g16_r04_floor.npz), not fitted:
      * `floor_*_bf16lat` - the ORACLE with bf16-rounded weights and the latent stored in bf16 between evaluations, which
        is what the reference does with a bf16 UNet (0.059 / 0.108: the bridge moves the latent by ~1 bf16 ulp per step, so
        storage rounding dominates; round 3's measured 0.108 was exactly this).  A reference-style loop on the GPU
        (`sched.step` on bf16 latents) must stay within 1.25x of it (measured: 1.001x);
      * `floor_*` - the oracle with bf16-rounded weights only (1.8e-3 / 2.0e-3).  The product's bridge
        (I2SBLDMPipeline._bridge) carries the latent in fp32 between evaluations: what remains is weight rounding plus
        the rounding of the activation tensors, bounded here by 3x (measured 1.46x)."""
    from afldm_amd.configs import FFHQ_DDIM_CONFIG
    from afldm_amd.pipelines.i2sb_pipeline import I2SBLDMPipeline
    from afldm_amd.schedulers.i2sb import I2SBScheduler
    g, fl = golden("g14_r03.npz"), golden("g16_r04_floor.npz")
    cfg = {k: v for k, v in FFHQ_DDIM_CONFIG.items() if k != "set_alpha_to_one"}
    unet, _, _ = build_unet("ffhq", dtype)
    sched = I2SBScheduler.from_config(cfg)
    pipe = I2SBLDMPipeline(None, unet, sched)
    pipe.set_progress_bar_config(disable=True)
    start = t(g["i2sb_start"]).cuda().to(dtype)
    out = pipe._bridge(start, 100, True, None)
    assert out.dtype == dtype
    r99 = rel_rms(out.float(), g["i2sb_final99"])
    # the reference-style loop (latent in the UNet's dtype), with the mid-point
    lat = start
    sched.set_timesteps(100)
    for k, ts in enumerate(sched._timesteps_host[:99]):
        lat = sched.step(unet(lat, ts).sample, ts, lat, is_ode=True).prev_sample
        if k == 49:
            s50 = rel_rms(lat.float(), g["i2sb_eval50"])
    s99 = rel_rms(lat.float(), g["i2sb_final99"])
    fw50, fw99 = float(fl["floor_eval50"]), float(fl["floor_final99"])
    fs50, fs99 = float(fl["floor_eval50_bf16lat"]), float(fl["floor_final99_bf16lat"])
    print(f"[C5 bridge] {dtype}: product bridge final99 {r99:.3e}; reference-style loop eval50 {s50:.3e} final99 {s99:.3e}; "
          f"oracle floors: bf16 weights {fw50:.3e} / {fw99:.3e}, + bf16 latent storage {fs50:.3e} / {fs99:.3e}")
    if dtype == torch.float32:
        assert r99 <= 1e-3 and s50 <= 1e-3 and s99 <= 1e-3
    else:
        assert s50 <= 1.25 * fs50 and s99 <= 1.25 * fs99, (s50, fs50, s99, fs99)      # measured 1.0005x / 1.001x
        assert r99 <= 3.0 * fw99, (r99, fw99)                                          # measured 1.46x (2.9e-3)


# ------------------------------------------------------------------------------------------------ item 5
def test_ddim_eta_vs_oracle():
    """Stochastic DDIM (eta = 0.7): DDIMScheduler.step and MyLDMPipeline.__call__(eta=...) (reference
    ldm_pipeline.py:38,96-109 forwards eta to diffusers' scheduler) against the oracle's restatement, same CPU generator."""
    from afldm_amd.pipelines.ldm_pipeline import MyLDMPipeline
    from afldm_amd.schedulers.ddim import ffhq_ddim_scheduler
    from oracle import ddim as od, unet as ou
    unet, ucfg, sd = build_unet("tiny", torch.float32)
    x = torch.randn(2, 4, 16, 16, generator=torch.Generator().manual_seed(3))
    o = od.DDIM()
    o.set_timesteps(8)
    lat, gen = x.clone(), torch.Generator().manual_seed(11)
    for ts in o.timesteps:
        lat = o.step(ou.unet_forward(sd, ucfg, lat, ts), ts, lat, eta=0.7, generator=gen)
    pipe = MyLDMPipeline(None, unet, ffhq_ddim_scheduler())
    pipe.set_progress_bar_config(disable=True)
    got = pipe(latents=x, eta=0.7, num_inference_steps=8, generator=torch.Generator().manual_seed(11), output_type="latent")
    det = pipe(latents=x, eta=0.0, num_inference_steps=8, output_type="latent")
    r = rel_rms(got, lat)
    print(f"[DDIM eta=0.7, 8 steps] rel-RMS vs oracle {r:.3e}")
    assert r <= 1e-3 and rel_rms(det, lat) > 1e-2
    # one step, given variance_noise
    s = ffhq_ddim_scheduler()
    s.set_timesteps(8)
    e = torch.randn(2, 4, 16, 16, generator=torch.Generator().manual_seed(4))
    nz = torch.randn(2, 4, 16, 16, generator=torch.Generator().manual_seed(12))
    one = s.step(e.cuda(), 501, x.cuda(), eta=0.5, variance_noise=nz.cuda()).prev_sample
    ref = o.step(e, 501, x, eta=0.5, generator=torch.Generator().manual_seed(12))
    assert close32(one, ref)
    with pytest.raises(ValueError):
        s.step(e.cuda(), 501, x.cuda(), eta=0.5, variance_noise=nz.cuda(), generator=torch.Generator())


# ------------------------------------------------------------------------------------------------ item 6 + 3b
def _nhwc(a, dtype=torch.float32):
    return t(a).permute(0, 2, 3, 1).contiguous().cuda().to(dtype)


def _nchw(y):
    return y.float().permute(0, 3, 1, 2).contiguous().cpu()


def test_warped_nonlinearity_any_module_vs_reference_file(golden):
    """WarpedNonlinearity wraps ANY module (reference af_blocks.py:12-28): Tanh runs as HIP upsample -> module -> HIP
    low-pass + decimate; SiLU as the fused kernel; < 4-D inputs get the plain nonlinearity."""
    import torch.nn as nn
    from afldm_amd.af_modules.af_blocks import WarpedNonlinearity
    g = golden("g15_r04_refpins.npz")
    x = _nhwc(g["afb_wn_4d_in"])
    assert close32(_nchw(WarpedNonlinearity(nn.Tanh())(x)), g["afb_wn_tanh_out"])
    assert close32(_nchw(WarpedNonlinearity(nn.SiLU())(x)), g["afb_wn_4d_out"])
    v = t(g["afb_wn_2d_in"]).cuda()
    assert close32(WarpedNonlinearity(nn.SiLU())(v), g["afb_wn_2d_out"])
    assert close32(WarpedNonlinearity(nn.Tanh())(v), torch.tanh(t(g["afb_wn_2d_in"])))


@pytest.mark.parametrize("N", [8, 16])
def test_af_block_modules_vs_reference_file(golden, N):
    """AliasFreeDownsample2D (padding 1 and the padding == 0 branch) / AliasFreeUpsample2D module forwards of the product
    against the outputs of the reference's own modules (af_blocks.py:64-106,135-152) on the same convolution weights."""
    import torch.nn as nn
    from afldm_amd.af_modules.af_blocks import AliasFreeDownsample2D, AliasFreeUpsample2D
    g = golden("g15_r04_refpins.npz")
    C = g[f"afb_x_{N}"].shape[1]
    x = _nhwc(g[f"afb_x_{N}"])
    for pad in (1, 0):
        conv = nn.Conv2d(C, C, 3, 2, pad)
        conv.load_state_dict({"weight": t(g[f"afb_down_w_{N}_{pad}"]), "bias": t(g[f"afb_down_b_{N}_{pad}"])})
        blk = AliasFreeDownsample2D(C, True, out_channels=C, padding=pad, ori_conv=conv).cuda()
        assert conv.stride in (1, (1, 1))
        assert close32(_nchw(blk(x)), g[f"afb_down_{N}_{pad}"]), pad
    conv = nn.Conv2d(C, C, 3, 1, 1)
    conv.load_state_dict({"weight": t(g[f"afb_up_w_{N}"]), "bias": t(g[f"afb_up_b_{N}"])})
    blk = AliasFreeUpsample2D(C, True, ori_conv=conv, out_channels=C).cuda()
    assert close32(_nchw(blk(x)), g[f"afb_up_{N}"])
    yb = _nchw(blk.to(torch.bfloat16)(x.to(torch.bfloat16)))
    assert rel_rms(yb, g[f"afb_up_bf16_{N}"]) <= 2e-2


def test_cross_frame_processor_vs_reference_file(golden):
    """afldm_amd.pipelines.cross_frame_attn.CrossFrameAttnProcessor on an Attention block of the product against the
    outputs of the REFERENCE's CrossFrameAttnProcessor (cross_frame_attn.py:66-130) on the same layers: STORE into both
    map sets, LOAD with the batch-repeat branch (stored batch 1, query batch 2), enable_interp blend, IDLE."""
    from afldm_amd.models.blocks import Attention
    from afldm_amd.pipelines.cross_frame_attn import AttnState, CrossFrameAttnProcessor
    g = golden("g15_r04_refpins.npz")
    C, heads, groups = 32, 2, 8
    attn = Attention(C, heads=heads, dim_head=C // heads, eps=1e-5, norm_num_groups=groups, residual_connection=True,
                     bias=True, upcast_softmax=True, _from_deprecated_attn_block=True)
    attn.load_state_dict({k[len("cfa_sd_"):]: t(g[k]) for k in g.files if k.startswith("cfa_sd_")})
    attn = attn.cuda()
    st = AttnState()
    proc = CrossFrameAttnProcessor(st, enable_interp=True)
    attn.set_processor(proc)
    xa, xb, xq = _nhwc(g["cfa_xa"]), _nhwc(g["cfa_xb"]), _nhwc(g["cfa_xq"])
    st.set_timestep(torch.tensor(7))
    assert close32(_nchw(attn(xa)), g["cfa_store0"], 5e-5)
    st.set_store_id(1)
    assert close32(_nchw(attn(xb)), g["cfa_store1"], 5e-5)
    st.to_load()
    st.set_alpha(0.3)
    assert close32(_nchw(attn(xq)), g["cfa_load_interp"], 5e-5)
    proc.enable_interp = False
    assert close32(_nchw(attn(xq)), g["cfa_load"], 5e-5)
    st.to_idle()
    assert close32(_nchw(attn(xq)), g["cfa_idle"], 5e-5)


# ------------------------------------------------------------------------------------------------ fused attention front end
def _attn_block_reference(x, gamma, beta, G, eps, wq, wk, wv, bq, bk, bv, heads):
    """fp32 CPU restatement of group_norm -> to_q / to_k / to_v -> SDPA (oracle/unet.py::_attention_core up to to_out)."""
    import torch.nn.functional as F
    B, T, C = x.shape
    h = F.group_norm(x.transpose(1, 2), G, gamma, beta, eps).transpose(1, 2)
    q, k, v = F.linear(h, wq, bq), F.linear(h, wk, bk), F.linear(h, wv, bv)
    d = C // heads
    q, k, v = (z.view(B, T, heads, d).transpose(1, 2) for z in (q, k, v))
    return F.scaled_dot_product_attention(q, k, v).transpose(1, 2).reshape(B, T, C)


@pytest.mark.parametrize("T,C,heads,B,hot", [(1024, 192, 8, 3, 1.0), (256, 384, 16, 5, 1.0), (256, 64, 4, 2, 1.0),
                                             (64, 128, 8, 3, 1.0), (1024, 192, 8, 2, 6.0), (256, 384, 16, 2, 5.0)])
def test_attn_block_fused_vs_three_launch_path_and_reference(T, C, heads, B, hot):
    """afldm_attn_block_fused (GroupNorm-apply + per-head q|k|v projection + attention with K / V^T resident in LDS, one
    launch) against (a) the fp32 restatement of diffusers' AttnProcessor2_0 front end and (b) the three-launch path it
    replaces (afldm_gn_apply, afldm_conv2d, afldm_attention) on the same bf16 inputs.  `hot` scales to_q / to_k so that the
    scores have a standard deviation of tens of log2 units: the row maxima then outgrow the first tile's reference by
    more than 2^10 and the lazy-rescale branch (incl. the fix-up of the score tile computed one step ahead) runs.
    Asymmetric random data throughout (a transposed or permuted fragment cannot pass)."""
    from afldm_amd import ops
    gen = torch.Generator().manual_seed(T + C)
    G, eps = 32, 1e-5
    x = (torch.randn(B, T, C, generator=gen) * (1.0 + torch.rand(1, 1, C, generator=gen)) + 0.5 * torch.randn(1, 1, C, generator=gen))
    gamma, beta = 0.5 + torch.rand(C, generator=gen), 0.3 * torch.randn(C, generator=gen)
    ws = [torch.randn(C, C, generator=gen) / C ** 0.5 for _ in range(3)]
    ws[0], ws[1] = ws[0] * hot, ws[1] * hot
    bs = [0.2 * torch.randn(C, generator=gen) for _ in range(3)]
    xb = x.to(torch.bfloat16)
    wb = [w.to(torch.bfloat16) for w in ws]
    ref = _attn_block_reference(xb.float(), gamma, beta, G, eps, *[w.float() for w in wb], *bs, heads)
    xg = xb.cuda()
    side = int(T ** 0.5)
    stats = ops.gn_stats(xg.view(B, side, side, C), G)
    gg, bg = gamma.cuda(), beta.cuda()
    wpack = ops.pack_weight(torch.cat(wb, 0).float().cuda(), torch.bfloat16)
    bpack = torch.cat(bs, 0).cuda()
    got = ops.attn_block_fused(xg, stats, gg, bg, G, eps, wpack, bpack, heads, (C // heads) ** -0.5)
    torch.cuda.synchronize()
    # (b) the path it replaces
    hn = ops.gn_apply(xg.view(B, side, side, C), stats, gg, bg, G, eps, act=0).view(B, T, C)
    qk, vt = ops.linear_split(hn, wpack, bpack, 2 * C)
    old = ops.attention(qk[:, :, :C], qk[:, :, C:], vt, heads, scale=(C // heads) ** -0.5)
    r_ref, r_old, r_old_ref = rel_rms(got.float(), ref), rel_rms(got.float(), old.float().cpu()), rel_rms(old.float(), ref)
    print(f"[attn fused] T={T} C={C} heads={heads} hot={hot}: vs fp32 reference {r_ref:.3e} (three-launch path {r_old_ref:.3e}), "
          f"vs three-launch path {r_old:.3e}")
    # (hot: the softmax is nearly one-hot, bf16 rounding of q / k moves scores by ~0.1 log2 units and both paths sit at ~2e-2)
    assert r_ref <= (2e-2 if hot == 1.0 else 5e-2) and r_ref <= 1.5 * r_old_ref + 1e-3 and r_old <= (2e-2 if hot == 1.0 else 5e-2)
    assert torch.equal(got, ops.attn_block_fused(xg, stats, gg, bg, G, eps, wpack, bpack, heads, (C // heads) ** -0.5))
    # the two attention loops of the kernel: the bounded one (Cauchy-Schwarz says no score can outgrow the first tile's
    # reference by 2^80: no maxima, no rescaling) and the one that tracks row maxima with lazy rescaling.  `hot` inputs
    # take the second by themselves (the forced run is then bit-identical); ordinary inputs take the first, and forcing
    # the second must give the same result up to the rounding of the (different) references
    import os
    os.environ["AFLDM_ATTNF_SLOW"] = "1"
    try:
        slow = ops.attn_block_fused(xg, stats, gg, bg, G, eps, wpack, bpack, heads, (C // heads) ** -0.5)
    finally:
        del os.environ["AFLDM_ATTNF_SLOW"]
    r_slow = rel_rms(slow.float(), ref)
    print(f"             row-maxima loop forced: vs fp32 reference {r_slow:.3e}, vs default {rel_rms(slow.float(), got.float().cpu()):.3e}")
    assert r_slow <= (2e-2 if hot == 1.0 else 5e-2) and r_slow <= 1.5 * r_old_ref + 1e-3
    if hot > 1.0:
        assert torch.equal(slow, got)


@pytest.mark.parametrize("T,C,heads", [(1024, 192, 8), (256, 384, 16)])
def test_attn_block_fused_full_size_properties(T, C, heads):
    """BASELINE configs[1] size (batch 64, the FFHQ UNet's 32^2 / 16^2 attention levels), where the fp32 restatement takes
    too long: size-independent properties of GroupNorm -> q | k | v -> softmax attention.  (a) batch invariance: a sample
    computed inside the batch of 64 is bit-identical to the same sample launched on its own (a workgroup is one (sample,
    head) and nothing else enters it); (b) token-permutation equivariance: GroupNorm statistics, projections and the
    softmax sums are invariant under a permutation P of the tokens, so out(P x) = P out(x) up to the order of the fp32
    accumulation over keys and the bf16 rounding of the softmax weights; (c) the output rows are convex combinations of the
    value rows: every output channel lies inside the [min, max] of that channel of v over the sample (v from the
    three-launch path's projection); (d) finite everywhere."""
    from afldm_amd import ops
    B, G, eps = 64, 32, 1e-5
    gen = torch.Generator().manual_seed(C + 7)
    x = (torch.randn(B, T, C, generator=gen) * (0.7 + torch.rand(1, 1, C, generator=gen)) + 0.3 * torch.randn(1, 1, C, generator=gen))
    xg = x.to(torch.bfloat16).cuda()
    gamma, beta = (0.5 + torch.rand(C, generator=gen)).cuda(), (0.3 * torch.randn(C, generator=gen)).cuda()
    w = torch.randn(3 * C, C, generator=gen) / C ** 0.5
    wpack = ops.pack_weight(w.cuda(), torch.bfloat16)
    bpack = (0.2 * torch.randn(3 * C, generator=gen)).cuda()
    side = int(T ** 0.5)
    scale = (C // heads) ** -0.5
    run = lambda xx: ops.attn_block_fused(xx, ops.gn_stats(xx.view(xx.shape[0], side, side, C), G), gamma, beta, G, eps, wpack, bpack,
                                          heads, scale)
    out = run(xg)
    torch.cuda.synchronize()
    assert bool(torch.isfinite(out.float()).all())
    for b in (0, 17, 63):                                                            # (a)
        assert torch.equal(run(xg[b:b + 1].contiguous())[0], out[b])
    perm = torch.randperm(T, generator=gen).cuda()                                   # (b)
    outp = run(xg[:, perm].contiguous())
    r = rel_rms(outp.float(), out[:, perm].float().cpu())
    print(f"[attn fused full size] T={T} C={C}: token permutation rel-RMS {r:.3e}")
    assert r <= 4e-3
    hn = ops.gn_apply(xg.view(B, side, side, C), ops.gn_stats(xg.view(B, side, side, C), G), gamma, beta, G, eps, act=0).view(B, T, C)
    _, vt = ops.linear_split(hn, wpack, bpack, 2 * C)                                # (c) v^T [B, C, T]
    vmin, vmax = vt.float().amin(2), vt.float().amax(2)                              # [B, C]
    o = out.float()
    slack = 2e-2 * (vmax - vmin).unsqueeze(1) + 1e-3                                 # bf16 rounding of the weights and of v itself
    assert bool((o <= vmax.unsqueeze(1) + slack).all()) and bool((o >= vmin.unsqueeze(1) - slack).all())


def test_attn_block_fused_rejects_what_it_has_no_kernel_for():
    from afldm_amd import _lib, ops
    x = torch.zeros(2, 64, 384, dtype=torch.bfloat16, device="cuda")
    st = ops.gn_stats(x.view(2, 8, 8, 384), 32)
    z = torch.zeros(384, device="cuda")
    w = torch.zeros(3 * 384, 1, 1, 384, dtype=torch.bfloat16, device="cuda")
    with pytest.raises(_lib.AfldmError, match="no kernel"):
        ops.attn_block_fused(x, st, z, z, 32, 1e-5, w, torch.zeros(3 * 384, device="cuda"), 16, 0.2)
    with pytest.raises(_lib.AfldmError, match="bf16 only"):
        xf = x.float()
        ops.attn_block_fused(xf, ops.gn_stats(xf.view(2, 8, 8, 384), 32), z, z, 32, 1e-5, w.float(),
                             torch.zeros(3 * 384, device="cuda"), 16, 0.2)
    assert not ops.attn_block_fused_ok(x, 16, 32)


@pytest.mark.parametrize("cfg_name", ["tiny", "ffhq"])
def test_unet_forward_through_fused_attention_blocks(golden, cfg_name, monkeypatch):
    """The UNet forward with every eligible attention block on the fused launch (the policy's workgroup floor lifted so
    that a batch of 1-2 takes it) against the oracle fixtures, and against the three-launch path of the same model."""
    from afldm_amd import ops
    g = golden("g6_tiny_unet.npz" if cfg_name == "tiny" else "g6_ffhq_unet.npz")
    unet, _, _ = build_unet(cfg_name, torch.bfloat16)
    x = t(g["x"]).cuda()
    ts, key = (501, "y_af") if cfg_name == "tiny" else (981, "y_t981")
    monkeypatch.setattr(ops, "_FUSED_ATTN_MIN_WGS", 0)
    calls = []
    real = ops.attn_block_fused
    monkeypatch.setattr(ops, "attn_block_fused", lambda *a, **k: (calls.append(a[0].shape), real(*a, **k))[1])
    y_f = unet(x, ts).sample
    assert len(calls) >= (2 if cfg_name == "tiny" else 10), calls
    monkeypatch.setattr(ops, "_FUSED_ATTN", False)
    y_o = unet(x, ts).sample
    r_f, r_o = rel_rms(y_f.float(), g[key]), rel_rms(y_o.float(), g[key])
    print(f"[UNet through fused attention] {cfg_name}: {len(calls)} fused blocks; vs oracle {r_f:.3e} (three-launch {r_o:.3e}); "
          f"fused vs three-launch {rel_rms(y_f.float(), y_o.float().cpu()):.3e}")
    assert r_f <= 2e-2 and r_f <= 2.0 * r_o + 1e-3


# ------------------------------------------------------------------------------------------------ persistent halo-patch tiles
@pytest.mark.parametrize("case", [
    # variant, reference variant, B, H, W, Cin, Cout, temb, residual
    (63, 58, 5, 64, 64, 128, 128, False, True),      # 80 tiles of 8 x 32: every workgroup gets exactly one ... at 256 CUs; B below gives several
    (63, 58, 40, 64, 64, 128, 128, True, True),      # 640 tiles: 2-3 per workgroup, producers run across tile boundaries
    (63, 58, 9, 128, 128, 128, 256, False, False),   # two n tiles, 1152 tiles: 4-5 per workgroup
    (63, 58, 3, 72, 96, 128, 256, False, True),      # non-square plane (9 x 3 blocks of 8 x 32 pixels)
    (64, 46, 24, 32, 32, 192, 192, True, True),      # UNet 32^2 level: 4 rows per tile, 192 tiles
    (64, 46, 66, 32, 32, 384, 192, False, True),     # 528 tiles: 2-3 per workgroup, ragged
])
def test_conv3x3_halo_patch_persistent_tiles_bit_identical(case):
    """conv3h.hip k_conv3h_pers (round 4): persistent workgroups whose LDS-DMA producers run on across tile boundaries
    (next tile's first patch / weights land while the epilogue of the current tile is staged through the other patch
    buffer) must be BIT-IDENTICAL - output and GroupNorm partial sums - to the one-tile-per-workgroup kernel of the same
    tile shape, and match F.conv2d."""
    import ctypes
    import torch.nn.functional as F
    from afldm_amd import _lib, ops
    vid, ref_vid, B, H, W, Cin, Cout, use_temb, use_res = case
    dt = torch.bfloat16
    g = torch.Generator().manual_seed(vid + B + H)
    x = torch.randn(B, H, W, Cin, generator=g).to(dt).cuda()
    w = (torch.randn(Cout, Cin, 3, 3, generator=g) / (3 * Cin ** 0.5)).to(dt)
    b = torch.randn(Cout, generator=g).cuda()
    temb = torch.randn(B, Cout, generator=g).to(dt).cuda() if use_temb else None
    res = torch.randn(B, H, W, Cout, generator=g).to(dt).cuda() if use_res else None
    wp = ops.pack_weight(w.float().cuda(), dt)
    outs = {}
    for v in (vid, ref_vid):
        try:
            _lib.check(_lib.lib.afldm_conv2d_tune(v, -1), "tune")
            ys = [ops.conv2d(x, wp, b, temb=temb, temb_stride=Cout if use_temb else 0, residual=res, want_stats=True) for _ in range(2)]
            probe = ops.conv_args(x, wp, b, temb=temb, temb_stride=Cout if use_temb else 0, residual=res, out=ys[0])
            assert _lib.lib.afldm_conv2d_variant(ctypes.byref(probe)) & 255 == v, f"variant {v} did not take this shape"
        finally:
            _lib.lib.afldm_conv2d_tune(-1, -1)
        torch.cuda.synchronize()
        assert torch.equal(ys[0], ys[1]) and torch.equal(ys[0].gn_partial, ys[1].gn_partial)
        outs[v] = ys[0]
    assert torch.equal(outs[vid], outs[ref_vid]), float((outs[vid].float() - outs[ref_vid].float()).abs().max())
    assert torch.equal(outs[vid].gn_partial, outs[ref_vid].gn_partial)
    # and against F.conv2d on a few samples
    for i in (0, B - 1):
        ref = F.conv2d(x[i:i + 1].float().permute(0, 3, 1, 2).cpu(), w.float(), b.cpu(), padding=1)
        if use_temb:
            ref = ref + temb[i].float().cpu()[None, :, None, None]
        if use_res:
            ref = ref + res[i:i + 1].float().permute(0, 3, 1, 2).cpu()
        assert rel_rms(outs[vid][i:i + 1].float().permute(0, 3, 1, 2), ref) <= 6e-3


def test_af_act_trace_fills_stamps_and_leaves_the_result_alone():
    """afldm_af_act_trace: a diagnostic - with a stamp buffer armed the plane kernel writes monotonically increasing
    shader-clock stamps for each workgroup's first items and produces the very same output."""
    from afldm_amd import _lib, ops
    B, N, C, G = 8, 16, 64, 32
    gen = torch.Generator().manual_seed(5)
    x = (torch.randn(B, N, N, C, generator=gen) * 1.3 + 0.2).to(torch.bfloat16).cuda()
    st = ops.gn_stats(x, G)
    gamma, beta = (0.5 + torch.rand(C, generator=gen)).cuda(), torch.randn(C, generator=gen).cuda()
    ref = ops.af_act(x, None, st, gamma, beta, G, 1e-5)
    tr = torch.zeros(1024 * 4 * 2 * 10, dtype=torch.int64, device="cuda")
    _lib.check(_lib.lib.afldm_af_act_trace(tr.data_ptr()), "af_act_trace")
    try:
        got = ops.af_act(x, None, st, gamma, beta, G, 1e-5)
        torch.cuda.synchronize()
    finally:
        _lib.check(_lib.lib.afldm_af_act_trace(None), "af_act_trace")
    assert torch.equal(got, ref)
    a = tr.cpu().view(1024, 4, 2, 10)
    used = a[:, 0, 0, 0] != 0
    assert int(used.sum()) >= 1
    first = a[used][:, :, 0, :]                                   # [workgroup, wave, stamp] of the first item
    assert bool((first[:, :, 1:] >= first[:, :, :-1]).all()) and bool((first[:, :, 9] > first[:, :, 0]).all())
    # switched off again: the buffer stays untouched
    tr.zero_()
    ops.af_act(x, None, st, gamma, beta, G, 1e-5)
    torch.cuda.synchronize()
    assert int(tr.abs().sum()) == 0


def test_box_probes_run_and_check_their_arguments():
    """The measurement-only entry points bench.py's `box` record uses: both MFMA probes run; the random-operand probe
    refuses an iteration count that is not a multiple of its four rotating operand sets."""
    from afldm_amd import _lib
    lib = _lib.lib
    st = torch.cuda.current_stream().cuda_stream
    out = torch.zeros(64 * 256, dtype=torch.float32, device="cuda")
    assert lib.afldm_probe_mfma(out.data_ptr(), 64, 400, st) == 0
    assert lib.afldm_probe_mfma_random(out.data_ptr(), 64, 400, st) == 0
    torch.cuda.synchronize()
    assert lib.afldm_probe_mfma_random(out.data_ptr(), 64, 401, st) != 0
    assert b"multiple of 4" in lib.afldm_last_error()
    assert lib.afldm_probe_mfma_random(None, 64, 400, st) != 0


def test_bench_two_ranks_on_one_gpu_runs_the_multi_rank_path():
    """bench.py --gpus 2 launched the way the driver launches it (torch.distributed.run, one process per rank), on the ONE
    GPU of the test box: AFLDM_SAME_GPU=1 puts both ranks on device 0 and the backend is gloo (RCCL refuses two ranks on
    one device).  Not a scaling point - it executes everything the N > 1 line needs besides RCCL itself: per-rank noise
    shards and graph capture, barrier + MAX-over-ranks timing, the all-gather of the final latents inside the timed
    region, the process-group record; and it checks the record's honesty (2 ranks, ONE distinct device)."""
    import json, os, socket, subprocess, sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    env = dict(os.environ, AFLDM_SAME_GPU="1", AFLDM_DIST_BACKEND="gloo")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
           "--master-port", str(port), "bench.py", "--gpus", "2", "--steps", "3", "--warmup", "1", "--regions", "1",
           "--no-cpu-baseline", "--no-extras", "--no-roofline"]
    r = subprocess.run(cmd, cwd=root, env=env, capture_output=True, text=True, timeout=900)
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert r.returncode == 0 and len(lines) == 1, r.stdout[-2000:] + r.stderr[-3000:]
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and d["steps"] == 3 and d["scaling"] == "weak" and d["config"]["latents_finite"]
    assert d["config"]["global_batch"] == 128 and d["value"] > 0
    rc = d["rccl"]
    assert rc["world_size"] == 2 and rc["backend"] == "gloo" and len(rc["ranks"]) == 2 and rc["distinct_devices"] == 1
    assert rc["all_gather_verified"] is True and rc["all_gather_bytes"] == 128 * 4 * 32 * 32 * 4


@pytest.mark.parametrize("B,H,Cin,Cout,va,vb", [(8, 32, 192, 192, 55, 65), (1, 32, 192, 192, 55, 65), (1, 16, 384, 384, 54, 66),
                                                (4, 16, 768, 384, 54, 66), (8, 16, 384, 384, 54, 66)])
def test_conv3x3_small_batch_64_pixel_tiles_vs_128_pixel_tiles(B, H, Cin, Cout, va, vb):
    """Variants 65 / 66 of the halo-patch kernel (64-pixel tiles: the small-batch default of round 4) against the 128-pixel
    tiles they replace, on the same inputs with bias, residual and GroupNorm statistics: the K order is the same, so the
    outputs are bit-identical whenever both plans use the same number of channel-block slices (otherwise they differ by
    the order of the slab sums: one bf16 ulp); the per-channel statistics agree to fp32 summation order."""
    import ctypes
    from afldm_amd import _lib, ops
    g = torch.Generator().manual_seed(B + H + Cin)
    x = torch.randn(B, H, H, Cin, generator=g).to(torch.bfloat16).cuda()
    w = ops.pack_weight((torch.randn(Cout, Cin, 3, 3, generator=g) / (3 * Cin ** 0.5)).cuda(), torch.bfloat16)
    b = torch.randn(Cout, generator=g).cuda()
    res = torch.randn(B, H, H, Cout, generator=g).to(torch.bfloat16).cuda()
    outs = {}
    for v in (va, vb):
        _lib.check(_lib.lib.afldm_conv2d_tune(v, -1), "tune")
        try:
            y = ops.conv2d(x, w, b, residual=res, want_stats=True)
            a = ops.conv_args(x, w, b, residual=res, out=torch.empty_like(y))
            need = _lib.lib.afldm_conv2d_workspace(ctypes.byref(a))
            ws = torch.empty(max(need, 4) // 4, dtype=torch.float32, device="cuda")
            a.workspace, a.workspace_bytes = ops.ptr(ws), need
            code = _lib.lib.afldm_conv2d_variant(ctypes.byref(a))
        finally:
            _lib.lib.afldm_conv2d_tune(-1, -1)
        assert (code & 255) == v, f"variant {v} was not taken (plan says {code & 255})"
        outs[v] = (y, y.gn_partial.sum(1), (code >> 8) & 255)
    (ya, sa, za), (yb, sb, zb) = outs[va], outs[vb]
    if za == zb:
        assert torch.equal(ya, yb)
    else:
        assert float((ya.float() - yb.float()).abs().max()) <= 2.0 ** -6 * float(ya.float().abs().max())
    assert float((sa - sb).abs().max()) <= 1e-4 * float(sa.abs().max())
    ref = torch.nn.functional.conv2d(x.float().cpu().permute(0, 3, 1, 2), w.float().cpu().view(Cout, 3, 3, Cin).permute(0, 3, 1, 2),
                                     b.cpu(), padding=1).permute(0, 2, 3, 1) + res.float().cpu()
    assert rel_rms(yb.float(), ref) <= 6e-3
