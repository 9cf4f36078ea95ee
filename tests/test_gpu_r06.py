"""Round-6 parity cases on MI355X.

(a) The plane-constant form of the 2x2 level: WarpedNonlinearity of a 2x2 plane leaves one value in all four pixels
    (reference ideal_lpf.py:17-21: lpf(4) = [1,0,0,0]), so the activation kernels store it once
    ([B, C]) and the 3x3 convolutions behind it run as dense layers over Cin columns with tap-summed weights
    (afldm_af_act_const2, afldm_af_act_slabs act = 2, blocks.packed_conv_dense2x2_const).  Asserted: the four pixels of
    the full form ARE bit-equal and equal the stored value; a ResnetBlock2D on 2x2 planes gives the full form's output
    within the rounding of one weight sum; the oracle-anchored forwards (test_gpu_unet / test_gpu_r02 / test_gpu_r05) run
    with the form on by default.
(b) The north-star harness on the graph path (reference scripts/shift_ldm_ffhq.py:85-151): CrossFrameSampler's STORE /
    LOAD graphs against the ORACLE's equivariance fixtures (same budgets as the eager tests: 0.2 dB fp32 / 1 dB bf16),
    against the eager loop, batched against one-run-per-offset, and replay after a new STORE pass.
"""
import numpy as np
import pytest
import torch

from test_gpu_r02 import build_unet, rel_rms

pytestmark = pytest.mark.gpu


# ----------------------------------------------------------------------------- (a) plane-constant 2x2 level
@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
def test_af_act_const2_is_the_full_forms_value(dtype):
    from afldm_amd import ops
    g = torch.Generator().manual_seed(3)
    B, C1, C2, G = 5, 96, 48, 8
    x1 = torch.randn(B, 2, 2, C1, generator=g).cuda().to(dtype)
    x2 = (torch.randn(B, 2, 2, C2, generator=g) * 2 + 0.5).cuda().to(dtype)
    gamma = (torch.rand(C1 + C2, generator=g) + 0.5).cuda()
    beta = (torch.randn(C1 + C2, generator=g) * 0.3).cuda()
    for a, b in ((x1, None), (x1, x2)):
        Ct = C1 + (0 if b is None else C2)
        st = ops.gn_stats(a, G, x2=b)
        full = ops.af_act(a, b, st, gamma[:Ct].contiguous(), beta[:Ct].contiguous(), G, 1e-5)
        const = ops.af_act(a, b, st, gamma[:Ct].contiguous(), beta[:Ct].contiguous(), G, 1e-5, out_const=True)
        assert const.shape == (B, Ct) and getattr(const, "const2", False)
        for h in range(2):
            for w in range(2):
                assert torch.equal(full[:, h, w, :], const), (h, w)
        # and without normalisation (a bare WarpedNonlinearity)
        assert torch.equal(ops.af_act(a, b)[:, 1, 0, :], ops.af_act(a, b, out_const=True))


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
def test_af_act_slabs_const2(dtype):
    from afldm_amd import ops
    g = torch.Generator().manual_seed(4)
    B, C, G, ns = 6, 192, 32, 3
    slabs = torch.randn(ns, B * 4, C, generator=g).cuda()
    bias = torch.randn(C, generator=g).cuda()
    temb = torch.randn(B, C, generator=g).cuda().to(dtype)
    gamma, beta = (torch.rand(C, generator=g) + 0.5).cuda(), torch.randn(C, generator=g).cuda()
    full = ops.af_act_slabs(slabs, ns, bias, temb, C, gamma, beta, G, 1e-5, B, 2, C, dtype, act=True)
    const = ops.af_act_slabs(slabs, ns, bias, temb, C, gamma, beta, G, 1e-5, B, 2, C, dtype, act=2)
    assert const.shape == (B, C)
    for p in range(4):
        assert torch.equal(full.view(B, 4, C)[:, p], const)


@pytest.mark.parametrize("dtype,tol", [(torch.float32, 2e-6), (torch.bfloat16, 8e-3)])
@pytest.mark.parametrize("cin,cout,concat,batch", [(128, 128, False, 64), (64, 192, False, 3), (384, 128, True, 64), (192, 64, True, 1)])
def test_resnet_block_on_2x2_planes_const_form_vs_full_form(monkeypatch, dtype, tol, cin, cout, concat, batch):
    """ResnetBlock2D (af_api surgery applied) on 2x2 planes: the plane-constant form against the flattened-plane form
    (AFLDM_NO_CONST2) - same function, weights summed over the taps before the one rounding instead of after the products."""
    from afldm_amd.af_modules.af_blocks import WarpedNonlinearity
    from afldm_amd.models import blocks
    torch.manual_seed(11)
    blk = blocks.ResnetBlock2D(in_channels=cin, out_channels=cout, temb_channels=64, groups=8, eps=1e-5)
    blk.nonlinearity = WarpedNonlinearity(blk.nonlinearity)
    blk = blk.cuda().to(dtype)
    g = torch.Generator().manual_seed(12)
    x = torch.randn(batch, 2, 2, cin, generator=g).cuda().to(dtype)
    inp = (x[..., :cin // 2].contiguous(), x[..., cin // 2:].contiguous()) if concat else x
    temb = torch.randn(1, cout, generator=g).cuda().to(dtype)
    gn = torch.nn.GroupNorm(8, cout).cuda().to(dtype)
    outs = {}
    assert blk._const2_ok(inp)
    for form in (True, False):
        monkeypatch.setattr(blocks, "_CONST2", form)
        blocks.invalidate_packed(blk)
        for ngn in (None, gn):
            y = blk(inp, temb.view(-1), 0, next_gn=ngn)
            assert y.shape == (batch, 2, 2, cout)
            outs[(form, ngn is not None)] = (y.float().clone(), getattr(y, "gn_partial", None), getattr(y, "gn_applied", None))
    for k in (False, True):
        a, b = outs[(True, k)][0], outs[(False, k)][0]
        r = rel_rms(a, b.cpu().numpy())
        assert r <= tol, (k, r)
    # the statistics the const form hands on describe its own output
    y, st, _ = outs[(True, False)]
    assert st is not None
    s = st.double().sum(1)
    assert torch.allclose(s[..., 0], y.double().sum((1, 2)), rtol=1e-4, atol=1e-3)


@pytest.mark.parametrize("dtype,tol", [(torch.float32, 3e-6), (torch.bfloat16, 6e-3)])
@pytest.mark.parametrize("B,Cin,Cout,G,per_sample_temb", [(64, 768, 768, 32, False), (3, 1536, 768, 32, True), (17, 256, 96, 8, False),
                                                          (1, 256, 64, 8, True), (16, 512, 192, 8, False)])
def test_conv2x2_const_norm_act_vs_separate_launches(dtype, tol, B, Cin, Cout, G, per_sample_temb):
    """afldm_conv2x2_const_norm_act (conv1 + temb -> norm2 -> activation in one launch, csrc/dense2.hip) against the launches it
    replaces - afldm_conv2d on the tap-summed weights, afldm_af_act_const2 on its output and statistics - and against a plain
    fp64 evaluation of the same chain."""
    from afldm_amd import ops
    from afldm_amd.models import blocks
    assert ops.conv2x2_const_norm_act_ok(Cin, Cout, G, dtype, batch=64)
    assert not ops.conv2x2_const_norm_act_ok(Cin, Cout, G, dtype, batch=1)       # (policy: the two launches win below batch 64)
    g = torch.Generator().manual_seed(B * 7 + Cin)
    conv = torch.nn.Conv2d(Cin, Cout, 3, padding=1).cuda()
    with torch.no_grad():
        conv.weight.copy_(torch.randn(conv.weight.shape, generator=g) * (2.0 / (9 * Cin)) ** 0.5)
        conv.bias.copy_(torch.randn(Cout, generator=g) * 0.2)
    a = torch.randn(B, Cin, generator=g).cuda().to(dtype)
    temb = torch.randn(B if per_sample_temb else 1, Cout, generator=g).cuda().to(dtype)
    ts = Cout if per_sample_temb else 0
    gamma, beta = (torch.rand(Cout, generator=g) + 0.5).cuda(), (torch.randn(Cout, generator=g) * 0.3).cuda()
    got = ops.conv2x2_const_norm_act(a, blocks.packed_conv_dense2x2_const_cm(conv, dtype), blocks._bias_f32(conv), temb.view(-1) if ts == 0 else temb,
                                     ts, gamma, beta, G, 1e-5)
    assert got.shape == (B, Cout)
    # the separate launches
    w1, b1 = blocks.packed_conv_dense2x2_const(conv, dtype)
    y = ops.conv2d(a, w1, b1, temb=temb.view(-1) if ts == 0 else temb, temb_stride=ts, temb_mod=Cout, want_stats=True)
    y4 = blocks._plane2_view(y, B, Cout)
    ref = ops.af_act(y4, None, ops.gn_stats(y4, G), gamma, beta, G, 1e-5, out_const=True)
    assert rel_rms(got.float(), ref.float().cpu().numpy()) <= tol
    # fp64 on the host: full 3x3 convolution of the constant planes, GroupNorm, up x2 -> SiLU -> mean
    from oracle import ideal_filters as idf
    xa = a.double().cpu()[:, :, None, None].expand(B, Cin, 2, 2)
    conv_out = torch.nn.functional.conv2d(xa, conv.weight.detach().double().cpu(), conv.bias.detach().double().cpu(), padding=1) + \
        temb.double().cpu().expand(B, Cout)[:, :, None, None]
    if dtype == torch.bfloat16:
        conv_out = conv_out.to(torch.bfloat16).double()
    hn = torch.nn.functional.group_norm(conv_out, G, gamma.double().cpu(), beta.double().cpu(), 1e-5)
    Um = torch.from_numpy(np.asarray(idf.up_matrix(2, 2))).double()
    up = torch.einsum("ph,bchw,qw->bcpq", Um, hn, Um)
    want = torch.nn.functional.silu(up).mean((2, 3))
    assert rel_rms(got.float(), want.detach().numpy()) <= (1e-5 if dtype == torch.float32 else 1.5e-2)


def test_ffhq_forward_runs_the_const_form(monkeypatch):
    """The FFHQ forward takes the plane-constant form for all 14 3x3 convolutions of the 2x2 level (7 ResnetBlock2D)."""
    from afldm_amd import ops
    calls = []
    real = ops.af_act
    monkeypatch.setattr(ops, "af_act", lambda *a, **k: (calls.append(bool(k.get("out_const"))), real(*a, **k))[1])
    unet, _, _ = build_unet("ffhq", torch.bfloat16)
    x = torch.randn(2, 4, 32, 32, generator=torch.Generator().manual_seed(1)).cuda()
    y = unet(x, 500).sample
    assert torch.isfinite(y).all()
    assert sum(calls) >= 7, sum(calls)


# ----------------------------------------------------------------------------- (b) harness on the graph path
def _sampler_for(unet, steps):
    from afldm_amd.harness import CrossFrameSampler
    from afldm_amd.schedulers.ddim import ffhq_ddim_scheduler
    return CrossFrameSampler(unet, ffhq_ddim_scheduler(), steps)


@pytest.mark.parametrize("dtype,budget_db", [(torch.float32, 0.2), (torch.bfloat16, 1.0)])
def test_ffhq_equivariance_graph_path_vs_oracle(golden, dtype, budget_db):
    """test_ffhq_equivariance_vs_oracle (tests/test_gpu_r02.py) on the GRAPH path: the STORE pass and the shifted LOAD passes
    as replayed unrolled graphs with the stored pass's projected K / V (CrossFrameSampler), against the oracle's latents
    and its masked equivariance MSE (tests/golden/g13_r03.npz; reference shift_ldm_ffhq.py:124-151)."""
    from afldm_amd.pipelines.cross_frame_attn import get_unet_attn_processors, set_unet_attn_processor
    from afldm_amd.shift_utils.metrics import mask_mse
    from afldm_amd.shift_utils.shifters import ImageShifter
    g = golden("g13_r03.npz")
    x = torch.from_numpy(g["ffhq_x"]).cuda()
    unet, _, _ = build_unet("ffhq", dtype)
    smp = _sampler_for(unet, 4)
    prev = smp.install()
    try:
        base = smp.run(x, load=False)
        tol = 1e-3 if dtype == torch.float32 else 5e-2
        assert rel_rms(base.float(), g["ffhq_equiv_base"]) <= tol
        for k, tj in enumerate((0.375, 1.0)):
            xs, mask = ImageShifter("ideal_crop", 8).shift(x, 0, tj)
            ref, _ = ImageShifter("ideal_crop", 8).shift(base, 0, tj)
            den = smp.run(xs, load=True)
            assert rel_rms(den.float(), g[f"ffhq_equiv_lat_{k}"]) <= tol
            mse, want = float(mask_mse(den, ref, mask)), float(g[f"ffhq_equiv_mse_{k}"])
            db = 10 * np.log10(mse / want)
            print(f"[equivariance vs oracle, graph path] {dtype} tj={tj}: mask_mse {mse:.4e}, oracle {want:.4e} ({db:+.3f} dB)")
            assert abs(db) <= budget_db, (tj, mse, want)
        assert len(smp.engines) == 2                      # one STORE graph, one LOAD graph (replayed for the second offset)
        # a second STORE replay (new noise) refreshes what the LOAD graph reads: same as a fresh sampler would give
        x2 = torch.randn(1, 4, 32, 32, generator=torch.Generator().manual_seed(77)).cuda()
        base2 = smp.run(x2, load=False)
        den2 = smp.run(ImageShifter("ideal_crop", 8).shift(x2, 0, 0.375)[0], load=True)
        assert len(smp.engines) == 2
    finally:
        set_unet_attn_processor(unet, dict(prev))
    fresh = _sampler_for(unet, 4)
    prev = fresh.install()
    try:
        assert torch.equal(fresh.run(x2, load=False), base2)
        assert torch.equal(fresh.run(ImageShifter("ideal_crop", 8).shift(x2, 0, 0.375)[0], load=True), den2)
    finally:
        set_unet_attn_processor(unet, dict(prev))
    assert all(type(p).__name__ == "AttnProcessor2_0" for p in get_unet_attn_processors(unet).values())


@pytest.mark.parametrize("dtype,db_tol", [(torch.float32, 0.2), (torch.bfloat16, 1.0)])
def test_shift_equivariance_harness_graph_path(golden, dtype, db_tol):
    """test_shift_equivariance_harness (tests/test_gpu_unet.py: tiny model, oracle fixture g6) on the graph path, the LOAD
    passes both one by one and as one batch of two."""
    from test_gpu_unet import build
    from afldm_amd.pipelines.cross_frame_attn import set_unet_attn_processor
    from afldm_amd.shift_utils.metrics import mask_mse
    from afldm_amd.shift_utils.shifters import ImageShifter
    g = golden("g6_tiny_unet.npz")
    unet, cfg, _ = build("tiny", dtype)
    smp = _sampler_for(unet, 4)
    prev = smp.install()
    try:
        x = torch.from_numpy(g["x"])[:1].cuda()
        base = smp.run(x, load=False)
        assert rel_rms(base, g["equiv_base"]) <= (1e-3 if dtype == torch.float32 else 5e-2)
        shifter = ImageShifter("ideal_crop", 8)
        pairs = [shifter.shift(x, 0, tj) for tj in (0.375, 1.0)]
        both = smp.run(torch.cat([p[0] for p in pairs], 0), load=True)
        for k, tj in enumerate((0.375, 1.0)):
            xs, mask = pairs[k]
            den = smp.run(xs, load=True)
            ref, _ = ImageShifter("ideal_crop", 8).shift(base, 0, tj)
            want = float(g[f"equiv_mse_{k}"])
            for name, d in (("one by one", den), ("batched", both[k:k + 1])):
                mse = float(mask_mse(d, ref, mask))
                assert abs(10 * np.log10(mse / want)) <= db_tol, (name, tj, mse, want)
    finally:
        set_unet_attn_processor(unet, dict(prev))


def test_shift_ldm_graph_path_matches_eager_loop_and_replays():
    """afldm_amd.harness.shift_ldm: the default (graph) path against use_graph=False (the loop that follows reference
    shift_ldm_ffhq.py:85-108 statement by statement) on tiny models - frames and errors agree; a second call with another
    seed replays the cached graphs and still agrees with the eager loop; the timings dict is filled."""
    from test_gpu_vae import build_vae
    from afldm_amd.af_modules.af_api import make_af_unet
    from afldm_amd.harness import shift_ldm
    from afldm_amd.models.unet_2d import UNet2DModel
    from afldm_amd.pipelines.ldm_pipeline import MyLDMPipeline
    from afldm_amd.schedulers.ddim import ffhq_ddim_scheduler
    from oracle import configs as oc, unet as ou
    vae, _, _ = build_vae(torch.float32)
    ucfg = oc.tiny_unet()
    unet = UNet2DModel.from_config(ucfg)
    unet.load_state_dict(ou.randomize_norm_affine(ou.init_unet_params(ucfg, seed=0, conv_out_scale=0.1)))
    make_af_unet(unet)
    pipe = MyLDMPipeline(vae, unet.cuda(), ffhq_ddim_scheduler())
    for seed in (1, 2):
        tm = {}
        fr_g, er_g = shift_ldm(pipe, num_inference_steps=3, num_shift_steps=3, output_path=None,
                               generator=torch.Generator().manual_seed(seed), reference_exact=False, timings=tm)
        fr_e, er_e = shift_ldm(pipe, num_inference_steps=3, num_shift_steps=3, output_path=None,
                               generator=torch.Generator().manual_seed(seed), reference_exact=False, use_graph=False)
        assert len(fr_g) == 3 and tm["unet_s"] > 0 and tm["vae_s"] > 0 and tm["total_s"] >= tm["unet_s"] + tm["vae_s"]
        for a, b in zip(fr_g, fr_e):
            assert (a - b).abs().max() <= 2e-4
        assert np.allclose(er_g, er_e, rtol=2e-3, atol=1e-9), (er_g, er_e)
        smp = pipe._xframe_sampler
        assert sorted(smp.engines) == [(False, 1), (True, 3)]
    # new weights: the sampler notices and re-captures instead of replaying the old model
    with torch.no_grad():
        pipe.unet.conv_in.weight.mul_(1.5)
    fr_g, er_g = shift_ldm(pipe, num_inference_steps=3, num_shift_steps=3, output_path=None,
                           generator=torch.Generator().manual_seed(2), reference_exact=False)
    fr_e, er_e = shift_ldm(pipe, num_inference_steps=3, num_shift_steps=3, output_path=None,
                           generator=torch.Generator().manual_seed(2), reference_exact=False, use_graph=False)
    for a, b in zip(fr_g, fr_e):
        assert (a - b).abs().max() <= 2e-4


# ----------------------------------------------------------------------------- (c) the I2SB ODE bridge on replayed graphs
@pytest.mark.parametrize("dtype,tol", [(torch.float32, 2e-6), (torch.bfloat16, 2e-3)])
def test_i2sb_ode_bridge_graph_path_matches_eager_loop(golden, dtype, tol):
    """I2SBLDMPipeline._bridge (deterministic, unclipped, latent carried in fp32: what scripts/shift_ldm_sr.py runs; reference
    i2sb_pipeline.py:48-56 + i2sb_scheduler.py:382-459) on the captured-graph engine against its own eager loop - same kernels,
    same coefficients - at FFHQ size, 12 evaluations, batch 2; the full-length oracle comparisons of test_gpu_r02 / r04 / r05 run
    through the graph path by default."""
    from afldm_amd.configs import FFHQ_DDIM_CONFIG
    from afldm_amd.pipelines.i2sb_pipeline import I2SBLDMPipeline
    from afldm_amd.schedulers.i2sb import I2SBScheduler
    g = golden("g14_r03.npz")
    cfg = {k: v for k, v in FFHQ_DDIM_CONFIG.items() if k != "set_alpha_to_one"}
    unet, _, _ = build_unet("ffhq", dtype)
    pipe = I2SBLDMPipeline(None, unet, I2SBScheduler.from_config(cfg))
    pipe.set_progress_bar_config(disable=True)
    start = torch.from_numpy(g["i2sb_start"]).cuda().to(dtype)
    start = torch.cat([start, start.flip(-1)], 0)
    a = pipe._bridge(start, 13, True, None)
    b = pipe._bridge(start, 13, True, None, use_graph=False)
    assert "_ode_engines" in pipe.__dict__ and len(pipe._ode_engines) == 1
    assert rel_rms(a.float(), b.float().cpu().numpy()) <= tol
    assert torch.equal(a, pipe._bridge(start, 13, True, None))                 # replay: bit-identical
    # the clipped configuration (the reference scheduler's default) keeps the eager loop
    clip = I2SBLDMPipeline(None, unet, I2SBScheduler.from_config(dict(cfg, clip_sample=True)))
    clip.set_progress_bar_config(disable=True)
    assert clip.scheduler.ode_schedule(13) is None
    assert torch.isfinite(clip._bridge(start, 4, True, None)).all() and "_ode_engines" not in clip.__dict__


def test_shift_ldm_sr_graph_path_matches_eager_loop():
    """harness.shift_ldm_sr (reference scripts/shift_ldm_sr.py:43-150) on replayed graphs against use_graph=False, tiny models."""
    from afldm_amd.af_modules.af_api import make_af_unet, make_af_vae_from_config
    from afldm_amd.configs import tiny_unet_config
    from afldm_amd.harness import shift_ldm_sr
    from afldm_amd.models.unet_2d import UNet2DModel
    from afldm_amd.models.vae import AutoencoderKL
    from afldm_amd.pipelines.i2sb_pipeline import I2SBLDMPipeline
    from afldm_amd.schedulers.i2sb import I2SBScheduler
    torch.manual_seed(0)
    unet = UNet2DModel.from_config(tiny_unet_config(sample_size=8))
    vae = AutoencoderKL(in_channels=3, out_channels=3, down_block_types=["DownEncoderBlock2D"] * 4,
                        up_block_types=["UpDecoderBlock2D"] * 4, block_out_channels=[32, 32, 64, 64],
                        layers_per_block=1, latent_channels=4, norm_num_groups=8, scaling_factor=0.6, mid_act=True,
                        down_filtered_act=[False, True, True, True], up_filtered_act=[True, True, True, False],
                        up_rescale=[True, True, True])
    pipe = I2SBLDMPipeline(vae, unet, I2SBScheduler(clip_sample=False)).to("cuda")
    make_af_unet(pipe.unet)
    make_af_vae_from_config(pipe.vae)
    img = torch.rand(1, 3, 64, 64, generator=torch.Generator().manual_seed(5)) * 2 - 1
    fg, eg = shift_ldm_sr(pipe, num_inference_steps=5, num_shift_steps=3, output_path=None, image=img)
    fe, ee = shift_ldm_sr(pipe, num_inference_steps=5, num_shift_steps=3, output_path=None, image=img, use_graph=False)
    assert sorted(pipe._xframe_sampler.engines) == [(False, 1), (True, 3)]
    for x, y in zip(fg, fe):
        assert (x - y).abs().max() <= 5e-4
    assert np.allclose(eg, ee, rtol=5e-3, atol=1e-9), (eg, ee)


def test_ddim_inversion_graph_path_and_harness_with_input_image(tmp_path):
    """MyLDMPipeline.ddim_inversion (reference ldm_pipeline.py:133-160) on the captured-graph engine against its eager loop (fp32
    latents; the oracle comparisons of test_gpu_r02 run through the graph path by default), and harness.shift_ldm with an input
    IMAGE (VAE-encode -> inversion -> STORE -> LOAD; reference shift_ldm_ffhq.py:110-116) on graphs against use_graph=False."""
    from PIL import Image
    from test_gpu_vae import build_vae
    from afldm_amd.af_modules.af_api import make_af_unet
    from afldm_amd.harness import shift_ldm
    from afldm_amd.models.unet_2d import UNet2DModel
    from afldm_amd.pipelines.ldm_pipeline import MyLDMPipeline
    from afldm_amd.schedulers.ddim import ffhq_ddim_scheduler
    from oracle import configs as oc, unet as ou
    vae, _, _ = build_vae(torch.float32)
    ucfg = oc.tiny_unet()
    unet = UNet2DModel.from_config(ucfg)
    unet.load_state_dict(ou.randomize_norm_affine(ou.init_unet_params(ucfg, seed=0, conv_out_scale=0.1)))
    make_af_unet(unet)
    pipe = MyLDMPipeline(vae, unet.cuda(), ffhq_ddim_scheduler())
    pipe.set_progress_bar_config(disable=True)
    s = ucfg["sample_size"]
    x = torch.randn(2, 4, s, s, generator=torch.Generator().manual_seed(9)).cuda()
    pipe.scheduler.set_timesteps(5, device="cuda")
    a = pipe.ddim_inversion(x, bar=False)
    b = pipe.ddim_inversion(x, bar=False, use_graph=False)
    assert "_inv_engines" in pipe.__dict__ and rel_rms(a, b.cpu().numpy()) <= 2e-6
    assert torch.equal(a, pipe.ddim_inversion(x, bar=False))
    rng = np.random.default_rng(0)
    path = str(tmp_path / "in.png")
    Image.fromarray((rng.random((96, 96, 3)) * 255).astype(np.uint8)).save(path)
    torch.manual_seed(123)           # (vae_encode draws the posterior sample from the global generator, as the reference does)
    fg, eg = shift_ldm(pipe, num_inference_steps=3, num_shift_steps=2, output_path=None, input_path=path, reference_exact=False)
    torch.manual_seed(123)
    fe, ee = shift_ldm(pipe, num_inference_steps=3, num_shift_steps=2, output_path=None, input_path=path, reference_exact=False,
                       use_graph=False)
    for u, v in zip(fg, fe):
        assert (u - v).abs().max() <= 5e-4
    assert np.allclose(eg, ee, rtol=5e-3, atol=1e-9), (eg, ee)
