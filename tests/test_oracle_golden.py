"""The oracle's alias-free pieces vs fixtures recorded from the IMPORTED reference
(oracle/gen_golden.py part A), plus known-answer values for the scheduler / masks
(SURVEY.md Appendix B/C).  CPU only."""
import numpy as np
import pytest
import torch
import torch.nn.functional as F

from oracle import ddim, ideal_filters as idf, shift, unet


def t(a):
    return torch.from_numpy(np.asarray(a))


def test_masks_vs_reference(golden):
    g = golden("g1_masks.npz")
    n = 0
    for key in g.files:
        kind, _, N, c = key.split("_")
        N = int(N)
        cutoff = 0.5 if c == "h" else 0.125
        fn1 = idf.lpf_rect_1d if kind == "lpf" else idf.recon_rect_1d
        fn2 = idf.lpf_rect_2d if kind == "lpf" else idf.recon_rect_2d
        ref = g[key]
        mine = fn2(N, cutoff).numpy() if ref.ndim == 2 else fn1(N, cutoff).numpy()
        assert np.array_equal(mine, ref), key
        n += 1
    assert n >= 28


def test_mask_known_answers():
    # SURVEY.md Appendix B
    assert idf.lpf_rect_1d(4).tolist() == [1, 0, 0, 0]
    assert idf.recon_rect_1d(4).tolist() == [1, .5, 0, .5]
    assert idf.lpf_rect_1d(8).tolist() == [1, 1, 0, 0, 0, 0, 0, 1]
    assert idf.recon_rect_1d(8).tolist() == [1, 1, .5, 0, 0, 0, .5, 1]
    assert idf.lpf_rect_1d(6).tolist() == idf.recon_rect_1d(6).tolist() == [1, 1, 0, 0, 0, 1]
    assert float(idf.lpf_rect_1d(32).sum()) == 15 and float(idf.recon_rect_1d(32).sum()) == 16
    assert float(idf.lpf_rect_1d(64).sum()) == 31 and float(idf.recon_rect_1d(64).sum()) == 32
    r = idf.recon_rect_1d(256, 1 / 8)
    assert float(r.sum()) == 32 and r[16] == .5 and r[240] == .5 and r[15] == 1 and r[17] == 0


@pytest.mark.parametrize("N", [2, 4, 8, 16, 32])
def test_filters_bitexact_vs_reference(golden, N):
    g = golden("g2_filters.npz")
    x = t(g[f"x_{N}"])
    assert torch.equal(idf.lpf_rfft(x.clone()), t(g[f"lpf_{N}"]))
    assert torch.equal(idf.upsample_rfft(x.clone(), 2), t(g[f"up2_{N}"]))
    if N <= 16:
        assert torch.equal(idf.upsample_rfft(x.clone(), 8), t(g[f"up8_{N}"]))
    assert torch.equal(idf.subpixel_shift(x.clone(), 2, 1, 1), t(g[f"subpix_{N}"]))
    assert torch.equal(idf.warped_nonlinearity(t(g[f"wx_{N}"])), t(g[f"wy_{N}"]))


@pytest.mark.parametrize("N", [2, 4, 8, 16, 32])
def test_dense_matrix_form_equals_fft_form(golden, N):
    """U X U^T and D silu(.) D^T (fp64 matrices) == the reference FFT path to fp32 rounding.
    This is the identity the HIP kernels are built on."""
    g = golden("g2_filters.npz")
    x = g[f"x_{N}"].astype(np.float64)
    U = idf.up_matrix(N, 2)
    assert np.allclose(U[::2], np.eye(N), atol=1e-12)          # even phases = identity
    assert np.allclose(U.sum(1), 1, atol=1e-12)
    up = np.einsum("ah,nchw,bw->ncab", U, x, U)
    assert np.abs(up - g[f"up2_{N}"]).max() < 2e-5
    D = idf.down_matrix(2 * N)
    assert np.allclose(D.sum(1), 1, atol=1e-12)
    wx = g[f"wx_{N}"].astype(np.float64)
    z = np.einsum("ah,nchw,bw->ncab", U, wx, U)
    z = z / (1 + np.exp(-z))
    y = np.einsum("ah,nchw,bw->ncab", D, z, D)
    assert np.abs(y - g[f"wy_{N}"]).max() < 2e-5
    if N >= 4:
        Dn = idf.down_matrix(N)
        lp = np.einsum("ah,nchw,bw->ncab", Dn, x, Dn)
        assert np.abs(lp - g[f"lpf_{N}"][:, :, ::2, ::2]).max() < 2e-5
    if N <= 16:
        U8 = idf.up_matrix(N, 8)
        assert np.abs(np.einsum("ah,nchw,bw->ncab", U8, x, U8) - g[f"up8_{N}"]).max() < 5e-5


def test_n2_plane_degenerates_to_mean(golden):
    g = golden("g2_filters.npz")
    y = g["wy_2"]
    assert np.abs(y - y.mean(axis=(2, 3), keepdims=True)).max() < 1e-6


@pytest.mark.parametrize("N", [8, 16])
def test_af_resample_vs_reference(golden, N):
    g = golden("g4_af_resample.npz")
    x, w, b = t(g[f"x_{N}"]), t(g[f"w_{N}"]), t(g[f"b_{N}"])
    assert torch.allclose(idf.af_downsample(x, w, b, padding=1), t(g[f"down_{N}"]), atol=1e-6)
    assert torch.allclose(idf.af_upsample(x, w, b), t(g[f"up_{N}"]), atol=1e-6)


def test_shifters_and_metrics_vs_reference(golden):
    g = golden("g5_shift_metrics.npz")
    lat, img = t(g["lat"]), t(g["img"])
    for k, tj in enumerate((0.125, 0.5, 1.0, 2.0)):
        for mode in ("ideal", "ideal_crop"):
            w, m = shift.shift_ideal(lat, 0, tj, 8, crop=(mode == "ideal_crop"))
            assert torch.equal(w, t(g[f"{mode}_{k}"])), (mode, tj)
            assert torch.equal(m, t(g[f"{mode}_mask_{k}"]))
        w, m = shift.shift_bilinear(img, 0, tj * 8)
        assert torch.allclose(w, t(g[f"bilinear_{k}"]), atol=1e-6)
        assert torch.equal(m, t(g[f"bilinear_mask_{k}"]))
    w, m = shift.shift_ideal(lat, 0.375, -0.625, 8, crop=True)
    assert torch.equal(w, t(g["ideal_crop_2d"])) and torch.equal(m, t(g["ideal_crop_2d_mask"]))
    for k, (ti, tj) in enumerate(((1.5, -2.25), (-0.5, 0.0), (0.0, 3.0))):
        assert torch.equal(shift.gen_valid_mask((1, 1, 8, 8), ti, tj), t(g[f"valid_mask_{k}"]))
    a, b, m = t(g["ma"]), t(g["mb"]), t(g["mm"])
    assert torch.equal(shift.mask_mse(a, b, m), t(g["mask_mse"]))
    assert torch.equal(shift.mask_psnr(a, b, m), t(g["mask_psnr"]))
    assert torch.equal(shift.psnr(a, b), t(g["psnr"]))


def test_scheduler_known_answers():
    # SURVEY.md Appendix C (fp32)
    s = ddim.DDIM()
    s.set_timesteps(50)
    ts = s.timesteps.tolist()
    assert ts[:3] == [981, 961, 941] and ts[-3:] == [41, 21, 1] and len(ts) == 50
    ac = s.alphas_cumprod
    for idx, val in ((0, 0.998499990), (1, 0.996994436), (21, 0.965732038),
                     (981, 0.000201956), (999, 0.000142304)):
        assert abs(float(ac[idx]) - val) < 5e-9 + 2e-6 * val, (idx, float(ac[idx]))
    one, half = torch.tensor(1.0), torch.tensor(0.5)
    assert abs(float(s.step(half, 981, one)) - 1.1040677) < 2e-6
    assert abs(float(s.step(half, 1, one)) - 0.9926875) < 2e-6
    e = unet.timestep_embedding(torch.tensor([981]), 192)[0]
    assert abs(float(e[0]) - 0.6799572) < 1e-5       # cos[0]
    assert abs(float(e[95]) - 0.9941760) < 1e-5      # cos[95]
    assert abs(float(e[96]) - 0.7332518) < 1e-5      # sin[0]


def test_sr4x_degrade_matches_reference_fixture(golden):
    """oracle.superresolution.build_sr4x vs outputs of the imported reference build_sr4x
    (afldm/af_libs/superresolution.py:288-320; fixtures: oracle/gen_golden.py part d)."""
    from oracle.superresolution import build_sr4x
    g = golden("g9_sr4x.npz")
    x = torch.from_numpy(g["x64"])
    for flt in ("bicubic", "pool"):
        y = build_sr4x(flt, 64)(x)
        assert (y - torch.from_numpy(g[f"y64_{flt}"])).abs().max() < 1e-6, flt
    x256 = torch.from_numpy(g["x256"].astype(np.float32))
    y = build_sr4x("bicubic", 256)(x256)
    assert (y[:, :, 96:160, 96:160] - torch.from_numpy(g["y256_bicubic_crop"])).abs().max() < 1e-6
    assert abs(float(y.double().sum()) - g["y256_bicubic_sum"][0]) < 1e-2
    assert abs(float((y.double() ** 2).sum()) - g["y256_bicubic_sum"][1]) < 1e-2
    y3 = build_sr4x("bicubic", 64)(x[0])                      # 3-D input keeps its rank
    assert y3.shape == (3, 64, 64)



UPFIRDN_CASES = [  # name, filter, up, down, padding, flip, gain  (oracle/gen_golden.py part e)
    ("c0", "f2", 1, 1, 0, False, 1),
    ("c1", "f2", 2, 1, [2, 1, 3, 0], False, 4),
    ("c2", "f2", [1, 3], [2, 1], [1, 2, 2, 2], True, 0.5),
    ("c3", "f1", 2, 2, [3, 3, 2, 2], False, 2),
    ("c4", "f1", 1, 3, [-1, 2, 0, -2], True, 1),
    ("c5", "f2", 3, 2, [-2, 4, 5, -3], False, 1.5),
    ("c6", None, 2, 1, 0, False, 1),
]
FRAC_SHIFTS = 5
SHIFT_T = [(0.125, 0.5), (1.0, -2.375), (-3.5, 0.25)]


def test_upfirdn2d_oracle_vs_reference(golden):
    """oracle/upfirdn.py against outputs of the imported reference's upfirdn2d (ref path)."""
    from oracle import upfirdn as ou
    g = golden("g10_upfirdn.npz")
    x = t(g["x"])
    for name, fn, up, down, pad, flip, gain in UPFIRDN_CASES:
        f = None if fn is None else t(g[fn])
        mine = ou.upfirdn2d(x, f, up=up, down=down, padding=pad, flip_filter=flip, gain=gain)
        assert mine.shape == g[name].shape, name
        torch.testing.assert_close(mine, t(g[name]), rtol=0, atol=0, msg=name)
    fs = ou.setup_filter([1, 3, 3, 1])
    assert np.array_equal(fs.numpy(), g["setup_1331"])
    assert np.array_equal(ou.setup_filter([1, 2, 3, 4, 4, 3, 2, 1], gain=2).numpy(), g["setup_sep"])
    assert np.array_equal(ou.filter2d(x, fs, padding=1).numpy(), g["filter2d"])
    assert np.array_equal(ou.upsample2d(x, fs, up=2).numpy(), g["upsample2d"])
    assert np.array_equal(ou.downsample2d(x, fs, down=2).numpy(), g["downsample2d"])


def test_translations_oracle_vs_reference(golden):
    from oracle import upfirdn as ou
    g = golden("g10_upfirdn.npz")
    img = t(g["img"])
    for k in range(FRAC_SHIFTS):
        tx, ty = (float(v) for v in g[f"frac{k}_t"])
        z, m = ou.apply_fractional_translation(img, tx, ty)
        assert np.array_equal(z.numpy(), g[f"frac{k}_z"]) and np.array_equal(m.numpy(), g[f"frac{k}_m"]), k
        z, m = ou.apply_integer_translation(img, tx, ty)
        assert np.array_equal(z.numpy(), g[f"int{k}_z"]) and np.array_equal(m.numpy(), g[f"int{k}_m"]), k
    for k, (ti, tj) in enumerate(SHIFT_T):
        w = ou.fourier_shift_batch(img, ti, tj)
        assert np.array_equal(w.numpy(), g[f"shift_fourier{k}_w"]), k
        assert np.array_equal(g[f"shift_fourier{k}_m"], np.ones_like(g[f"shift_fourier{k}_w"]))
        mc = shift.gen_valid_mask(w.shape, ti, tj)
        assert np.array_equal(mc.numpy(), g[f"shift_fourier_crop{k}_m"])
        assert np.array_equal((w * mc).numpy(), g[f"shift_fourier_crop{k}_w"])
        z, m = ou.apply_fractional_translation(img, tj / 32, ti / 32)     # shifters.py:158-161
        assert np.array_equal(z.numpy(), g[f"shift_lanczos{k}_w"])
        assert np.array_equal(m[:, 0:1].numpy(), g[f"shift_lanczos{k}_m"])


def test_image_samplers_and_general_cutoff_oracle_vs_reference(golden):
    from oracle import upfirdn as ou
    g = golden("g10_upfirdn.npz")
    img, img30 = t(g["img"]), t(g["img30"])
    for mode in ("blur", "ideal", "nearest", "bilinear"):
        assert np.array_equal(ou.image_upsample(img, 2, mode).numpy(), g[f"up_{mode}"]), mode
        if mode in ("blur", "ideal"):
            assert np.array_equal(ou.image_low_pass(img, 2, mode).numpy(), g[f"lowpass_{mode}"]), mode
    assert int(g["down_blur_raises"]) == 1
    with pytest.raises(RuntimeError):
        ou.image_downsample(img, 2, "blur")
    with pytest.raises(IndexError):       # LPF_RFFT(cutoff=2) mask on a plane with N % 4 == 0
        ou.image_downsample(img, 2, "ideal")
    for mode in ("ideal", "nearest", "bilinear"):
        src = img30 if mode == "ideal" else img
        assert np.array_equal(ou.image_downsample(src, 2, mode).numpy(), g[f"down_{mode}"]), mode
    xz = t(g["xz"])
    assert int(g["lpf_fft_raises"]) == 1
    assert np.array_equal(idf.lpf_rfft(xz.clone(), 0.25).numpy(), g["lpf_q"])
    assert np.array_equal(idf.lpf_recon_rfft(xz.clone(), 0.5).numpy(), g["recon_h"])
    assert np.array_equal(idf.lpf_recon_rfft(xz.clone(), 0.25).numpy(), g["recon_q"])
    np.testing.assert_allclose(idf.lpf_recon_rfft(xz.clone(), 0.5).numpy(), g["recon_fft"], rtol=0, atol=2e-6)
    assert np.array_equal(idf.upsample_rfft(xz[:, :, :12, :12].clone(), 4).numpy(), g["up4"])
    assert np.array_equal(idf.upsample_rfft(xz[:, :, :12, :12].clone(), 2, factor=0.5).numpy(), g["up2_f2"])


# ----------------------------------------------------------------------------- round 4: pins of reference-resident files
def test_i2sb_oracle_vs_reference_file(golden):
    """oracle/i2sb.py against fixtures recorded by importing /root/reference/afldm/schedulers/i2sb_scheduler.py under a
    plumbing-only diffusers stub (oracle/gen_golden.py part i): tables, timesteps, ODE and seeded stochastic steps with
    clip_sample off / on, add_noise, compute_label - bit-equal."""
    from oracle.i2sb import I2SB
    g = golden("g15_r04_refpins.npz")
    o = I2SB()
    for name in ("betas", "std_fwd", "std_bwd", "std_sb", "mu_x0", "mu_x1"):
        assert torch.equal(getattr(o, name), t(g[f"i2sb_{name}"])), name
    for n in (50, 100):
        o.set_timesteps(n)
        assert torch.equal(o.timesteps, t(g[f"i2sb_timesteps_{n}"]))
    x, e = t(g["i2sb_step_x"]), t(g["i2sb_step_eps"])

    def same(a, b):       # bit-equal, NaN == NaN: at t = 1 of the 100-step schedule prev_t = -9 indexes std_fwd from the
        return torch.equal(torch.nan_to_num(a, nan=12345.0), torch.nan_to_num(b, nan=12345.0))   # END of the table and the
    #                       reference itself returns NaN (std_delta = sqrt(negative)); its pipeline never takes that step
    #                       (i2sb_pipeline.py:48 iterates timesteps[:-1]) - the oracle must reproduce that, too
    assert np.isnan(g["i2sb_step_ode_noclip_1"]).all()
    for clip, tag in ((False, "noclip"), (True, "clip")):
        o = I2SB(clip_sample=clip)
        o.set_timesteps(100)
        for ts in (991, 501, 11, 1):
            prev, x0 = o.step(e, ts, x, is_ode=True, return_x0=True)
            assert same(prev, t(g[f"i2sb_step_ode_{tag}_{ts}"])), (tag, ts)
            assert torch.equal(x0, t(g[f"i2sb_step_x0_{tag}_{ts}"]))
            sde = o.step(e, ts, x, is_ode=False, generator=torch.Generator().manual_seed(1000 + ts))
            assert same(sde, t(g[f"i2sb_step_sde_{tag}_{ts}"])), (tag, ts)
            assert ts == 1 or not torch.equal(sde, prev)
            assert o.previous_timestep(ts) == int(g[f"i2sb_prev_t_{ts}"])
    # the clamp must have been exercised by the fixture (|x0| > 1 somewhere without it)
    assert np.abs(g["i2sb_step_x0_noclip_991"]).max() > 1.0 and np.abs(g["i2sb_step_x0_clip_991"]).max() <= 1.0
    o = I2SB()
    x0, x1, nz, ts = t(g["i2sb_x0"]), t(g["i2sb_x1"]), t(g["i2sb_noise"]), t(g["i2sb_ts"])
    assert torch.equal(o.add_noise(x0, x1, ts, is_ode=True), t(g["i2sb_add_noise_ode"]))
    xt = o.add_noise(x0, x1, ts, noise=nz, is_ode=False)
    assert torch.equal(xt, t(g["i2sb_add_noise_sde"]))
    assert torch.equal(o.compute_label(ts, x0, xt), t(g["i2sb_label"]))


def test_product_i2sb_host_tables_vs_reference_file(golden):
    """The product scheduler's host-side tables / timesteps / add_noise / compute_label are the reference file's, bit for
    bit (its `step` runs on the GPU: tests/test_gpu_r04.py)."""
    from afldm_amd.configs import FFHQ_DDIM_CONFIG
    from afldm_amd.schedulers.i2sb import I2SBScheduler
    g = golden("g15_r04_refpins.npz")
    s = I2SBScheduler.from_config({k: v for k, v in FFHQ_DDIM_CONFIG.items() if k != "set_alpha_to_one"})
    for name in ("betas", "std_fwd", "std_bwd", "std_sb", "mu_x0", "mu_x1"):
        assert torch.equal(getattr(s, name), t(g[f"i2sb_{name}"])), name
    for n in (50, 100):
        s.set_timesteps(n)
        assert torch.equal(s.timesteps, t(g[f"i2sb_timesteps_{n}"]))
    for ts in (991, 501, 11, 1):
        assert s.previous_timestep(ts) == int(g[f"i2sb_prev_t_{ts}"])
    x0, x1, nz, ts = t(g["i2sb_x0"]), t(g["i2sb_x1"]), t(g["i2sb_noise"]), t(g["i2sb_ts"])
    assert torch.equal(s.add_noise(x0, x1, ts, is_ode=True), t(g["i2sb_add_noise_ode"]))
    xt = s.add_noise(x0, x1, ts, is_ode=False, noise=nz)
    assert torch.equal(xt, t(g["i2sb_add_noise_sde"]))
    assert torch.equal(s.compute_label(ts, x0, xt), t(g["i2sb_label"]))


@pytest.mark.parametrize("N", [8, 16])
def test_af_block_forward_bodies_vs_reference_file(golden, N):
    """oracle.ideal_filters.af_downsample / af_upsample / warped_nonlinearity against the forward() bodies of
    /root/reference/afldm/af_modules/af_blocks.py run under stub base classes (part i): padding 1 and the padding == 0
    branch (explicit (1,1,1,1) pad, af_blocks.py:143-145), the bf16 -> fp32 cast around the resampler (:80-96)."""
    g = golden("g15_r04_refpins.npz")
    x = t(g[f"afb_x_{N}"])
    for pad in (1, 0):
        got = idf.af_downsample(x, t(g[f"afb_down_w_{N}_{pad}"]), t(g[f"afb_down_b_{N}_{pad}"]), padding=pad)
        ref = t(g[f"afb_down_{N}_{pad}"])
        assert got.shape == ref.shape == (2, x.shape[1], N // 2, N // 2)
        assert torch.equal(got, ref), pad
    w, b = t(g[f"afb_up_w_{N}"]), t(g[f"afb_up_b_{N}"])
    assert torch.equal(idf.af_upsample(x, w, b), t(g[f"afb_up_{N}"]))
    # bf16: resample in fp32, round, convolve in bf16
    xb = x.to(torch.bfloat16)
    up = idf.upsample_rfft(xb.float(), 2).to(torch.bfloat16)
    got = F.conv2d(up, w.to(torch.bfloat16), b.to(torch.bfloat16), padding=1).float()
    assert torch.equal(got, t(g[f"afb_up_bf16_{N}"]))


def test_warped_nonlinearity_module_vs_reference_file(golden):
    """WarpedNonlinearity.forward (af_blocks.py:19-28): ndim < 4 -> the plain nonlinearity; any wrapped module works."""
    g = golden("g15_r04_refpins.npz")
    assert torch.equal(idf.warped_nonlinearity(t(g["afb_wn_2d_in"])), t(g["afb_wn_2d_out"]))
    x = t(g["afb_wn_4d_in"])
    assert torch.equal(idf.warped_nonlinearity(x), t(g["afb_wn_4d_out"]))
    assert torch.equal(idf.warped_nonlinearity(x, torch.tanh), t(g["afb_wn_tanh_out"]))


def test_cross_frame_control_flow_vs_reference_file(golden):
    """oracle.unet.attention_block's STORE / LOAD / batch-repeat / enable_interp logic against
    /root/reference/afldm/pipelines/cross_frame_attn.py:66-130 driven on the same layers (part i; the base
    AttnProcessor2_0 both sides call is the oracle's diffusers restatement, which stays unpinned)."""
    g = golden("g15_r04_refpins.npz")
    sd = {"a." + k[len("cfa_sd_"):]: t(g[k]) for k in g.files if k.startswith("cfa_sd_")}
    cfg = dict(attention_head_dim=16, norm_num_groups=8, norm_eps=1e-5)
    cache = unet.AttnCache(enable_interp=True)
    c = unet._Ctx(sd, cfg, True, cache, None)
    xa, xb, xq = t(g["cfa_xa"]), t(g["cfa_xb"]), t(g["cfa_xq"])
    cache.state, cache.timestep = unet.AttnCache.STORE, 7
    assert torch.equal(unet.attention_block(c, "a", xa), t(g["cfa_store0"]))
    cache.store_id = 1
    assert torch.equal(unet.attention_block(c, "a", xb), t(g["cfa_store1"]))
    cache.state, cache.alpha = unet.AttnCache.LOAD, 0.3
    assert torch.equal(unet.attention_block(c, "a", xq), t(g["cfa_load_interp"]))
    cache.enable_interp = False
    assert torch.equal(unet.attention_block(c, "a", xq), t(g["cfa_load"]))
    cache.state = unet.AttnCache.IDLE
    assert torch.equal(unet.attention_block(c, "a", xq), t(g["cfa_idle"]))
    assert not torch.equal(t(g["cfa_load"]), t(g["cfa_idle"]))
