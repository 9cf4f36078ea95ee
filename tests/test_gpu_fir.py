"""upfirdn2d and what the reference builds on it on MI355X: the HIP kernel
through the C ABI against the oracle (oracle/upfirdn.py, bit-exact against the imported reference,
tests/test_oracle_golden.py) and against fixtures recorded from the reference itself
(tests/golden/g10_upfirdn.npz, g5_shift_metrics.npz): Lanczos / integer translations, the lanczos /
fourier / bilinear ImageShifter modes, blur / ideal image samplers and general-cutoff ideal filters.

Tolerances: fp32 max-abs <= 2e-5 * max|ref| per op - the FIR kernels themselves agree
to a few ulp; bf16 I/O rel-RMS <= 1e-2; masks exact."""
import numpy as np
import pytest
import torch

from test_oracle_golden import FRAC_SHIFTS, SHIFT_T, UPFIRDN_CASES

pytestmark = pytest.mark.gpu


def t(a):
    return torch.from_numpy(np.asarray(a))


def close32(got, ref, what, tol=2e-5):
    got, ref = got.float().cpu(), torch.as_tensor(ref).float()
    assert got.shape == ref.shape, (what, got.shape, ref.shape)
    assert torch.isfinite(got).all(), what
    err = float((got - ref).abs().max())
    assert err <= tol * max(float(ref.abs().max()), 1e-3), (what, err)


def rel_rms(got, ref):
    got, ref = got.double().cpu(), torch.as_tensor(ref).double()
    return float((got - ref).pow(2).mean().sqrt() / ref.pow(2).mean().sqrt().clamp_min(1e-12))


def _up():
    from afldm_amd.af_libs.torch_utils.ops import upfirdn2d
    return upfirdn2d


def test_upfirdn2d_vs_reference_fixture(golden):
    up = _up()
    g = golden("g10_upfirdn.npz")
    x = t(g["x"]).cuda()
    for name, fn, u, d, pad, flip, gain in UPFIRDN_CASES:
        f = None if fn is None else t(g[fn]).cuda()
        y = up.upfirdn2d(x, f, up=u, down=d, padding=pad, flip_filter=flip, gain=gain)
        close32(y, g[name], name, tol=3e-6)
    fs = up.setup_filter([1, 3, 3, 1], device="cuda")
    assert np.array_equal(fs.cpu().numpy(), g["setup_1331"])
    assert np.array_equal(up.setup_filter([1, 2, 3, 4, 4, 3, 2, 1], gain=2).numpy(), g["setup_sep"])
    close32(up.filter2d(x, fs, padding=1), g["filter2d"], "filter2d", 3e-6)
    close32(up.upsample2d(x, fs, up=2), g["upsample2d"], "upsample2d", 3e-6)
    close32(up.downsample2d(x, fs, down=2), g["downsample2d"], "downsample2d", 3e-6)


@pytest.mark.parametrize("seed", range(12))
def test_upfirdn2d_random_configs_vs_oracle(seed):
    """Random up / down / padding (incl. crops) / filter shapes, 2-D and separable, both dtypes."""
    from oracle import upfirdn as ou
    up = _up()
    rng = np.random.default_rng(seed)
    g = torch.Generator().manual_seed(seed)
    B, C = int(rng.integers(1, 4)), int(rng.integers(1, 6))
    H, W = int(rng.integers(5, 70)), int(rng.integers(5, 90))
    ux, uy, dx, dy = (int(v) for v in rng.integers(1, 5, size=4))
    sep = bool(rng.integers(0, 2))
    fh, fw = int(rng.integers(1, 9)), int(rng.integers(1, 9))
    f = torch.randn(fh if sep else fh, generator=g) if sep else torch.randn(fh, fw, generator=g)
    if sep:
        fw = fh
    pad = [int(v) for v in rng.integers(-3, 7, size=4)]
    if W * ux + pad[0] + pad[1] < fw or H * uy + pad[2] + pad[3] < fh:
        pad = [abs(p) + 4 for p in pad]
    flip, gain = bool(rng.integers(0, 2)), float(rng.choice([1.0, 0.5, 4.0]))
    x = torch.randn(B, C, H, W, generator=g)
    ref = ou.upfirdn2d(x, f, up=[ux, uy], down=[dx, dy], padding=pad, flip_filter=flip, gain=gain)
    y = up.upfirdn2d(x.cuda(), f.cuda(), up=[ux, uy], down=[dx, dy], padding=pad, flip_filter=flip, gain=gain)
    close32(y, ref, f"seed {seed}", tol=5e-6)
    xb = x.to(torch.bfloat16)
    refb = ou.upfirdn2d(xb.float(), f, up=[ux, uy], down=[dx, dy], padding=pad, flip_filter=flip, gain=gain)
    yb = up.upfirdn2d(xb.cuda(), f.cuda(), up=[ux, uy], down=[dx, dy], padding=pad, flip_filter=flip, gain=gain)
    assert yb.dtype == torch.bfloat16 and rel_rms(yb, refb) <= 1e-2


def test_upfirdn2d_argument_errors():
    up = _up()
    x = torch.randn(1, 1, 4, 4, device="cuda")
    with pytest.raises(AssertionError):                      # plane smaller than the filter (reference :159-162)
        up.upfirdn2d(x, torch.ones(7, 7, device="cuda"))
    with pytest.raises(AssertionError):
        up.upfirdn2d(x, torch.ones(2, 2, device="cuda"), up=0)
    with pytest.raises(AssertionError):
        up.upfirdn2d(x, torch.ones(2, 2, dtype=torch.float64, device="cuda"))
    with pytest.raises(RuntimeError):
        up.upfirdn2d(x.cpu(), torch.ones(2, 2))
    y = up.upfirdn2d(x, None, up=2)                          # identity filter: zero stuffing
    assert torch.equal(y[:, :, ::2, ::2], x) and float(y[:, :, 1::2].abs().max()) == 0


def test_translations_vs_reference_fixture(golden):
    from afldm_amd.af_libs import equivariance as eq
    g = golden("g10_upfirdn.npz")
    img = t(g["img"]).cuda()
    for k in range(FRAC_SHIFTS):
        tx, ty = (float(v) for v in g[f"frac{k}_t"])
        z, m = eq.apply_fractional_translation(img, tx, ty)
        close32(z, g[f"frac{k}_z"], f"frac{k}", 3e-6)
        assert np.array_equal(m.cpu().numpy(), g[f"frac{k}_m"]), k
        z, m = eq.apply_integer_translation(img, tx, ty)
        assert np.array_equal(z.cpu().numpy(), g[f"int{k}_z"]) and np.array_equal(m.cpu().numpy(), g[f"int{k}_m"]), k


def test_image_shifter_modes_vs_reference_fixture(golden):
    from afldm_amd.shift_utils.shifters import ImageShifter
    g = golden("g10_upfirdn.npz")
    img = t(g["img"]).cuda()
    for flt, tol in (("lanczos", 3e-6), ("fourier", 2e-5), ("fourier_crop", 2e-5)):
        sh = ImageShifter(flt)
        for k, (ti, tj) in enumerate(SHIFT_T):
            w, m = sh.shift(img, ti, tj)
            close32(w, g[f"shift_{flt}{k}_w"], f"{flt}{k}", tol)
            assert np.array_equal(m.cpu().numpy(), g[f"shift_{flt}{k}_m"]), (flt, k)
    g5 = golden("g5_shift_metrics.npz")
    img64 = t(g5["img"]).cuda()
    sh = ImageShifter()
    for k, tj in enumerate((0.125, 0.5, 1.0, 2.0)):
        w, m = sh.shift(img64, 0, tj * 8)
        # grid_sample's normalise / un-normalise round trip costs the reference ~2e-6 relative (an
        # integer shift is not exact there; it is here)
        close32(w, g5[f"bilinear_{k}"], f"bilinear{k}", 1e-5)
        assert np.array_equal(m.cpu().numpy(), g5[f"bilinear_mask_{k}"])


def test_bilinear_shifter_2d_vs_oracle():
    from oracle import shift as osh
    from afldm_amd.shift_utils.shifters import ImageShifter
    g = torch.Generator().manual_seed(5)
    img = torch.randn(2, 3, 40, 40, generator=g)
    sh = ImageShifter()
    for ti, tj in ((1.25, -3.5), (-0.75, 0.0), (6.0, 2.125), (-41.0, 3.0)):
        ref, rm = osh.shift_bilinear(img, ti, tj)
        w, m = sh.shift(img.cuda(), ti, tj)
        close32(w, ref, f"bilinear {ti},{tj}", 1e-5)
        assert torch.equal(m.cpu(), rm)
    bg = sh.translate_with_occ_bg(img.cuda(), 1.25, -3.5, ImageShifter.BgType.ORIGINAL_IMG)
    ref, rm = osh.shift_bilinear(img, 1.25, -3.5)
    close32(bg, ref * rm + img * (1 - rm), "occ bg", 1e-5)


def test_image_samplers_vs_reference_fixture(golden):
    from afldm_amd.shift_utils import shifters as S
    g = golden("g10_upfirdn.npz")
    img, img30 = t(g["img"]).cuda(), t(g["img30"]).cuda()
    for mode, tol in (("blur", 3e-6), ("ideal", 2e-5), ("nearest", 0), ("bilinear", 1e-6)):
        up = S.ImageUpsampler(2, mode)
        close32(up.upsample(img), g[f"up_{mode}"], f"up_{mode}", tol or 1e-9)
        if mode in ("blur", "ideal"):
            close32(up.low_pass(img), g[f"lowpass_{mode}"], f"lowpass_{mode}", tol)
    with pytest.raises(RuntimeError):                        # the reference's blur downsampler raises too
        S.ImageDownsampler(2, "blur").downsample(img)
    with pytest.raises(IndexError):                          # LPF_RFFT(cutoff=2) on N % 4 == 0
        S.ImageDownsampler(2, "ideal").downsample(img)
    close32(S.ImageDownsampler(2, "ideal").downsample(img30), g["down_ideal"], "down_ideal", 2e-5)
    close32(S.ImageDownsampler(2, "bilinear").downsample(img), g["down_bilinear"], "down_bilinear", 1e-6)
    z = S.upsample_pad_zero(img, 2)
    assert torch.equal(z[:, :, ::2, ::2], img) and float(z.abs().sum() - img.abs().sum()) == 0
    assert torch.equal(S.get_blur_kernel(3, 5)[2, 1], S.get_blur_kernel(1, 5)[0, 0])


def test_general_cutoff_ideal_filters_vs_reference_fixture(golden):
    from afldm_amd.af_libs import ideal_lpf as L
    g = golden("g10_upfirdn.npz")
    xz = t(g["xz"]).cuda()
    close32(L.LPF_RFFT(cutoff=0.25)(xz), g["lpf_q"], "lpf 1/4")
    close32(L.LPF_RECON_RFFT(cutoff=0.5)(xz), g["recon_h"], "recon 1/2")
    close32(L.LPF_RECON_RFFT(cutoff=0.25)(xz), g["recon_q"], "recon 1/4")
    close32(L.LPF_RECON_RFFT(cutoff=0.5, transform_mode="fft")(xz), g["recon_fft"], "recon fft")
    close32(L.UpsampleRFFT(4)(xz[:, :, :12, :12].contiguous()), g["up4"], "up4")
    close32(L.UpsampleRFFT(2, factor=0.5)(xz[:, :, :12, :12].contiguous()), g["up2_f2"], "up2 factor 1/2")
    with pytest.raises(TypeError):
        L.LPF_RFFT(cutoff=0.5, transform_mode="fft")(xz)
    # the host-built circulant agrees with the C library's matrices where both exist
    M = L._circulant(L._rect_1d(32, 0.5, 0.0).numpy())
    from afldm_amd import _lib
    assert np.abs(M - _lib.filter_matrix(2, 32).numpy()).max() <= 1e-7


def test_full_size_image_properties():
    """256^2 RGB batch (the decoded-image shifter's size): an integer Lanczos shift is an exact zero-padded
    translation, the fractional shift is linear, and a Fourier shift by t then -t restores band-limited content."""
    from afldm_amd.af_libs import equivariance as eq
    from afldm_amd.shift_utils.shifters import ImageShifter
    g = torch.Generator().manual_seed(11)
    img = torch.randn(64, 3, 256, 256, generator=g).cuda()
    z, m = eq.apply_fractional_translation(img, 5 / 256, -3 / 256)
    ref = torch.zeros_like(img)
    ref[:, :, :253, 5:] = img[:, :, 3:, :251]
    assert float((z - ref).abs().max()) <= 2e-6 * float(img.abs().max())
    zi, mi = eq.apply_integer_translation(img, 5 / 256, -3 / 256)
    assert torch.equal(zi, ref) and float(mi.sum()) == 64 * 3 * 253 * 251
    other = torch.randn(64, 3, 256, 256, generator=g).cuda()
    a, _ = eq.apply_fractional_translation(0.5 * img - 2 * other, 2.375 / 256, 0.5 / 256)
    b1, _ = eq.apply_fractional_translation(img, 2.375 / 256, 0.5 / 256)
    b2, _ = eq.apply_fractional_translation(other, 2.375 / 256, 0.5 / 256)
    assert float((a - (0.5 * b1 - 2 * b2)).abs().max()) <= 1e-5 * float(a.abs().max())
    from afldm_amd.af_libs.ideal_lpf import LPF_RFFT
    band = LPF_RFFT()(img[:4].contiguous())                  # no Nyquist content: the phase ramp is invertible
    sh = ImageShifter("fourier")
    w, _ = sh.shift(band, 3.25, -1.5)
    back, _ = sh.shift(w, -3.25, 1.5)
    assert rel_rms(back, band.cpu()) <= 1e-5


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
def test_masked_metrics_kernel_vs_oracle(golden, dtype):
    """mask_mse / mask_psnr / psnr on device tensors (one reduction kernel) against the oracle's torch form
    (itself bit-equal to the reference fixture), with full and broadcast ([B,1,H,W]) masks."""
    from afldm_amd.shift_utils import metrics
    from oracle import shift as osh
    g5 = golden("g5_shift_metrics.npz")
    a, b, m = (t(g5[k]) for k in ("ma", "mb", "mm"))
    assert abs(float(metrics.mask_mse(a.cuda(), b.cuda(), m.cuda())) - float(g5["mask_mse"])) <= 1e-6 * float(g5["mask_mse"])
    assert abs(float(metrics.mask_psnr(a.cuda(), b.cuda(), m.cuda())) - float(g5["mask_psnr"])) <= 1e-4
    assert abs(float(metrics.psnr(a.cuda(), b.cuda())) - float(g5["psnr"])) <= 1e-4
    gen = torch.Generator().manual_seed(9)
    x = torch.randn(3, 4, 32, 32, generator=gen).to(dtype)
    y = (x.float() + 0.1 * torch.randn(3, 4, 32, 32, generator=gen)).to(dtype)
    for mask in ((torch.rand(3, 4, 32, 32, generator=gen) > 0.3).float(), (torch.rand(3, 1, 32, 32, generator=gen) > 0.5).float()):
        ref_mse = osh.mask_mse(x.float(), y.float(), mask)          # un-expanded: the reference sums the mask's own shape
        ref_psnr = osh.mask_psnr(x.float(), y.float(), mask)
        got_mse = metrics.mask_mse(x.cuda(), y.cuda(), mask.cuda())
        got_psnr = metrics.mask_psnr(x.cuda(), y.cuda(), mask)          # a host mask is moved by the wrapper
        assert abs(float(got_mse) - float(ref_mse)) <= 2e-6 * float(ref_mse)
        assert abs(float(got_psnr) - float(ref_psnr)) <= 1e-4
    assert abs(float(metrics.psnr(x.cuda(), y.cuda(), i_max=2.0)) - float(osh.psnr(x.float(), y.float(), 2.0))) <= 1e-4
