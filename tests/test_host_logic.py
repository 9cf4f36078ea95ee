"""CPU-only tests: C-ABI exports vs the header, host-side logic (scheduler, configs, surgery,
shifter masks, sharding incl. a 2-process gloo run), and that the product refuses to compute
without a GPU (no CPU fallback)."""
import os
import re
import subprocess
import sys

import numpy as np
import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_library_exports_every_symbol_the_header_declares():
    from afldm_amd import _lib
    hdr = open(os.path.join(ROOT, "include", "afldm_hip.h")).read()
    declared = set(re.findall(r"\b(afldm_[a-z0-9_]+)\s*\(", hdr))
    declared -= {"afldm_conv_args"}
    assert len(declared) >= 20
    for name in declared:
        assert hasattr(_lib.lib, name), f"{name} declared in afldm_hip.h but not exported"
    assert declared == set(_lib.EXPORTS), (declared ^ set(_lib.EXPORTS))
    assert _lib.lib.afldm_version() >= 100
    # the product library carries no experiment: everything it exports is declared in the product header
    import subprocess
    nm = subprocess.run(["nm", "-D", "--defined-only", _lib.LIB_PATH], capture_output=True, text=True, check=True).stdout
    exported = {l.split()[-1] for l in nm.splitlines() if " T afldm_" in l}
    assert exported == declared, (exported ^ declared)


def test_experimental_library_is_separate_and_matches_its_header():
    """include/afldm_hip_experimental.h <-> libafldm_exp.so (measured-slower designs kept for A/B work): same symbols, none of
    them in the product header, and importing the package does not load the library."""
    from afldm_amd import _exp, _lib
    hdr = open(os.path.join(ROOT, "include", "afldm_hip_experimental.h")).read()
    decl = r"^(?:int|size_t|const char\s*\*)\s+(afldm_[a-z0-9_]+)\s*\("        # (the prose of the comments cites product calls)
    declared = set(re.findall(decl, hdr, re.M))
    product = open(os.path.join(ROOT, "include", "afldm_hip.h")).read()
    assert len(declared) >= 8 and not (declared & set(re.findall(decl, product, re.M)))
    assert _exp.available(), "python -m afldm_amd.build also links libafldm_exp.so"
    code = ("import sys; sys.path.insert(0, %r); import afldm_amd, afldm_amd.ops, afldm_amd.models.unet_2d, afldm_amd.harness; "
            "print(any('libafldm_exp' in l for l in open('/proc/self/maps')))" % ROOT)
    r = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, check=True)
    assert r.stdout.strip().splitlines()[-1] == "False", r.stdout
    assert declared == set(_exp.exports()), (declared ^ set(_exp.exports()))


def test_filter_matrices_match_oracle_and_reject_bad_sizes():
    from afldm_amd import _lib
    from oracle import ideal_filters as idf
    for N in (2, 4, 8, 16, 32):
        assert np.abs(_lib.filter_matrix(0, N, 2).numpy() - idf.up_matrix(N, 2)).max() < 1e-7
        assert np.abs(_lib.filter_matrix(1, 2 * N).numpy() - idf.down_matrix(2 * N)).max() < 1e-7
    assert np.abs(_lib.filter_matrix(0, 32, 8).numpy() - idf.up_matrix(32, 8)).max() < 1e-7
    L = _lib.filter_matrix(2, 16).numpy()
    assert np.abs(L[::2] - idf.down_matrix(16)).max() < 1e-7
    with pytest.raises(_lib.AfldmError):
        _lib.filter_matrix(0, 1, 8)          # the reference raises IndexError on this size too
    with pytest.raises(_lib.AfldmError):
        _lib.filter_matrix(1, 7)


def test_configs_agree_with_oracle_copies():
    from afldm_amd import configs as pc
    from oracle import configs as oc
    for k, v in oc.FFHQ_UNET.items():
        assert pc.FFHQ_UNET_CONFIG[k] == v, k
    for k, v in oc.FFHQ_DDIM.items():
        assert pc.FFHQ_DDIM_CONFIG[k] == v, k
    assert pc.tiny_unet_config()["block_out_channels"] == oc.tiny_unet()["block_out_channels"]


def test_unet_surface_state_dict_and_surgery():
    from afldm_amd.af_modules.af_api import make_af_unet
    from afldm_amd.af_modules.af_blocks import AliasFreeDownsample2D, AliasFreeUpsample2D, WarpedNonlinearity
    from afldm_amd.configs import FFHQ_UNET_CONFIG
    from afldm_amd.models.unet_2d import UNet2DModel
    from afldm_amd.pipelines.cross_frame_attn import get_unet_attn_processors
    from oracle import configs as oc, unet as ou
    unet = UNet2DModel.from_config(FFHQ_UNET_CONFIG)
    sd = ou.init_unet_params(oc.FFHQ_UNET)
    assert list(sd.keys()) and set(sd) == set(unet.state_dict())
    assert sum(p.numel() for p in unet.parameters()) == 256_401_796
    keys_before = list(unet.state_dict().keys())
    make_af_unet(unet)
    assert list(unet.state_dict().keys()) == keys_before, "AF surgery must not add parameters or buffers"
    n_warp = sum(isinstance(m, WarpedNonlinearity) for m in unet.modules())
    assert n_warp == 27                                                      # one per ResnetBlock2D
    assert sum(isinstance(m, AliasFreeDownsample2D) for m in unet.modules()) == 4
    assert sum(isinstance(m, AliasFreeUpsample2D) for m in unet.modules()) == 4
    assert all(d.conv.stride in (1, (1, 1)) for d in unet.modules() if isinstance(d, AliasFreeDownsample2D))
    assert not isinstance(unet.conv_act, WarpedNonlinearity)
    assert unet.config.sample_size == 32 and unet.config.in_channels == 4 and unet.dtype == torch.float32
    sites = [k[:-len(".processor")] for k in get_unet_attn_processors(unet)]
    assert sites == ou.attention_sites(oc.FFHQ_UNET) and len(sites) == 21


def test_no_cpu_fallback():
    from afldm_amd import ops
    from afldm_amd.configs import tiny_unet_config
    from afldm_amd.models.unet_2d import UNet2DModel
    from afldm_amd.schedulers.ddim import ffhq_ddim_scheduler
    unet = UNet2DModel.from_config(tiny_unet_config())
    with pytest.raises(RuntimeError, match="MI355X"):
        unet(torch.zeros(1, 4, 16, 16), 3)
    with pytest.raises(RuntimeError, match="no CPU path"):
        ops.silu(torch.zeros(4))
    s = ffhq_ddim_scheduler()
    s.set_timesteps(50)
    with pytest.raises(RuntimeError, match="MI355X"):
        s.step(torch.zeros(1, 4, 8, 8), 981, torch.zeros(1, 4, 8, 8))
    src = open(os.path.join(ROOT, "afldm_amd", "ops.py")).read() + open(os.path.join(ROOT, "afldm_amd", "engine.py")).read()
    assert "oracle" not in src, "the product must never import the oracle"


def test_scheduler_matches_oracle_tables():
    from afldm_amd.schedulers.ddim import DDIMScheduler, ffhq_ddim_scheduler
    from oracle.ddim import DDIM
    s, o = ffhq_ddim_scheduler(), DDIM()
    s.set_timesteps(50)
    o.set_timesteps(50)
    assert torch.equal(s.timesteps, o.timesteps) and torch.equal(s.alphas_cumprod, o.alphas_cumprod)
    assert float(s.final_alpha_cumprod) == float(o.final_alpha_cumprod) and s.init_noise_sigma == 1.0
    for t in (981, 501, 21, 1):
        assert s.coefficients(t) == o.coefficients(t)
    assert s.coefficient_table("cpu").shape == (50, 4)
    s2 = DDIMScheduler.from_config(s.config)
    assert dict(s2.config) == dict(s.config)
    with pytest.raises(ValueError):
        s.set_timesteps(2000)


def test_valid_masks_match_reference_fixture(golden):
    from afldm_amd.shift_utils.shifters import gen_valid_mask
    g = golden("g5_shift_metrics.npz")
    for k, (ti, tj) in enumerate(((1.5, -2.25), (-0.5, 0.0), (0.0, 3.0))):
        assert torch.equal(gen_valid_mask((1, 1, 8, 8), ti, tj), torch.from_numpy(g[f"valid_mask_{k}"]))


def test_metrics_match_reference_fixture(golden):
    from afldm_amd.shift_utils import metrics
    g = golden("g5_shift_metrics.npz")
    a, b, m = (torch.from_numpy(g[k]) for k in ("ma", "mb", "mm"))
    assert torch.equal(metrics.mask_mse(a, b, m), torch.from_numpy(g["mask_mse"]))
    assert torch.equal(metrics.mask_psnr(a, b, m), torch.from_numpy(g["mask_psnr"]))
    assert torch.equal(metrics.psnr(a, b), torch.from_numpy(g["psnr"]))


def test_randn_tensor_is_device_independent():
    from afldm_amd.utils import randn_tensor
    a = randn_tensor((2, 4, 8, 8), generator=torch.Generator().manual_seed(5))
    b = torch.randn((2, 4, 8, 8), generator=torch.Generator().manual_seed(5))
    assert torch.equal(a, b)


def test_shard_ranges_cover_and_balance():
    from afldm_amd.parallel import shard_range
    for total in (1, 7, 64, 512, 513):
        for world in (1, 2, 3, 8):
            rs = [shard_range(total, r, world) for r in range(world)]
            assert rs[0][0] == 0 and rs[-1][1] == total
            assert all(rs[i][1] == rs[i + 1][0] for i in range(world - 1))
            sizes = [e - s for s, e in rs]
            assert max(sizes) - min(sizes) <= 1


_WORKER = r"""
import os, sys, torch
sys.path.insert(0, sys.argv[1])
from afldm_amd import parallel
rank, world, _ = parallel.init_distributed("gloo")
total = int(sys.argv[2])
fn = lambda z: z * 2.0 + z.flatten(1).sum(1).view(-1, 1, 1, 1)      # per-sample, like the sampler
out = parallel.sample_sharded(fn, total, (4, 8, 8), 1234, rank, world, "cpu")
ref = fn(parallel.global_noise(total, (4, 8, 8), 1234))
assert out.shape == ref.shape and torch.equal(out, ref), (rank, out.shape)
parallel.barrier()
print("OK", rank)
"""


_WORKER_OFFSETS = r"""
import os, sys, torch
sys.path.insert(0, sys.argv[1])
from afldm_amd import parallel
rank, world, _ = parallel.init_distributed("gloo")
n = int(sys.argv[2])
# the harness's offset sharding (harness.shift_ldm / shift_ldm_sr) on a stub "denoiser": per-offset frames
# (CPU tensors) and errors (floats), interleaved over ranks, merged by ONE all_gather_object
def per_rank(mine):
    frames = {i: torch.full((1, 3, 4, 4), float(i)) * torch.arange(4.).view(1, 1, 4, 1) for i in mine}
    errors = {i: 0.5 * i + 0.25 for i in mine}
    return frames, errors
frames, errors = parallel.run_interleaved(n, rank, world, per_rank)
f1, e1 = per_rank(list(range(n)))
assert sorted(frames) == list(range(n)) == sorted(errors), (rank, sorted(frames))
assert all(torch.equal(frames[i], f1[i]) for i in range(n)) and errors == e1
assert parallel.interleaved(n, rank, world) == list(range(rank, n, world))
rec = parallel.rccl_record("cpu", rank, payload=torch.full((3, 4), float(rank)))
assert rec["world_size"] == 2 and rec["backend"] == "gloo" and rec["all_gather_verified"] and len(rec["ranks"]) == 2
assert sorted(r["rank"] for r in rec["ranks"]) == [0, 1] and rec["all_gather_bytes"] == 2 * 3 * 4 * 4
parallel.barrier()
print("OK", rank)
"""


def _free_port():
    import socket
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _torchrun(nproc, args, timeout):
    """`torch.distributed.run` on an OS-assigned free port; ONE retry on a failed run (a rendezvous port can be taken
    between the probe and the launch, and the first `import torch` of several processes at once can be slow)."""
    r = None
    for _ in range(2):
        cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={nproc}", "--master-addr",
               "127.0.0.1", "--master-port", str(_free_port())] + [str(a) for a in args]
        r = subprocess.run(cmd, capture_output=True, text=True, timeout=timeout, env=dict(os.environ, OMP_NUM_THREADS="1"))
        if r.returncode == 0:
            break
    return r



@pytest.mark.parametrize("n", [16, 5])
def test_two_process_gloo_harness_offset_sharding(tmp_path, n):
    """harness.shift_ldm's multi-GPU leg (SURVEY.md 8e 'Harness variant'): offsets interleaved over ranks, results merged
    once; plus the process-group record bench.py prints (world size, backend, ranks, the all-gather verified)."""
    import inspect
    from afldm_amd import harness
    src = inspect.getsource(harness)
    assert src.count("parallel.interleaved(") == 2 and src.count("parallel.gather_indexed(") == 2
    script = tmp_path / "worker_offsets.py"
    script.write_text(_WORKER_OFFSETS)
    r = _torchrun(2, [script, ROOT, n], 240)
    assert r.returncode == 0, r.stdout + r.stderr
    assert r.stdout.count("OK") == 2


@pytest.mark.parametrize("total", [8, 7])
def test_two_process_gloo_sharded_sampling(tmp_path, total):
    script = tmp_path / "worker.py"
    script.write_text(_WORKER)
    r = _torchrun(2, [script, ROOT, total], 240)
    assert r.returncode == 0, r.stdout + r.stderr
    assert r.stdout.count("OK") == 2


_WORKER8 = r"""
import os, sys, torch
sys.path.insert(0, sys.argv[1])
from afldm_amd import parallel
rank, world, local = parallel.init_distributed("gloo")
assert world == 8
total = int(sys.argv[2])
s, e = parallel.shard_range(total, rank, world)
sizes = [parallel.shard_range(total, r, world)[1] - parallel.shard_range(total, r, world)[0] for r in range(world)]
assert sum(sizes) == total and max(sizes) - min(sizes) <= 1 and e - s == sizes[rank]
fn = lambda z: z * 2.0 + z.flatten(1).sum(1).view(-1, 1, 1, 1)      # per-sample, like the sampler
out = parallel.sample_sharded(fn, total, (4, 8, 8), 4321, rank, world, "cpu")
ref = fn(parallel.global_noise(total, (4, 8, 8), 4321))
assert out.shape == ref.shape and torch.equal(out, ref), (rank, out.shape)
rec = parallel.rccl_record("cpu", local, payload=torch.full((e - s if total % world == 0 else 5, 4), float(rank)))
assert rec["world_size"] == 8 and rec["backend"] == "gloo" and rec["all_gather_verified"] and len(rec["ranks"]) == 8
assert sorted(r["rank"] for r in rec["ranks"]) == list(range(8)) and rec["distinct_devices"] == 8, rec
parallel.barrier()
print("OK", rank)
"""


@pytest.mark.parametrize("total", [512, 500])
def test_eight_process_gloo_sharded_sampling_and_record(tmp_path, total):
    """World size 8 - the shape of BASELINE configs[2] (batch 512 over 8 GPUs) - on CPU/gloo: even (512 = 8 x 64) and
    uneven (500: four ranks of 63, four of 62) shards through `sample_sharded`'s ONE all-gather equal the single-process
    result bit for bit, and the bench line's process-group record counts 8 ranks / 8 distinct local ranks
    (the world = 8 bookkeeping exercised before the first real SCALE record)."""
    script = tmp_path / "worker8.py"
    script.write_text(_WORKER8)
    r = _torchrun(8, [script, ROOT, total], 400)
    assert r.returncode == 0, r.stdout + r.stderr
    assert r.stdout.count("OK") == 8


def test_afldm_alias_and_diffusers_shim():
    """`afldm.X` is the same module object as `afldm_amd.X`; the diffusers shim exposes the names
    the reference's hot-path files import (af_blocks.py:6-7, cross_frame_attn.py:3, ldm_pipeline.py:1-4)."""
    import importlib
    import afldm_amd.compat as compat
    shimmed = compat.install()
    from afldm.af_modules.af_api import make_af_unet as a
    from afldm_amd.af_modules.af_api import make_af_unet as b
    assert a is b
    from afldm.pipelines.cross_frame_attn import AttnState as s1
    from afldm_amd.pipelines.cross_frame_attn import AttnState as s2
    assert s1 is s2
    import afldm.shift_utils.shifters, afldm.pipelines.ldm_pipeline, afldm.io_utils, afldm.af_libs.ideal_lpf  # noqa
    with pytest.raises(ModuleNotFoundError):
        importlib.import_module("afldm.trainers.ldm_trainer")        # out of scope: not provided
    if shimmed:
        from diffusers.models import UNet2DModel
        from diffusers.models.attention_processor import AttnProcessor2_0  # noqa
        from diffusers.models.downsampling import Downsample2D  # noqa
        from diffusers.pipelines.pipeline_utils import DiffusionPipeline, ImagePipelineOutput  # noqa
        from diffusers.schedulers import DDIMScheduler  # noqa
        from diffusers.utils.torch_utils import randn_tensor  # noqa
        from afldm_amd.models.unet_2d import UNet2DModel as U2
        assert UNet2DModel is U2


def test_i2sb_scheduler_tables_and_known_answers():
    """sigma_fwd[0] = 0.0387298, sigma_fwd[999] = sigma_bwd[0] = 2.9672334."""
    from afldm_amd.configs import FFHQ_DDIM_CONFIG
    from afldm_amd.schedulers.i2sb import I2SBScheduler
    from oracle.i2sb import I2SB
    cfg = {k: v for k, v in FFHQ_DDIM_CONFIG.items() if k not in ("set_alpha_to_one",)}
    s, o = I2SBScheduler.from_config(cfg), I2SB()
    assert abs(float(s.std_fwd[0]) - 0.0387298) < 1e-6 and abs(float(s.std_fwd[999]) - 2.9672334) < 2e-6
    assert abs(float(s.std_bwd[0]) - 2.9672334) < 2e-6
    assert torch.equal(s.std_fwd, o.std_fwd) and torch.equal(s.mu_x0, o.mu_x0) and torch.equal(s.std_sb, o.std_sb)
    s.set_timesteps(100)
    o.set_timesteps(100)
    assert torch.equal(s.timesteps, o.timesteps) and s.previous_timestep(991) == 981
    x0, x1 = torch.randn(3, 4, 8, 8), torch.randn(3, 4, 8, 8)
    ts = torch.tensor([5, 500, 999])
    xt = s.add_noise(x0, x1, ts, is_ode=True)
    assert torch.equal(xt, o.add_noise(x0, x1, ts, is_ode=True))
    assert torch.allclose(s.compute_label(ts, x0, xt), o.compute_label(ts, x0, xt))
    with pytest.raises(RuntimeError, match="MI355X"):
        s.step(x0, 991, x1)


def test_sr4x_degrade_matrix_matches_oracle():
    """The host-built degrade matrix of the product equals the oracle's (same SVD truncation)."""
    from afldm_amd.af_libs.superresolution import degrade_matrix
    from oracle.superresolution import degrade_matrix as ref
    for flt in ("bicubic", "pool"):
        for n in (64, 256):
            assert (degrade_matrix(flt, n) - ref(flt, n)).abs().max() < 1e-6



def test_host_built_filter_matrices_for_general_cutoffs():
    """af_libs.ideal_lpf builds circulants on the host for cutoffs the C library has no kind for: they must agree
    with afldm_filter_matrix where both exist and with the oracle's FFT form elsewhere (CPU only: matrices)."""
    import numpy as np
    from afldm_amd import _lib
    from afldm_amd.af_libs import ideal_lpf as L
    from oracle import ideal_filters as idf
    for N in (4, 8, 16, 32):
        lp = L._circulant(L._rect_1d(N, 0.5, 0.0).numpy())
        assert np.abs(lp - _lib.filter_matrix(2, N).numpy()).max() <= 1e-7
        up = 2 * L._circulant(L._rect_1d(2 * N, 0.5, 0.5).numpy())[:, ::2]
        assert np.abs(up - _lib.filter_matrix(0, N, 2).numpy()).max() <= 1e-7
    # a cutoff only the host path serves: y = C x C^T must equal the rfft2 -> mask -> irfft2 form
    g = torch.Generator().manual_seed(4)
    x = torch.randn(1, 2, 24, 24, generator=g, dtype=torch.float64)
    C = torch.from_numpy(L._circulant(L._rect_1d(24, 0.25, 0.0).numpy()))
    assert (C @ x @ C.T - idf.lpf_rfft(x.float(), 0.25).double()).abs().max() <= 2e-6
    R = torch.from_numpy(L._circulant(L._rect_1d(24, 0.25, 0.5).numpy()))
    assert (R @ x @ R.T - idf.lpf_recon_rfft(x.float(), 0.25).double()).abs().max() <= 2e-6
    with pytest.raises(IndexError):            # cutoff 2 on N % 4 == 0: the reference's mask indexes out of range
        L._rect_1d(32, 2, 0.0)


def test_phase_circulant_and_translation_masks():
    """Host pieces of the fourier / lanczos shifters: an integer phase ramp is a cyclic permutation, a fractional
    one matches the oracle's fft2 form, and the validity boxes of the Lanczos / integer translations match the
    reference fixtures (CPU only: no kernel is launched)."""
    import numpy as np
    from afldm_amd.af_libs import equivariance as eq
    from afldm_amd.shift_utils import shifters as S
    from oracle import upfirdn as ou
    re, im = S._phase_circulant(8, 3)
    assert np.abs(im).max() <= 1e-12 and np.allclose(re, np.roll(np.eye(8), 3, axis=0), atol=1e-12)
    g = torch.Generator().manual_seed(6)
    x = torch.randn(1, 1, 16, 16, generator=g, dtype=torch.float64)
    (hr, hi), (wr, wi) = S._phase_circulant(16, 1.375), S._phase_circulant(16, -0.5)
    y = torch.from_numpy(hr) @ x @ torch.from_numpy(wr).T - torch.from_numpy(hi) @ x @ torch.from_numpy(wi).T
    assert (y - ou.fourier_shift_batch(x, 1.375, -0.5)).abs().max() <= 2e-6     # the reference builds its phase ramp in fp32
    probe = torch.zeros(1, 1, 32, 32)
    for ix, iy, a in ((3, -2, 3), (0, 0, 3), (40, 1, 3), (-30, 31, 3)):
        b = a - 1
        box = eq._box_mask(probe, max(iy + a, 0), min(iy - b, 0) + 32, max(ix + a, 0), min(ix - b, 0) + 32)
        _, ref = ou.apply_fractional_translation(probe, ix / 32, iy / 32)
        assert torch.equal(box, ref), (ix, iy)


def test_upfirdn2d_host_argument_handling():
    from afldm_amd.af_libs.torch_utils.ops import upfirdn2d as up
    from oracle import upfirdn as ou
    assert up._parse_scaling(3) == (3, 3) and up._parse_scaling([2, 5]) == (2, 5)
    assert up._parse_padding(2) == (2, 2, 2, 2) and up._parse_padding([1, 3]) == (1, 1, 3, 3)
    assert up._parse_padding([1, -2, 3, 4]) == (1, -2, 3, 4)
    with pytest.raises(AssertionError):
        up._parse_scaling(0)
    for f, kw in (([1, 3, 3, 1], {}), ([1, 2, 3, 4, 4, 3, 2, 1], {"gain": 2}), ([[1, 2], [3, 4]], {"flip_filter": True}),
                  (None, {}), ([1, 2, 1], {"normalize": False, "separable": True})):
        assert torch.equal(up.setup_filter(f, **kw), ou.setup_filter(f, **kw)), (f, kw)
    with pytest.raises(RuntimeError):          # no CPU path
        up.upfirdn2d(torch.zeros(1, 1, 4, 4), torch.ones(2, 2))


def test_integration_doc_lists_every_entry_point():
    """INTEGRATION.md's C-ABI summary must name every function include/afldm_hip.h declares."""
    import re
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    header = open(os.path.join(root, "include", "afldm_hip.h")).read()
    doc = open(os.path.join(root, "INTEGRATION.md")).read()
    names = sorted(set(re.findall(r"\b(afldm_[a-z0-9_]+)\s*\(", header)))
    assert len(names) >= 36
    missing = [n for n in names if n not in doc]
    assert not missing, missing


def test_graph_schedules_are_the_oracles_updates_in_linear_form():
    """The (c0, c1, c2, c3) rows the captured-graph engine replays - x0 = (x - c1 eps) / c0, x_prev = c2 x0 + c3 eps, the DDIM
    kernel's form - for the I2SB ODE bridge (I2SBScheduler.ode_schedule; reference i2sb_scheduler.py:382-459, i2sb_pipeline.py:48-50)
    and for DDIM inversion (MyLDMPipeline._inversion_rows; reference ldm_pipeline.py:133-160) against the oracle's scheduler
    arithmetic on random tensors: no GPU involved, only the host tables."""
    from afldm_amd.configs import FFHQ_DDIM_CONFIG
    from afldm_amd.pipelines.ldm_pipeline import MyLDMPipeline, _InversionSchedule
    from afldm_amd.schedulers.ddim import ffhq_ddim_scheduler
    from afldm_amd.schedulers.i2sb import I2SBScheduler
    from oracle import ddim as od, i2sb as oi
    g = torch.Generator().manual_seed(0)
    x, e = torch.randn(2, 4, 8, 8, generator=g), torch.randn(2, 4, 8, 8, generator=g)

    def linear(c):
        x0 = (x - c[1] * e) / c[0]
        return c[2] * x0 + c[3] * e

    cfg = {k: v for k, v in FFHQ_DDIM_CONFIG.items() if k != "set_alpha_to_one"}
    sched = I2SBScheduler.from_config(cfg)
    assert I2SBScheduler.from_config(dict(cfg, clip_sample=True)).ode_schedule(100) is None      # the clamp is not linear
    ode = sched.ode_schedule(100)
    ode.set_timesteps(99)
    ref = oi.I2SB()
    ref.set_timesteps(100)
    assert ode.evaluations == 99 and ode._timesteps_host == [int(t) for t in ref.timesteps[:99]]
    table = ode.coefficient_table("cpu")
    assert table.shape == (99, 4)
    for i in (0, 1, 50, 98):
        t = ode._timesteps_host[i]
        want = ref.step(e, t, x, is_ode=True)
        assert (linear([float(v) for v in table[i]]) - want).abs().max() <= 2e-6 * want.abs().max()
    # DDIM inversion over a 6-step schedule
    pipe = MyLDMPipeline.__new__(MyLDMPipeline)
    pipe.scheduler = ffhq_ddim_scheduler()
    pipe.scheduler.set_timesteps(6)
    rows = pipe._inversion_rows()
    ref = od.DDIM()
    ref.set_timesteps(6)
    ts = ref.timesteps.flip(0)
    assert [t for t, _ in rows] == [int(t) for t in ts]
    for i, (t, c) in enumerate(rows):
        a_t = ref.alphas_cumprod[int(ts[i])]
        a_prev = ref.alphas_cumprod[int(ts[i - 1])] if i > 0 else ref.final_alpha_cumprod
        want = a_t ** 0.5 * ((x - (1 - a_prev) ** 0.5 * e) / a_prev ** 0.5) + (1 - a_t) ** 0.5 * e
        assert (linear(c) - want).abs().max() <= 2e-6 * want.abs().max()
    inv = _InversionSchedule(pipe.scheduler, rows)
    assert inv.coefficient_table("cpu").shape == (6, 4) and inv._timesteps_host == [t for t, _ in rows]
