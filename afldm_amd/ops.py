"""Tensor-level wrappers over the C ABI (include/afldm_hip.h).  PyTorch here is only the owner
of device memory and streams; every op below is one or two HIP kernel launches from
libafldm_hip.so on torch's current stream.  All activations are NHWC ([B, H, W, C]) or
token-major ([B, T, C]) contiguous CUDA tensors in fp32 or bf16.  CPU tensors raise."""
import ctypes
import os

import torch

from . import _exp, _lib
from ._lib import AfActArgs, ConvArgs, DTYPE_CODE, SepArgs, check, stream_ptr

_FILTER_CACHE = {}
_PROFILE = None


class _RecordingLib:
    """libafldm_hip.so's entry points.  Outside a Profiler this is a plain pass-through; inside one every call is also
    noted as (function, arguments) so that the launches between _begin() and _end() can be re-issued later on the
    same buffers: bench.py captures each kernel family's launches of one step into a HIP graph and times the replay
    (kernel time without host gaps).  The stream is the last argument of every launching entry point; a replay
    substitutes the stream that is current then."""

    def __init__(self, raw):
        self._raw = raw
        self._wrapped = {}

    def __getattr__(self, name):
        w = self._wrapped.get(name)
        if w is None:
            f = getattr(self._raw, name)

            def w(*a, _f=f):
                if _PROFILE is not None:
                    _PROFILE.pending.append((_f, a))
                return _f(*a)
            self._wrapped[name] = w
        return w


lib = _RecordingLib(_lib.lib)


def ptr(t):
    """Device address of a tensor (None -> NULL).  Under a Profiler the tensor is also kept alive, so that a recorded
    launch can be replayed on buffers that still hold the step's real activations."""
    if t is None:
        return None
    if _PROFILE is not None:
        _PROFILE.live.append(t)
    return t.data_ptr()


class Profiler:
    """Per-op HIP-event timing on the launch stream (bench.py's roofline leg).  Records, per
    wrapper call, the kernel family, its ALGORITHMIC flops / bytes and two events around it."""

    def __init__(self):
        self.records = []
        self.pending = []        # (function, args) of the library calls since the last _begin()
        self.live = []           # tensors whose addresses went into recorded calls

    def __enter__(self):
        global _PROFILE
        _PROFILE = self
        return self

    def __exit__(self, *a):
        global _PROFILE
        _PROFILE = None

    def summary(self):
        torch.cuda.synchronize()
        agg = {}
        for kind, flops, nbytes, e0, e1, _ in self.records:
            d = agg.setdefault(kind, dict(launches=0, ms=0.0, flops=0.0, bytes=0.0))
            d["launches"] += 1
            d["ms"] += e0.elapsed_time(e1)
            d["flops"] += flops
            d["bytes"] += nbytes
        return agg


def _begin():
    if _PROFILE is None:
        return None
    _PROFILE.pending = []
    e = torch.cuda.Event(enable_timing=True)
    e.record()
    return e


def _replay_of(calls):
    """Re-issue recorded launches on the current stream (their last argument is the stream)."""
    def replay():
        st = stream_ptr()
        for f, a in calls:
            check(f(*a[:-1], st), "replay")
    return replay


def _end(tok, kind, flops=0.0, nbytes=0.0, replay=None):
    """replay: a callable that re-issues exactly this launch on the same buffers (bench.py replays a family's launches
    back to back from a captured graph to time the kernels without the host's launch gaps)."""
    if tok is None:
        return
    e = torch.cuda.Event(enable_timing=True)
    e.record()
    if replay is None:
        st = stream_ptr()
        calls = [(f, a) for f, a in _PROFILE.pending if a and a[-1] == st]       # launches only (queries take no stream)
        if calls:
            replay = _replay_of(calls)
    _PROFILE.pending = []
    _PROFILE.records.append((kind, float(flops), float(nbytes), tok, e, replay))


def _dev(t, name="tensor"):
    if not t.is_cuda:
        raise RuntimeError(f"afldm_amd: {name} must live on an MI355X (cuda) device; there is no CPU path")
    if not t.is_contiguous():
        raise RuntimeError(f"afldm_amd: {name} must be contiguous")
    return t


def _code(t):
    try:
        return DTYPE_CODE[t.dtype]
    except KeyError:
        raise RuntimeError(f"afldm_amd: dtype {t.dtype} unsupported (fp32 / bf16 only)")


def filter_matrices(N, device):
    """(U [2N,N], D [N,2N]) device fp32 matrices for a plane of size N (cached per device)."""
    key = ("act", N, str(device))
    if key not in _FILTER_CACHE:
        U = _lib.filter_matrix(0, N, 2).to(device)
        D = _lib.filter_matrix(1, 2 * N).to(device)
        _FILTER_CACHE[key] = (U, D)
    return _FILTER_CACHE[key]


def packed_filters(N, dtype, device):
    """Device LDS image of (U, D) for the MFMA alias-free activation (N = 16, 32), cached."""
    if N < 4:
        return None
    key = ("packed", N, dtype, str(device))
    if key not in _FILTER_CACHE:
        U, D = filter_matrices(N, device)
        nbytes = lib.afldm_af_pack_bytes(N, DTYPE_CODE[dtype])
        buf = torch.empty(nbytes, dtype=torch.uint8, device=device)
        check(lib.afldm_af_pack(ptr(U), ptr(D), N, DTYPE_CODE[dtype], ptr(buf), stream_ptr()), "af_pack")
        _FILTER_CACHE[key] = buf
    return _FILTER_CACHE[key]


def down_matrix(N, device):
    """D [N/2, N] for AliasFreeDownsample2D on an N x N plane."""
    key = ("down", N, str(device))
    if key not in _FILTER_CACHE:
        _FILTER_CACHE[key] = _lib.filter_matrix(1, N).to(device)
    return _FILTER_CACHE[key]


def up_matrix(N, up, device):
    key = ("up", N, up, str(device))
    if key not in _FILTER_CACHE:
        _FILTER_CACHE[key] = _lib.filter_matrix(0, N, up).to(device)
    return _FILTER_CACHE[key]


# ----------------------------------------------------------------------------- layout
def to_nhwc(x, dtype=torch.float32, out=None):
    """NCHW fp32 -> NHWC dtype."""
    _dev(x, "x")
    assert x.dtype == torch.float32 and x.ndim == 4
    B, C, H, W = x.shape
    if out is None:
        out = torch.empty((B, H, W, C), dtype=dtype, device=x.device)
    check(lib.afldm_nchw_to_nhwc(ptr(x), ptr(out), B, C, H, W, _code(out), stream_ptr()), "nchw_to_nhwc")
    return out


def to_nchw(x, out=None):
    """NHWC dtype -> NCHW fp32."""
    _dev(x, "x")
    B, H, W, C = x.shape
    if out is None:
        out = torch.empty((B, C, H, W), dtype=torch.float32, device=x.device)
    check(lib.afldm_nhwc_to_nchw(ptr(x), ptr(out), B, C, H, W, _code(x), stream_ptr()), "nhwc_to_nchw")
    return out


def pack_weight(w, dtype):
    """OIHW (or [O, I]) fp32 parameter -> OHWI dtype."""
    _dev(w, "weight")
    w = w.detach().to(torch.float32).contiguous()
    if w.ndim == 2:
        O, I = w.shape
        KH = KW = 1
    else:
        O, I, KH, KW = w.shape
    out = torch.empty((O, KH, KW, I), dtype=dtype, device=w.device)
    check(lib.afldm_pack_weight(ptr(w), ptr(out), O, I, KH, KW, _code(out), stream_ptr()), "pack_weight")
    return out


# ----------------------------------------------------------------------------- small ops
def timestep_embedding(t, dim, flip_sin_to_cos=True, freq_shift=0.0, dtype=torch.float32, out=None):
    _dev(t, "t")
    assert t.dtype == torch.float32 and t.ndim == 1
    rows = t.shape[0]
    if out is None:
        out = torch.empty((rows, dim), dtype=dtype, device=t.device)
    check(lib.afldm_timestep_embedding(ptr(t), ptr(out), rows, dim, int(flip_sin_to_cos), float(freq_shift),
                                       _code(out), stream_ptr()), "timestep_embedding")
    return out


def silu(x, out=None):
    _dev(x, "x")
    if out is None:
        out = torch.empty_like(x)
    check(lib.afldm_silu(ptr(x), ptr(out), x.numel(), _code(x), stream_ptr()), "silu")
    return out


def _cat_args(x1, x2):
    _dev(x1, "x1")
    C1 = x1.shape[-1]
    if x2 is None:
        return C1, None, 0
    _dev(x2, "x2")
    assert x2.shape[:-1] == x1.shape[:-1] and x2.dtype == x1.dtype
    return C1, x2, x2.shape[-1]


_MAX_SPLITS = 32       # row splits of GroupNorm partial sums a consumer should have to walk


class GNStats:
    """Per-channel GroupNorm partial sums of a tensor or of a virtual concat x1|x2:
    st1 [B, S1, C1, 2], st2 [B, S2, C2, 2] fp32 (sum, sum of squares per pixel split).  Producers:
    gn_stats (stand-alone pass) and conv2d(..., want_stats=True) (GEMM epilogue, attached to the
    output tensor as `.gn_partial`).  Consumers finish mean / rstd themselves."""
    __slots__ = ("st1", "st2")

    def __init__(self, st1, st2=None):
        self.st1, self.st2 = st1, st2

    @property
    def S1(self):
        return self.st1.shape[1]

    @property
    def S2(self):
        return 0 if self.st2 is None else self.st2.shape[1]


def _tensor_stats(x, out=None):
    """Per-channel partial sums of ONE NHWC tensor: the ones its producing conv attached, else a pass."""
    st = getattr(x, "gn_partial", None)
    if st is not None and out is None:
        return st
    assert not getattr(x, "c8", False), "a tensor in 8-channel blocks carries its producer's statistics (no stand-alone pass)"
    _dev(x, "x")
    B, C = x.shape[0], x.shape[-1]
    HW = x.numel() // (B * C)
    S = lib.afldm_gn_stats_splits(HW)
    if out is None:
        out = torch.empty((B, S, C, 2), dtype=torch.float32, device=x.device)
    tok = _begin()
    check(lib.afldm_gn_stats(ptr(x), C, ptr(out), B, HW, _code(x), stream_ptr()), "gn_stats")
    _end(tok, "gn_stats", 0, B * HW * C * x.element_size())
    if st is None:
        x.gn_partial = out      # a skip connection is normalised twice (down block, then up block): one pass
    return out


def gn_stats(x1, G=None, x2=None, out=None):
    """GroupNorm statistics (GNStats) of the virtual concat x1|x2.  `G` is accepted for source
    compatibility and ignored: the partial sums are per channel, any grouping is formed by the
    consumer.  `out` (single tensor only) forces a stand-alone pass into that buffer."""
    C1, x2, C2 = _cat_args(x1, x2)
    return GNStats(_tensor_stats(x1, out), None if x2 is None else _tensor_stats(x2))


def carry_stats(dst, src):
    """Views / reshapes create new tensor objects: hand the attached statistics over."""
    st = getattr(src, "gn_partial", None)
    if st is not None:
        dst.gn_partial = st
    return dst


def gn_mean_rstd(stats, G, n, eps):
    """(mean, rstd) [B, G] from GNStats - host-side helper for tests / debugging only."""
    parts = [stats.st1.double().sum(1)] + ([] if stats.st2 is None else [stats.st2.double().sum(1)])
    s = torch.cat(parts, 1)                                     # [B, C, 2]
    B, C, _ = s.shape
    s = s.view(B, G, C // G, 2).sum(2)
    mean = s[..., 0] / n
    var = (s[..., 1] / n - mean * mean).clamp_min(0)
    return mean.float(), (1.0 / torch.sqrt(var + eps)).float()


def _stats_ptrs(stats):
    if stats is None:
        return None, 0, None, 0
    return ptr(stats.st1), stats.S1, ptr(stats.st2), stats.S2


def gn_apply(x1, stats, gamma, beta, G, eps, act=0, x2=None, out=None):
    C1, x2, C2 = _cat_args(x1, x2)
    B = x1.shape[0]
    HW = x1.numel() // (B * C1)
    if out is None:
        out = torch.empty(tuple(x1.shape[:-1]) + (C1 + C2,), dtype=x1.dtype, device=x1.device)
    p1, S1, p2, S2 = _stats_ptrs(stats)
    tok = _begin()
    check(lib.afldm_gn_apply(ptr(x1), C1, ptr(x2), C2, p1, S1, p2, S2, ptr(gamma), ptr(beta), ptr(out), B, HW, G,
                             float(eps), int(act), _code(x1), stream_ptr()), "gn_apply")
    _end(tok, "gn_apply", 0, 2 * B * HW * (C1 + C2) * x1.element_size())
    return out


# ----------------------------------------------------------------------------- alias-free ops
# AFLDM_NO_C8=1: every tensor NHWC (A/B).  Default: inside a ResnetBlock2D of the 32^2 / 16^2 levels the tensors between the
# alias-free activations and the 3x3 convolutions travel in 8-channel blocks [B][C/8][N][N][8] (`.c8 = True` on the tensor
# object; same shape, same values): an activation item is a contiguous run then (profiles/r05/c8_layout_ab.txt).
_C8 = os.environ.get("AFLDM_NO_C8", "0") != "1"


def is_c8(x):
    return bool(getattr(x, "c8", False))


def af_act(x1, x2=None, stats=None, gamma=None, beta=None, G=0, eps=0.0, out=None, out_c8=False, out_const=False):
    """[GroupNorm-apply ->] WarpedNonlinearity(SiLU) on an NHWC tensor (virtual concat x1|x2).  out_c8: the output in
    8-channel blocks (tagged `.c8`); an input tagged `.c8` is read in that layout (N = 16 / 32, bf16).
    out_const (N = 2 only): the result of a 2 x 2 plane is plane-constant (lpf(4) = [1,0,0,0], ideal_lpf.py:17-21) - return
    it once per plane as [B, C] tagged `.const2` (afldm_af_act_const2)."""
    C1, x2, C2 = _cat_args(x1, x2)
    B, N, N2, _ = x1.shape
    assert N == N2, "the reference's ideal filters assume square planes (ideal_lpf.py:80)"
    if out_const:
        assert N == 2 and out is None and not is_c8(x1)
        out = torch.empty((B, C1 + C2), dtype=x1.dtype, device=x1.device)
        U, D = filter_matrices(2, x1.device)
        p1, S1, p2, S2 = _stats_ptrs(stats)
        tok = _begin()
        check(lib.afldm_af_act_const2(ptr(x1), C1, ptr(x2), C2, p1, S1, p2, S2, ptr(gamma), ptr(beta), int(G), float(eps), ptr(U),
                                      ptr(D), ptr(out), B, _code(x1), stream_ptr()), "af_act_const2")
        _end(tok, "af_act_N2", 24.0 * 8 * B * (C1 + C2), B * 5 * (C1 + C2) * x1.element_size())
        out.const2 = True
        return out
    if out is None:
        out = torch.empty((B, N, N, C1 + C2), dtype=x1.dtype, device=x1.device)
    if out_c8 or is_c8(x1):
        assert N in (16, 32) and x1.dtype == torch.bfloat16 and (x2 is None or not is_c8(x1)) and not is_c8(x2)
        U, D = filter_matrices(N, x1.device)
        packed = packed_filters(N, x1.dtype, x1.device)
        p1, S1, p2, S2 = _stats_ptrs(stats)
        tok = _begin()
        check(lib.afldm_af_act_c8(ptr(x1), C1, ptr(x2), C2, p1, S1, p2, S2, ptr(gamma), ptr(beta), int(G), float(eps), ptr(U),
                                  ptr(D), ptr(packed), ptr(out), B, N, _code(x1), 1 if is_c8(x1) else 0, 1 if out_c8 else 0,
                                  stream_ptr()), "af_act_c8")
        _end(tok, f"af_act_N{N}", 24.0 * N ** 3 * B * (C1 + C2), 2 * B * N * N * (C1 + C2) * x1.element_size())
        if out_c8:
            out.c8 = True
        return out
    if N > 32:
        if x2 is not None:
            raise RuntimeError("afldm_amd: the large-plane activation path (N > 32) does not take a virtual concat")
        if x1.dtype == torch.float32 and N > 64:
            # fp32 128^2 planes: the matrices do not fit LDS -> generic (VALU) separable products:
            # GN-apply, U x U^T, SiLU, D z D^T as four elementary launches (exact-parity mode only)
            xn = x1 if stats is None else gn_apply(x1, stats, gamma, beta, G, eps, act=0)
            U, D = filter_matrices(N, x1.device)
            return af_resample(silu(af_resample(xn, U)), D, out=out)
        return _af_act_large(x1, stats, gamma, beta, G, eps, out)
    U, D = filter_matrices(N, x1.device)
    packed = packed_filters(N, x1.dtype, x1.device)
    p1, S1, p2, S2 = _stats_ptrs(stats)
    tok = _begin()
    check(lib.afldm_af_act(ptr(x1), C1, ptr(x2), C2, p1, S1, p2, S2, ptr(gamma), ptr(beta), int(G), float(eps), ptr(U),
                           ptr(D), ptr(packed), ptr(out), B, N, _code(x1), stream_ptr()), "af_act")
    # dense separable form: 24 N^3 flop per plane; one read + one write of the tensor
    _end(tok, f"af_act_N{N}", 24.0 * N ** 3 * B * (C1 + C2), 2 * B * N * N * (C1 + C2) * x1.element_size())
    return out


def _resample_plane(x, M, R, out, want_stats):
    """One-kernel MFMA resample (afldm_af_resample_plane) for the UNet's 16 -> 32 / 32 -> 16 sites."""
    B, N, _, C = x.shape
    st = torch.empty((B, 1, C, 2), dtype=torch.float32, device=x.device) if want_stats else None
    tok = _begin()
    check(lib.afldm_af_resample_plane(ptr(x), ptr(M), ptr(out), ptr(st), B, N, C, R, _code(x), stream_ptr()),
          "af_resample_plane")
    _end(tok, "af_resample_plane", 4.0 * (N * N * R + N * R * R) * B * C / 2, B * (N * N + R * R) * C * x.element_size())
    if st is not None:
        out.gn_partial = st
    return out


_PLANE_RESAMPLE = os.environ.get("AFLDM_NO_RESAMPLE_PLANE", "0") != "1"


def af_up2(x, out=None, workspace=None):
    _dev(x, "x")
    B, N, N2, C = x.shape
    assert N == N2
    U = up_matrix(N, 2, x.device)
    if out is None:
        out = torch.empty((B, 2 * N, 2 * N, C), dtype=x.dtype, device=x.device)
    if N >= 32 and C % 16 == 0 and (x.dtype == torch.bfloat16 or N <= 64):
        return _resample_large(x, U, 2 * N, out)
    if N == 16 and C % 16 == 0 and _PLANE_RESAMPLE:
        return _resample_plane(x, U, 32, out, False)
    if workspace is None:
        workspace = torch.empty(B * 2 * N * N * C, dtype=torch.float32, device=x.device)
    assert workspace.numel() >= B * 2 * N * N * C
    tok = _begin()
    check(lib.afldm_af_up2(ptr(x), ptr(U), ptr(out), ptr(workspace), B, N, C, _code(x), stream_ptr()), "af_up2")
    _end(tok, "af_up2", 12.0 * N ** 3 * B * C, 5 * B * N * N * C * x.element_size())
    return out


def af_lpf_down2(x, out=None, workspace=None, want_stats=False):
    _dev(x, "x")
    B, N, N2, C = x.shape
    assert N == N2
    D = down_matrix(N, x.device)
    if out is None:
        out = torch.empty((B, N // 2, N // 2, C), dtype=x.dtype, device=x.device)
    if N >= 64 and C % 16 == 0 and (x.dtype == torch.bfloat16 or N <= 128):
        return _resample_large(x, D, N // 2, out)
    if N == 32 and C % 16 == 0 and _PLANE_RESAMPLE:
        return _resample_plane(x, D, 16, out, want_stats)
    if workspace is None:
        workspace = torch.empty(B * (N // 2) * N * C, dtype=torch.float32, device=x.device)
    assert workspace.numel() >= B * (N // 2) * N * C
    st = None
    if want_stats and C % 4 == 0 and N <= 32 and N & (N - 1) == 0:
        st = torch.empty((B, N // 2, C, 2), dtype=torch.float32, device=x.device)    # one split per output row
    tok = _begin()
    check(lib.afldm_af_lpf_down2(ptr(x), ptr(D), ptr(out), ptr(workspace), ptr(st), B, N, C, _code(x), stream_ptr()),
          "af_lpf_down2")
    _end(tok, "af_lpf_down2", 1.5 * N ** 3 * B * C, 1.25 * B * N * N * C * x.element_size())
    if st is not None:
        out.gn_partial = st
    return out


# ----------------------------------------------------------------------------- large planes (AF-VAE)
def sep_pass(x, y, M, K, R, outer_count, inner_count, in_outer_stride, in_k_stride, out_outer_stride, out_k_stride,
             M2=None, R2=0, gn_table=None, C=0, outer_per_sample=1, act=0, up_identity=False):
    """One separable pass (afldm_sep_pass): y[line][r] = act(M xn[line]) or M2 silu(M xn[line]).
    up_identity: M is the x2 upsampler U, whose even rows are the identity (afldm_sep_args.up_identity)."""
    a = SepArgs()
    a.x, a.y, a.M, a.M2, a.gn_table = ptr(x), ptr(y), ptr(M), ptr(M2), ptr(gn_table)
    a.outer_count, a.inner_count = int(outer_count), int(inner_count)
    a.in_outer_stride, a.in_k_stride = int(in_outer_stride), int(in_k_stride)
    a.out_outer_stride, a.out_k_stride = int(out_outer_stride), int(out_k_stride)
    a.K, a.R, a.R2, a.C, a.outer_per_sample, a.act = int(K), int(R), int(R2), int(C), int(outer_per_sample), int(act)
    a.dtype = _code(x)
    a.up_identity = int(up_identity)       # True / 1: where it pays; 2: every size that has the form (tests)
    tok = _begin()
    check(lib.afldm_sep_pass(ctypes.byref(a), stream_ptr()), "sep_pass")
    lines = outer_count * inner_count
    _end(tok, "sep_pass", 2.0 * lines * K * R + (2.0 * lines * R * R2 if R2 else 0.0),
         lines * (K + (R2 if R2 else R)) * x.element_size())
    return y


def gn_table(stats, gamma, beta, B, C, G, HW, eps):
    """[B, C, 2] (scale, shift) table from GroupNorm partial sums (for the large-plane passes)."""
    out = torch.empty((B, C, 2), dtype=torch.float32, device=stats.st1.device)
    check(lib.afldm_gn_table(ptr(stats.st1), stats.S1, ptr(gamma), ptr(beta), ptr(out), B, C, G, HW, float(eps),
                             stream_ptr()), "gn_table")
    return out


_UP_IDENTITY = os.environ.get("AFLDM_NO_UP_IDENTITY", "0") != "1"


def _af_act_large(x, stats, gamma, beta, G, eps, out):
    """WarpedNonlinearity on planes too large for LDS (N >= 64): up-H, (up-W, SiLU, down-W) chained,
    down-H — three MFMA passes through HBM with intermediates in the activation dtype."""
    B, N, _, C = x.shape
    U, D = filter_matrices(N, x.device)
    table = gn_table(stats, gamma, beta, B, C, G, N * N, eps) if stats is not None else None
    t1 = torch.empty((B, 2 * N, N, C), dtype=x.dtype, device=x.device)
    sep_pass(x, t1, U, N, 2 * N, B, N * C, N * N * C, N * C, 2 * N * N * C, N * C,
             gn_table=table, C=C, outer_per_sample=1)
    v = torch.empty((B, 2 * N, N, C), dtype=x.dtype, device=x.device)
    sep_pass(t1, v, U, N, 2 * N, B * 2 * N, C, N * C, C, N * C, C, M2=D, R2=N, up_identity=_UP_IDENTITY)
    del t1
    sep_pass(v, out, D, 2 * N, N, B, N * C, 2 * N * N * C, N * C, N * N * C, N * C)
    return out


def _resample_large(x, M, R, out):
    """y = M x M^T per plane via two MFMA passes (planes N >= 32 of the AF-VAE)."""
    B, N, _, C = x.shape
    tmp = torch.empty((B, R, N, C), dtype=x.dtype, device=x.device)
    sep_pass(x, tmp, M, N, R, B, N * C, N * N * C, N * C, R * N * C, N * C)
    sep_pass(tmp, out, M, N, R, B * R, C, N * C, C, R * C, C)
    return out


def softmax_rows(x, scale=1.0, out=None):
    _dev(x, "x")
    cols = x.shape[-1]
    rows = x.numel() // cols
    if out is None:
        out = torch.empty_like(x)
    check(lib.afldm_softmax_rows(ptr(x), ptr(out), rows, cols, float(scale), _code(x), stream_ptr()), "softmax_rows")
    return out


_BATCHED_DENSE_ATTN = os.environ.get("AFLDM_NO_BATCHED_DENSE_ATTN", "0") != "1"


def attention_dense(q, k, vt, scale):
    """Single-head attention with a large head_dim (the VAE mid block: d = 512, T = 1024) as two
    per-sample GEMMs around a row softmax.  q, k: [B, T, C] contiguous; vt: [B, C, T]."""
    B, T, C = q.shape
    Tk = k.shape[1]
    if k.shape[0] != B or tuple(k.shape) != (B, Tk, C) or tuple(vt.shape) != (B, C, Tk):
        raise RuntimeError(f"afldm_amd: attention_dense needs k [B, Tk, C] and vt [B, C, Tk] with q's batch; got q "
                           f"{tuple(q.shape)}, k {tuple(k.shape)}, vt {tuple(vt.shape)} (repeat shared K / V over the batch)")
    _dev(q, "q"), _dev(k, "k"), _dev(vt, "vt")
    out = torch.empty_like(q)
    if (_BATCHED_DENSE_ATTN and q.dtype == torch.bfloat16 and Tk == T and T % 128 == 0 and C % 64 == 0
            and B * T * max(T, C) * 2 < (1 << 31)):
        # the whole batch in three launches: sample b's K_b / V_b^T are the "weights" of its rows (afldm_conv_args.w_batch_stride)
        try:
            s = conv2d(q.view(B, T, 1, C), k[0].view(T, 1, 1, C), w_batch_stride=T * C)      # [B, T, 1, T] scores
        except RuntimeError as e:
            if "w_batch_stride" not in str(e):
                raise
            s = None       # the plan for this shape splits K / has a tile that straddles samples: per-sample launches below
        if s is not None:
            p = softmax_rows(s.view(B * T, T), scale)
            try:
                conv2d(p.view(B, T, 1, T), vt[0].view(C, 1, 1, T), w_batch_stride=C * T, out=out.view(B, T, 1, C))
                return out
            except RuntimeError as e:
                if "w_batch_stride" not in str(e):
                    raise
    for b in range(B):
        s = conv2d(q[b].view(1, T, 1, C), k[b].view(Tk, 1, 1, C))               # [1, T, 1, Tk] scores
        p = softmax_rows(s.view(T, Tk), scale)
        conv2d(p.view(1, T, 1, Tk), vt[b].view(C, 1, 1, Tk), out=out[b].view(1, T, 1, C))
    return out


def lpf_matrix(N, device):
    key = ("lpf", N, str(device))
    if key not in _FILTER_CACHE:
        _FILTER_CACHE[key] = _lib.filter_matrix(2, N).to(device)
    return _FILTER_CACHE[key]


def af_resample(x, M, out=None, workspace=None):
    """y = M x M^T per plane: [B,N,N,C] -> [B,R,R,C] with M [R,N] device fp32."""
    _dev(x, "x")
    B, N, N2, C = x.shape
    assert N == N2 and M.shape[1] == N
    R = M.shape[0]
    if out is None:
        out = torch.empty((B, R, R, C), dtype=x.dtype, device=x.device)
    if workspace is None:
        workspace = torch.empty(B * R * N * C, dtype=torch.float32, device=x.device)
    check(lib.afldm_af_resample(ptr(x), ptr(M), ptr(out), ptr(workspace), B, N, C, R, _code(x), stream_ptr()),
          "af_resample")
    return out


def af_resample_hw(x, Mh, Mw, out=None, workspace=None):
    """y = Mh x Mw^T per plane (different matrices along H and W): [B,N,N,C] -> [B,R,R,C]."""
    _dev(x, "x")
    B, N, N2, C = x.shape
    assert N == N2 and Mh.shape == Mw.shape and Mh.shape[1] == N
    R = Mh.shape[0]
    if out is None:
        out = torch.empty((B, R, R, C), dtype=x.dtype, device=x.device)
    if workspace is None:
        workspace = torch.empty(B * R * N * C, dtype=torch.float32, device=x.device)
    check(lib.afldm_af_resample_hw(ptr(x), ptr(Mh), ptr(Mw), ptr(out), ptr(workspace), B, N, C, R, _code(x),
                                   stream_ptr()), "af_resample_hw")
    return out


def masked_metrics(a, b, mask):
    """afldm_masked_metrics: [B, 6] fp32 = per-sample (sum ((a-b) m)^2, sum m, max a m, min a m, max b m, min b m)."""
    _dev(a, "a")
    assert a.shape == b.shape
    if a.dtype not in (torch.float32, torch.bfloat16) or b.dtype != a.dtype:
        a, b = a.float(), b.float()
    a, b = a.contiguous(), b.contiguous()
    mask = mask.to(device=a.device, dtype=torch.float32)
    m = mask.expand_as(a).contiguous()
    B = a.shape[0]
    out = torch.empty((B, 6), dtype=torch.float32, device=a.device)
    check(lib.afldm_masked_metrics(ptr(a), ptr(b), ptr(m), ptr(out), B, a.numel() // B, _code(a), stream_ptr()),
          "masked_metrics")
    # The reference divides by mask.sum((1,2,3)) of the mask's OWN shape (metrics.py:6-7): a broadcast
    # [B,1,H,W] mask (what ImageShifter returns) counts each pixel once, not once per channel.  The kernel
    # summed the expanded mask: `rep` copies of every mask element.
    if mask.ndim == a.ndim:
        rep = (a.numel() // B) // max(mask.numel() // mask.shape[0], 1)
        if rep > 1:
            out[:, 1] /= rep
    return out


# ----------------------------------------------------------------------------- upfirdn2d (NCHW planes)
def upfirdn2d(x, f, upx=1, upy=1, downx=1, downy=1, padx0=0, padx1=0, pady0=0, pady1=0, flip_filter=False, gain=1.0,
              out=None):
    """afldm_upfirdn2d on an NCHW CUDA tensor (fp32 / bf16) with a 2-D device fp32 filter [fh, fw]."""
    _dev(x, "x")
    _dev(f, "f")
    assert x.ndim == 4 and f.ndim == 2 and f.dtype == torch.float32
    B, C, H, W = x.shape
    fh, fw = f.shape
    outW = lib.afldm_upfirdn2d_out_size(W, upx, downx, padx0, padx1, fw)
    outH = lib.afldm_upfirdn2d_out_size(H, upy, downy, pady0, pady1, fh)
    assert outW > 0 and outH > 0, "upfirdn2d: the padded / cropped up-sampled plane is smaller than the filter"
    if out is None:
        out = torch.empty((B, C, outH, outW), dtype=x.dtype, device=x.device)
    if B * C == 0:
        return out
    tok = _begin()
    check(lib.afldm_upfirdn2d(ptr(x), ptr(f), ptr(out), B * C, H, W, fh, fw, upx, upy, downx, downy, padx0, padx1,
                              pady0, pady1, int(bool(flip_filter)), float(gain), _code(x), stream_ptr()), "upfirdn2d")
    _end(tok, "upfirdn2d", 2.0 * out.numel() * ((fh + upy - 1) // upy) * ((fw + upx - 1) // upx),
         (x.numel() + out.numel()) * x.element_size())
    return out


# ----------------------------------------------------------------------------- conv / linear
_SYNC = {}
_SYNC_SPARE = {}


SYNC_WORDS = 32768        # int32 words of a sync buffer: [0, 8192) split-K tile counters, [8192] split-K error word, [8193] error word
                          # of the in-launch hand-overs (attention phase C, merged launches: 1 = gave up waiting, 2 = cluster
                          # straddles XCDs, 3 = a barrier of the experimental cooperative trunk timed out), [8200, 8201] that
                          # trunk's grid barrier, [16384, 32768) hand-over counters (one 128-byte line per sample)
_SYNC_OVERRIDE = None
_SCOPED_SYNC = []


def new_sync_buffer(device):
    """A private, zeroed sync buffer (see _sync_words) for ONE stream of launches - a DenoiseEngine (each of its
    branches) owns one, so that graphs captured on torch's shared capture stream and later replayed CONCURRENTLY on
    different streams never share counters."""
    buf = torch.zeros(SYNC_WORDS, dtype=torch.int32, device=device)
    import weakref
    _SCOPED_SYNC[:] = [r for r in _SCOPED_SYNC if r() is not None]
    _SCOPED_SYNC.append(weakref.ref(buf))
    return buf


class sync_scope:
    """with sync_scope(buf): every launch issued inside takes `buf` as its sync words instead of the per-stream one."""

    def __init__(self, buf):
        self.buf = buf

    def __enter__(self):
        global _SYNC_OVERRIDE
        self.prev, _SYNC_OVERRIDE = _SYNC_OVERRIDE, self.buf
        return self.buf

    def __exit__(self, *a):
        global _SYNC_OVERRIDE
        _SYNC_OVERRIDE = self.prev


def _sync_words(device):
    """Zero-initialised counter words for the in-kernel split-K reduction (afldm_conv_args.sync) and the cluster
    hand-overs of the merged launches: one buffer per (device, stream) - launches of ONE stream share it in stream order
    and each leaves it zero; concurrent streams (DenoiseEngine branches) must not share counters.  An
    engine scopes its own buffer over its launches (sync_scope): captured graphs all run on torch's one capture stream
    at capture time but may replay concurrently later."""
    if _SYNC_OVERRIDE is not None:
        return _SYNC_OVERRIDE
    key = (str(device), torch.cuda.current_stream(device).cuda_stream)
    if key not in _SYNC:
        # never allocate under graph capture (the buffer would live in that graph's private pool while this dict hands
        # it to later graphs and eager launches): streams first seen during a capture - torch's capture
        # stream, the engine's branch streams - take a buffer from a spare list filled on the first eager call
        spares = _SYNC_SPARE.setdefault(str(device), [])
        if torch.cuda.is_current_stream_capturing():
            if not spares:
                raise RuntimeError("afldm_amd: no pre-allocated sync buffer left for a stream first seen under graph "
                                   "capture; run one eager step (warm-up) before capturing")
            _SYNC[key] = spares.pop()
        else:
            _SYNC[key] = new_sync_buffer(device)
            while len(spares) < 8:
                spares.append(new_sync_buffer(device))
    return _SYNC[key]


def fused_splitk_error(device=None):
    """True if an in-kernel split-K reduction (AFLDM_FUSED_SPLITK) gave up waiting for a slice since the last call;
    reads and clears the error word of every sync buffer of the device (synchronises)."""
    hit = False
    for (dev, _), buf in _SYNC.items():
        if device is None or dev == str(device):
            if int(buf[8192].item()):
                hit = True
                buf[8192] = 0
    return hit


def conv_args(x1, w, bias=None, x2=None, temb=None, temb_stride=0, residual=None, out=None, out_mode=0,
              workspace=None, y_ld=None, out2=None, split_n=0, temb_mod=0):
    """Build the afldm_conv_args struct (keeps references to the tensors alive in `.keep`)."""
    C1, x2, C2 = _cat_args(x1, x2)
    _dev(w, "w")
    Cout, KS = w.shape[0], w.shape[1]
    assert w.shape[3] == C1 + C2, f"weight Cin {w.shape[3]} != {C1}+{C2}"
    if x1.ndim == 4:
        B, H, W_ = x1.shape[:3]
    else:                      # [rows, C] or [B, T, C]: a linear layer
        B, H, W_ = x1.numel() // C1, 1, 1
    a = ConvArgs()
    a.x1, a.x2, a.w, a.bias = ptr(x1), ptr(x2), ptr(w), ptr(bias)
    a.temb, a.residual, a.y = ptr(temb), ptr(residual), ptr(out)
    a.workspace = ptr(workspace)
    a.workspace_bytes = 0 if workspace is None else workspace.numel() * workspace.element_size()
    a.C1, a.C2, a.B, a.H, a.W, a.Cout, a.KS = C1, C2, B, H, W_, Cout, KS
    a.temb_stride = int(temb_stride)
    a.res_ld = Cout if residual is None else residual.shape[-1]
    a.y_ld = Cout if y_ld is None else int(y_ld)
    a.out_mode = int(out_mode)
    a.dtype = _code(x1)
    a.y2, a.split_n = ptr(out2), int(split_n)
    a.temb_mod = int(temb_mod)
    if out2 is not None and y_ld is None:
        a.y_ld = int(split_n)
    sync = _sync_words(x1.device)
    a.sync, a.sync_bytes = ptr(sync), sync.numel() * 4
    a.keep = (x1, x2, w, bias, temb, residual, out, workspace, sync)
    a.x_layout = 1 if is_c8(x1) else 0
    assert not is_c8(x2) and not is_c8(residual), "8-channel-block tensors are convolution inputs only"
    return a


_CONV_NORM = os.environ.get("AFLDM_NO_CONV_NORM", "0") != "1"


def conv2d_c8_ok(x1, w, bias=None, temb=None, temb_stride=0, residual=None, want_stats=True):
    """True when afldm_conv2d on this problem reads / writes 8-channel blocks (one halo-patch launch, bf16, 16^2 / 32^2).
    The probe carries what the real launch will carry - the time-embedding stride and a statistics output (the blocked layouts
    need the statistics formed in the epilogue, which depends on both: ADVICE r05)."""
    if not _C8 or x1.ndim != 4 or x1.dtype != torch.bfloat16 or x1.shape[1] != x1.shape[2] or x1.shape[1] not in (16, 32):
        return False
    key = (tuple(x1.shape), tuple(w.shape), temb is not None, int(temb_stride), residual is not None, bool(want_stats), str(x1.device))
    if key not in _C8_OK:
        a = conv_args(x1, w, bias, None, temb, temb_stride, residual, None)
        a.y = ptr(x1)                 # (a non-NULL, aligned placeholder: the queries do not dereference it)
        if want_stats and w.shape[0] % 4 == 0:
            a.stats_out = ptr(x1)
        a.x_layout = 0
        # (a plan that would split K given the room is not the one-launch halo kernel)
        _C8_OK[key] = (not lib.afldm_conv2d_workspace(ctypes.byref(a))) and bool(lib.afldm_conv2d_c8_ok(ctypes.byref(a)))
    return _C8_OK[key]


_C8_OK = {}


def conv2d(x1, w, bias=None, x2=None, temb=None, temb_stride=0, residual=None, out=None, out_mode=0,
           workspace=None, want_stats=False, temb_mod=0, w_batch_stride=0, norm_out=None, out_c8=False):
    """stride-1 'same' conv (KS in {1,3}) / linear on NHWC input with packed OHWI weights.
    out_mode 1 returns the channel-major [B, Cout, H*W] tensor (V^T for attention).
    want_stats: also emit the per-channel GroupNorm partial sums of the output (from the GEMM
    epilogue where possible) and attach them to the returned tensor as `.gn_partial`."""
    Cout = w.shape[0]
    if out is None:
        if out_mode == 0:
            out = torch.empty(tuple(x1.shape[:-1]) + (Cout,), dtype=x1.dtype, device=x1.device)
        else:
            B = x1.shape[0]
            out = torch.empty((B, Cout, x1.numel() // (B * x1.shape[-1])), dtype=x1.dtype, device=x1.device)
    a = conv_args(x1, w, bias, x2, temb, temb_stride, residual, out, out_mode, workspace, temb_mod=temb_mod)
    if out_c8:
        a.y_layout = 1
        out.c8 = True
    a.w_batch_stride = int(w_batch_stride)   # per-sample weights (elements between samples' weight tensors; `w` = sample 0's)
    if out_mode == 1 and x1.ndim == 3:      # [B, T, C] tokens: treat T as the pixel axis
        a.B, a.H, a.W = x1.shape[0], x1.shape[1], 1
    if workspace is None:
        need = lib.afldm_conv2d_workspace(ctypes.byref(a))
        if need:
            workspace = torch.empty(need // 4, dtype=torch.float32, device=x1.device)
            a.workspace, a.workspace_bytes = ptr(workspace), need
    st = None
    if want_stats and out_mode == 0 and Cout % 4 == 0:
        S = lib.afldm_conv2d_stats_splits(ctypes.byref(a))
        st = torch.empty((a.B, S, Cout, 2), dtype=torch.float32, device=x1.device)
        a.stats_out = ptr(st)
    hn = None
    if norm_out is not None and _CONV_NORM and st is not None:
        # norm_out = (gamma, beta, G, eps) of the GroupNorm that consumes this output next: applied by the convolution's own
        # epilogue where a tile holds a whole sample and whole groups (afldm_conv2d_norm_ok); the result rides on the output
        # as `.norm_applied`
        gamma, beta, G, eps = norm_out
        a.norm_gamma, a.norm_beta, a.norm_groups, a.norm_eps = ptr(gamma), ptr(beta), int(G), float(eps)
        if lib.afldm_conv2d_norm_ok(ctypes.byref(a)):
            hn = torch.empty_like(out)
            a.y_norm = ptr(hn)
            a.keep = a.keep + (hn, gamma, beta)
    tok = _begin()
    check(lib.afldm_conv2d(ctypes.byref(a), stream_ptr()), "conv2d")
    if hn is not None:
        out.norm_applied = hn
    if st is not None:
        if st.shape[1] > _MAX_SPLITS and st.shape[1] % _MAX_SPLITS == 0:
            # one split per 128-pixel tile is 512 per sample on a 256^2 plane: fold before the consumers walk them
            folded = torch.empty((a.B, _MAX_SPLITS, Cout, 2), dtype=torch.float32, device=x1.device)
            check(lib.afldm_gn_fold(ptr(st), st.shape[1], ptr(folded), _MAX_SPLITS, a.B, Cout, stream_ptr()), "gn_fold")
            st = folded
        out.gn_partial = st
    if tok is not None:
        M, Ct = a.B * a.H * a.W, a.C1 + a.C2
        es = x1.element_size()
        kind = "conv3x3" if a.KS == 3 else ("conv1x1" if a.H * a.W > 1 and x1.ndim == 4 else "linear")
        # algorithmic bytes: input + weights + output (+ the residual read), as tools/replay_conv3x3.py counts them
        keep = (a, st, out, workspace)
        _end(tok, kind, 2.0 * M * a.Cout * a.KS * a.KS * Ct,
             (M * Ct + a.Cout * a.KS * a.KS * Ct + M * a.Cout * (2 if residual is not None else 1)) * es,
             replay=lambda keep=keep: conv2d_launch(keep[0]))
    return out


_ACTCONV = os.environ.get("AFLDM_NO_ACTCONV", "0") != "1"
# plane sizes whose norm -> activation -> conv pairs ResnetBlock2D issues as merged launches.  EMPTY by default: built, bit-identical,
# and measured SLOWER in the step (profiles/r05/actconv_ab.txt: 4.99 -> 5.01 ms with the 16^2 pairs, 5.09 with 32^2 too);
# AFLDM_ACTCONV_N=16,32 turns them on (tests call the op directly)
_ACTCONV_N = tuple(int(v) for v in os.environ.get("AFLDM_ACTCONV_N", "").split(",") if v)
# per-site policy: "N:Cin,..." - the norm1 -> activation -> conv1 pair of the blocks whose (plane size, input channels) is listed
# runs as the merged launch, everything else separately (the isolated A/B showed the merged act -> conv ahead only on the
# concatenated inputs of the 16^2 up blocks: profiles/r05/actconv_ab.txt)
_ACTCONV_SITES = frozenset(tuple(int(x) for x in v.split(":")) for v in os.environ.get("AFLDM_ACTCONV_SITES", "").split(",") if v)


def act_conv_act(x1, x2, pre, w, bias=None, temb=None, temb_stride=0, residual=None, want_stats=False, post=None):
    """[pre: GroupNorm-apply -> WarpedNonlinearity(SiLU) ->] 3x3 conv [-> post: GroupNorm-apply -> WarpedNonlinearity] as
    ONE launch (afldm_act_conv_act, csrc/actconv.hip).
      pre  = None (x1 IS the convolution's input) or (stats, gamma, beta, G, eps) applied to the virtual concat x1 | x2
             (stats None: activation without normalisation);
      post = None or (gamma, beta, G, eps): the activation of the convolution's own output with the statistics its
             epilogue writes (gamma None: no normalisation).
    Returns (conv_out, post_out) - conv_out carries `.gn_partial` when want_stats or post needs them and `.act_input`
    (the activated input) when pre is given - or None when there is no merged kernel for the chain (the caller runs the
    separate ops).  Bit-identical to them."""
    if (not _ACTCONV or not _exp.available() or x1.ndim != 4 or x1.dtype != torch.bfloat16 or x1.shape[1] != x1.shape[2] or x1.shape[1] not in (16, 32)
            or (pre is None and post is None)):
        return None
    C1, x2, C2 = _cat_args(x1, x2)
    B, N = x1.shape[0], x1.shape[1]
    Cout = w.shape[0]
    if w.shape[1] != 3 or w.shape[3] != C1 + C2 or (pre is None and x2 is not None):
        return None
    act_out = torch.empty((B, N, N, C1 + C2), dtype=x1.dtype, device=x1.device) if pre is not None else x1
    out = torch.empty((B, N, N, Cout), dtype=x1.dtype, device=x1.device)
    a = conv_args(act_out, w, bias, None, temb, temb_stride, residual, out)
    if a.sync_bytes < (16384 + 64 * B) * 4:
        return None
    need = lib.afldm_conv2d_workspace(ctypes.byref(a))
    workspace = None
    if need:                                   # (the plan would split K given the room: no merged kernel then, as conv2d decides)
        workspace = torch.empty(need // 4, dtype=torch.float32, device=x1.device)
        a.workspace, a.workspace_bytes = ptr(workspace), need
    st = None
    post_norm = post is not None and post[0] is not None
    if (want_stats or post_norm) and Cout % 4 == 0:
        S = lib.afldm_conv2d_stats_splits(ctypes.byref(a))
        st = torch.empty((B, S, Cout, 2), dtype=torch.float32, device=x1.device)
        a.stats_out = ptr(st)
    U, D = filter_matrices(N, x1.device)
    packed = packed_filters(N, x1.dtype, x1.device)
    aa = pp = None
    keep = [a, x1, x2, U, D, packed, act_out, out, st, workspace]
    if pre is not None:
        stats, gamma, beta, G, eps = pre
        aa = AfActArgs()
        aa.x1, aa.x2, aa.C1, aa.C2 = ptr(x1), ptr(x2), C1, C2
        p1, S1, p2, S2 = _stats_ptrs(stats)
        aa.stats1, aa.S1, aa.stats2, aa.S2 = p1, S1, p2, S2
        aa.gamma, aa.beta, aa.G, aa.eps = ptr(gamma), ptr(beta), int(G), float(eps)
        aa.U, aa.D, aa.packed = ptr(U), ptr(D), ptr(packed)
        keep += [aa, stats, gamma, beta]
    post_out = None
    if post is not None:
        gamma2, beta2, G2, eps2 = post
        if post_norm and st is None:
            return None
        pp = AfActArgs()
        pp.x1, pp.x2, pp.C1, pp.C2 = ptr(out), None, Cout, 0
        pp.stats1, pp.S1, pp.stats2, pp.S2 = (ptr(st), st.shape[1], None, 0) if post_norm else (None, 0, None, 0)
        pp.gamma, pp.beta, pp.G, pp.eps = ptr(gamma2), ptr(beta2), int(G2), float(eps2)
        pp.U, pp.D, pp.packed = ptr(U), ptr(D), ptr(packed)
        post_out = torch.empty((B, N, N, Cout), dtype=x1.dtype, device=x1.device)
        keep += [pp, gamma2, beta2, post_out]
    ra = ctypes.byref(aa) if aa is not None else None
    rp = ctypes.byref(pp) if pp is not None else None
    if not _exp.lib().afldm_act_conv_act_merged(ra, ctypes.byref(a), rp):
        return None
    keep = tuple(keep)
    tok = _begin()
    check(_exp.lib().afldm_act_conv_act(ra, ctypes.byref(a), rp, ptr(post_out), stream_ptr()), "act_conv_act")
    if st is not None:
        out.gn_partial = st
    if tok is not None:
        M, Ct, es = B * N * N, C1 + C2, x1.element_size()
        # algorithmic work of the chain: the activations' dense separable form + the convolution; bytes: every tensor the
        # separate launches would read / write once (raw in, activated in, weights, output (+ residual), activated out)
        fl = 2.0 * M * Cout * 9 * Ct + (24.0 * N ** 3 * B * Ct if pre is not None else 0.0) + (24.0 * N ** 3 * B * Cout if post is not None else 0.0)
        by = (M * Ct * (2 if pre is not None else 1) + Cout * 9 * Ct + M * Cout * ((2 if residual is not None else 1) + (1 if post is not None else 0))) * es
        name = ("af_act_" if pre is not None else "") + "conv3x3" + ("_af_act" if post is not None else "") + f"_N{N}"

        def replay(ra=ra, rp=rp, a=a, post_out=post_out, keep=keep):
            check(_exp.lib().afldm_act_conv_act(ra, ctypes.byref(a), rp, ptr(post_out), stream_ptr()), "act_conv_act")
        _end(tok, name, fl, by, replay=replay)
    if pre is not None:
        out.act_input = act_out          # the activated tensor (kept alive with the output; tests read it)
    return out, post_out


def af_act_conv2d(x1, x2, stats, gamma, beta, G, eps, w, bias=None, temb=None, temb_stride=0, residual=None,
                  want_stats=False):
    """[GroupNorm-apply ->] WarpedNonlinearity(SiLU) -> 3x3 conv as ONE launch: act_conv_act without the trailing
    activation.  Returns the convolution's output or None (no merged kernel: the caller runs af_act + conv2d)."""
    got = act_conv_act(x1, x2, (stats, gamma, beta, G, eps), w, bias, temb, temb_stride, residual, want_stats)
    return None if got is None else got[0]


def actconv_error(device=None):
    """Error word of the merged launches' cluster hand-over (0 = fine; 1 = a workgroup gave up waiting for its cluster,
    2 = a workgroup did not run on the XCD its id implies) over every sync buffer seen so far; synchronises, clears."""
    worst = 0
    bufs = [b for (dev, _), b in _SYNC.items() if device is None or dev == str(device)] + [r() for r in _SCOPED_SYNC if r() is not None]
    for buf in bufs:
        v = int(buf[8193].item())
        if v:
            worst = max(worst, v)
            buf[8193] = 0
    return worst


def conv2d_slabs(x1, w, x2=None):
    """The split-K slabs of conv2d(x1 | x2, w) WITHOUT the reduction launch (afldm_conv_args.defer_reduce): returns
    (slabs fp32 [nslab, M, Cout], nslab), or None when the plan for this shape does not split K (the caller then runs
    the ordinary conv2d).  Bias, time embedding and residual are NOT applied: the consumer (af_act_slabs) adds them."""
    Cout = w.shape[0]
    out = torch.empty(tuple(x1.shape[:-1]) + (Cout,), dtype=x1.dtype, device=x1.device)      # (never written when K is split)
    a = conv_args(x1, w, None, x2, out=out)
    need = lib.afldm_conv2d_workspace(ctypes.byref(a))
    if not need:
        return None
    workspace = torch.empty(need // 4, dtype=torch.float32, device=x1.device)
    a.workspace, a.workspace_bytes = ptr(workspace), need            # (the plan only splits K when it is given the room)
    code = lib.afldm_conv2d_variant(ctypes.byref(a))
    nslab = (code >> 8) & 255 if code >= 0 else 1
    if code < 0 or nslab <= 1 or (code >> 16) & 1 or need < nslab * a.B * a.H * a.W * Cout * 4:
        return None
    a.defer_reduce = 1
    a.keep = a.keep + (workspace,)
    tok = _begin()
    check(lib.afldm_conv2d(ctypes.byref(a), stream_ptr()), "conv2d(slabs)")
    if tok is not None:
        M, Ct = a.B * a.H * a.W, a.C1 + a.C2
        es = x1.element_size()
        kind = "conv3x3" if a.KS == 3 else ("conv1x1" if a.H * a.W > 1 and x1.ndim == 4 else "linear")
        keep = (a, workspace, out)
        _end(tok, kind, 2.0 * M * a.Cout * a.KS * a.KS * Ct, (M * Ct + a.Cout * a.KS * a.KS * Ct + M * a.Cout) * es,
             replay=lambda keep=keep: conv2d_launch(keep[0]))
    return workspace[:nslab * a.B * a.H * a.W * Cout].view(nslab, a.B * a.H * a.W, Cout), nslab


def af_act_slabs(slabs, nslab, bias, temb, temb_stride, gamma, beta, G, eps, B, N, C, dtype, residual=None, want_raw=False,
                 act=True):
    """A convolution's split-K slabs straight into the following GroupNorm on 2x2 / 4x4 planes (afldm_af_act_slabs):
    act=True -> GroupNorm + WarpedNonlinearity (conv1 -> norm2 of a resnet), act=False -> GroupNorm only (conv2 ->
    Attention.group_norm).  Returns the [B, N, N, C] result, or (result, raw) with the finished convolution output
    itself when want_raw."""
    const = act == 2       # N = 2: the plane-constant activation stored once, [B, C] tagged `.const2` (af_act out_const)
    assert not const or N == 2
    out = torch.empty((B, C) if const else (B, N, N, C), dtype=dtype, device=slabs.device)
    raw = torch.empty((B, N, N, C), dtype=dtype, device=slabs.device) if want_raw else None
    U, D = filter_matrices(N, slabs.device)
    tok = _begin()
    check(lib.afldm_af_act_slabs(ptr(slabs), int(nslab), ptr(bias), ptr(temb), int(temb_stride), ptr(residual), ptr(raw),
                                 ptr(gamma), ptr(beta), int(G), float(eps), 2 if const else (1 if act else 0), ptr(U), ptr(D),
                                 ptr(out), B, C, N, _code(out), stream_ptr()), "af_act_slabs")
    if const:
        out.const2 = True
    _end(tok, f"af_act_N{N}" if act else "gn_apply", 24.0 * N ** 3 * B * C if act else 0.0,
         B * N * N * C * (4 * nslab + out.element_size() * (2 if want_raw else 1)))
    return (out, raw) if want_raw else out


# smallest batch the fused conv1 -> norm2 -> activation launch of the 2x2 level is used for: a workgroup owns a whole GroupNorm group
# (96 weight rows x all of K = 147 - 295 KB, which one CU pulls at ~30 GB/s whatever the order of the workgroups), i.e. 32 workgroups
# per 16 samples.  Same-box, alternating, ms/step with / without: batch 64 4.610 / 4.637, 4.628 / 4.664, 4.823 / 4.843 (three boxes);
# batch 32 3.217 / 3.202, 3.268 / 3.247; batch 16 2.620 / 2.600; batch 8 2.298 / 2.290; batch 1 2.026 / 2.002
# (profiles/r06/ab_dense2_r06c.log, ab_skinny_dense2_policy_r06d.log, ab_dense2_xcd_r06e.log): k_skinny's 192 x 24 KB wins below 64
_DENSE2_MIN_B = int(os.environ.get("AFLDM_DENSE2_MIN_B", "64"))


def conv2x2_const_norm_act_ok(Cin, Cout, G, dtype, batch=None):
    """True when afldm_conv2x2_const_norm_act has a kernel for the shape and (batch given) the policy picks it."""
    if batch is not None and batch < _DENSE2_MIN_B:
        return False
    return bool(lib.afldm_conv2x2_const_norm_act_supported(int(Cin), int(Cout), int(G), DTYPE_CODE[dtype]))


def conv2x2_const_norm_act(a, w_cm, bias, temb, temb_stride, gamma, beta, G, eps):
    """conv1 + temb -> norm2 -> WarpedNonlinearity of a ResnetBlock2D on 2x2 planes in ONE launch (afldm_conv2x2_const_norm_act):
    a [B, Cin] the plane-constant first activation, w_cm [4 Cout, 1, 1, Cin] the tap-summed dense weights with channel-major rows
    (row = 4 n + pixel), bias [Cout] fp32, temb the block's time_emb_proj slice.  Returns the plane-constant second activation
    [B, Cout] (tagged `.const2`)."""
    _dev(a, "a")
    B, Cin = a.shape
    Cout = w_cm.shape[0] // 4
    out = torch.empty((B, Cout), dtype=a.dtype, device=a.device)
    U, D = filter_matrices(2, a.device)
    tok = _begin()
    check(lib.afldm_conv2x2_const_norm_act(ptr(a), ptr(w_cm), ptr(bias), ptr(temb), int(temb_stride), ptr(gamma), ptr(beta), int(G),
                                           float(eps), ptr(U), ptr(D), ptr(out), B, Cin, Cout, _code(a), stream_ptr()),
          "conv2x2_const_norm_act")
    _end(tok, "linear", 2.0 * B * 4 * Cout * Cin, (B * Cin + 4 * Cout * Cin + B * Cout) * a.element_size())
    out.const2 = True
    return out


def conv_out_fused(x, w, bias, gamma, beta, G, eps):
    """conv_norm_out -> SiLU -> conv_out (3x3, <= 4 couts) of the UNet tail in one launch (afldm_conv_out_fused);
    returns None when the shape is not covered (the caller runs gn_apply + conv2d)."""
    if (x.dtype != torch.bfloat16 or x.ndim != 4 or x.shape[1] != 32 or x.shape[2] != 32 or x.shape[3] not in (64, 128, 192)
            or w.shape[0] > 4 or w.shape[1] != 3 or os.environ.get("AFLDM_NO_CONV_OUT_FUSED")
            or G > 64 or G <= 0 or x.shape[3] % G or not x.is_contiguous()):      # the C side's own constraints
        return None
    B, N, _, C = x.shape
    Cout = w.shape[0]
    st = _tensor_stats(x, None)
    out = torch.empty((B, N, N, Cout), dtype=x.dtype, device=x.device)
    tok = _begin()
    check(lib.afldm_conv_out_fused(ptr(x), ptr(st), st.shape[1], ptr(gamma), ptr(beta), int(G), float(eps), ptr(w), ptr(bias),
                                   ptr(out), B, N, C, Cout, _code(x), stream_ptr()), "conv_out_fused")
    _end(tok, "conv_out", 2.0 * B * N * N * Cout * 9 * C, (B * N * N * (C + Cout) + Cout * 9 * C) * x.element_size())
    return out


def conv_workspace_bytes(a):
    return lib.afldm_conv2d_workspace(ctypes.byref(a))


def linear_split(x, w, bias, split_n, workspace=None):
    """One GEMM over tokens x [B, T, C] with weight rows [0, split_n) -> token-major [B, T, split_n]
    and rows [split_n, Cout) -> channel-major [B, Cout - split_n, T] (fused Q|K|V projection)."""
    _dev(x, "x")
    B, T, C = x.shape
    Cout = w.shape[0]
    y = torch.empty((B, T, split_n), dtype=x.dtype, device=x.device)
    y2 = torch.empty((B, Cout - split_n, T), dtype=x.dtype, device=x.device)
    a = conv_args(x, w, bias, out=y, out2=y2, split_n=split_n)
    a.B, a.H, a.W = B, T, 1
    if workspace is None:
        need = lib.afldm_conv2d_workspace(ctypes.byref(a))
        if need:
            workspace = torch.empty(need // 4, dtype=torch.float32, device=x.device)
            a.workspace, a.workspace_bytes = ptr(workspace), need
    tok = _begin()
    check(lib.afldm_conv2d(ctypes.byref(a), stream_ptr()), "conv2d(split)")
    _end(tok, "linear", 2.0 * B * T * Cout * C, (B * T * C + Cout * C + B * T * Cout) * x.element_size())
    return y, y2


def conv2d_launch(a):
    check(lib.afldm_conv2d(ctypes.byref(a), stream_ptr()), "conv2d")


# ----------------------------------------------------------------------------- attention
def attention(q, k, vt, heads, scale=None, out=None):
    """q [B,Tq,C], k [Bk,Tk,C] token-major (may be column slices of wider buffers: the leading
    dimension is taken from stride(1)); vt [Bk,C,Tk] channel-major; returns [B,Tq,C]."""
    _dev(vt, "vt")
    if not (q.is_cuda and k.is_cuda):
        raise RuntimeError("afldm_amd: attention inputs must live on an MI355X (cuda) device; there is no CPU path")
    B, Tq, C = q.shape
    Bk, Tk, _ = k.shape
    ldq, ldk = q.stride(1), k.stride(1)
    assert q.stride(2) == 1 and k.stride(2) == 1 and q.stride(0) == Tq * ldq and k.stride(0) == Tk * ldk
    d = C // heads
    if scale is None:
        scale = d ** -0.5
    if out is None:
        out = torch.empty((B, Tq, C), dtype=q.dtype, device=q.device)
    tok = _begin()
    check(lib.afldm_attention(ptr(q), ldq, ptr(k), ldk, ptr(vt), ptr(out), C, B, Bk, heads, Tq, Tk, d, float(scale),
                              _code(q), stream_ptr()), "attention")
    _end(tok, "attention", 4.0 * B * heads * Tq * Tk * d, (2 * B * Tq * C + 2 * Bk * Tk * C) * q.element_size())
    return out


_FUSED_ATTN = os.environ.get("AFLDM_NO_FUSED_ATTN", "0") != "1"
# below this many (sample, head) workgroups the chip is too empty with one workgroup per pair: the three-launch path wins.
# In-step A/B (profiles/r04/small_batch_tiles_ab.txt): 128 against 256: batch 8 2.385 -> 2.362 (the 16^2 level fuses), batch 12
# 2.741 -> 2.710, batch 16 2.741 -> 2.717, batch 24 3.485 -> 3.373 ms/step; 64: batch 8 2.394 (the 32^2 level loses)
_FUSED_ATTN_MIN_WGS = int(os.environ.get("AFLDM_FUSED_ATTN_MIN_WGS", "128"))
# fewest tokens per sample the fused launch is used for (in-step A/B decides between the 32^2 level only and 32^2 + 16^2)
_FUSED_ATTN_MIN_T = int(os.environ.get("AFLDM_FUSED_ATTN_MIN_T", "256"))


def attn_block_fused_ok(x, heads, G):
    """True when afldm_attn_block_fused has a kernel for tokens x [B, T, C] (bf16) and the policy wants it."""
    if not _FUSED_ATTN or x.dtype != torch.bfloat16:
        return False
    B, T, C = x.shape
    if B * heads < _FUSED_ATTN_MIN_WGS or T < _FUSED_ATTN_MIN_T:
        return False
    return bool(lib.afldm_attn_block_fused_supported(T, C, C // heads, int(G)))


# AFLDM_ATTN_FUSED_OUT=0: to_out + residual as its own launch behind the fused front end (A/B; the default carries them in
# the attention launch where afldm_attn_block_fused_out has a kernel: the 32 x 32 level, batch a multiple of 8)
_FUSED_ATTN_OUT = os.environ.get("AFLDM_ATTN_FUSED_OUT", "1") != "0"


_HANDOVER_DEVICES = {}


def handover_device_ok(device):
    """The in-launch hand-over of afldm_attn_block_fused_out relies on the dispatcher dealing consecutive workgroup ids round
    robin to the 8 XCDs of an UNPARTITIONED, UNMASKED MI355X (a sample's 8 head workgroups then share one XCD and its L2):
    256 CUs visible and no CU mask in the environment.  Anything else (a partitioned device, HSA_CU_MASK / ROC_GLOBAL_CU_MASK)
    takes the two-launch path."""
    key = str(device)
    if key not in _HANDOVER_DEVICES:
        masked = any(os.environ.get(v) for v in ("HSA_CU_MASK", "ROC_GLOBAL_CU_MASK", "HSA_CU_MASK_SKIP_INIT"))
        props = torch.cuda.get_device_properties(device)
        _HANDOVER_DEVICES[key] = (not masked) and props.multi_processor_count == 256
    return _HANDOVER_DEVICES[key]


def attn_block_fused_out_ok(x, heads, G):
    B, T, C = x.shape
    return bool(_FUSED_ATTN_OUT and handover_device_ok(x.device)
                and lib.afldm_attn_block_fused_out_supported(B, T, C, C // heads, int(G)))


def sync_errors(bufs, clear=True):
    """(split-K error, hand-over error) over sync buffers: words 8192 / 8193 (see SYNC_WORDS), read with ONE device
    synchronisation per buffer.  A non-zero hand-over word means a launch finished on incomplete data: with `clear` the
    error words AND the hand-over counter lines are zeroed so that later launches start from a clean buffer."""
    sk = ho = 0
    for buf in bufs:
        w = buf[8192:8194].cpu()
        a, b = int(w[0]), int(w[1])
        if (a or b) and clear:
            buf[8192:8194].zero_()
            buf[16384:].zero_()
        sk, ho = max(sk, a), max(ho, b)
    return sk, ho


def attn_block_fused_out(x, stats, gamma, beta, G, eps, w_qkv, bias_qkv, heads, scale, w_out, bias_out):
    """The whole attention block in ONE launch: attn_block_fused + to_out + residual.  Returns y [B, T, C] = to_out(o) + x
    with its GroupNorm partial sums attached (`.gn_partial` [B, heads, C, 2])."""
    _dev(x, "x")
    B, T, C = x.shape
    o = torch.empty_like(x)
    y = torch.empty_like(x)
    st = torch.empty((B, heads, C, 2), dtype=torch.float32, device=x.device)
    sync = _sync_words(x.device)
    assert stats.st2 is None and stats.st1.shape[2] == C
    tok = _begin()
    check(lib.afldm_attn_block_fused_out(ptr(x), ptr(stats.st1), stats.S1, ptr(gamma), ptr(beta), int(G), float(eps),
                                         ptr(w_qkv), ptr(bias_qkv), ptr(o), ptr(w_out), ptr(bias_out), ptr(y), ptr(st),
                                         ptr(sync), sync.numel() * 4, B, T, C, heads, float(scale), _code(x), stream_ptr()),
          "attn_block_fused_out")
    d = C // heads
    _end(tok, "attn_fused", 2.0 * B * T * 4 * C * C + 4.0 * B * heads * T * T * d,
         (3 * B * T * C + 4 * C * C) * x.element_size())
    y.gn_partial = st
    return y


def attn_block_fused(x, stats, gamma, beta, G, eps, w_qkv, bias_qkv, heads, scale, out=None):
    """GroupNorm-apply -> q | k | v projection -> attention of one attention block in ONE launch.  x [B, T, C] raw
    tokens; stats = GNStats of x; w_qkv [3C, 1, 1, C] / bias_qkv [3C] = packed (to_q | to_k | to_v).  Returns the input
    of to_out, [B, T, C]."""
    _dev(x, "x")
    B, T, C = x.shape
    if out is None:
        out = torch.empty_like(x)
    assert stats.st2 is None and stats.st1.shape[2] == C
    tok = _begin()
    check(lib.afldm_attn_block_fused(ptr(x), ptr(stats.st1), stats.S1, ptr(gamma), ptr(beta), int(G), float(eps),
                                     ptr(w_qkv), ptr(bias_qkv), ptr(out), B, T, C, heads, float(scale), _code(x),
                                     stream_ptr()), "attn_block_fused")
    d = C // heads
    _end(tok, "attn_fused", 2.0 * B * T * 3 * C * C + 4.0 * B * heads * T * T * d,
         (2 * B * T * C + 3 * C * C) * x.element_size())
    return out


_ATTN_SMALL = os.environ.get("AFLDM_NO_ATTN_SMALL", "0") != "1"
# token counts the policy sends to the fused small-plane launch.  EMPTY by default: correct (tests) but not faster in the step -
# 4.662 / 4.665 ms without, 4.676 / 4.670 with the 4x4 sites, 4.718 with the 8x8 sites too (profiles/r05/attn_small_ab.txt: on
# its own 20.1 us at 4x4 against 17 + 8, 27.0 against 22.2 at 8x8 where the per-lane VALU attention over 64 keys is 12 us of it).
# AFLDM_ATTN_SMALL_T=16,64 enables it.
_ATTN_SMALL_T = tuple(int(v) for v in os.environ.get("AFLDM_ATTN_SMALL_T", "").split(",") if v)


def attn_small_fused_ok(x, heads):
    """True when afldm_attn_small_fused has a kernel for the (already normalised) tokens x [B, T, C]."""
    if not _ATTN_SMALL or x.dtype != torch.bfloat16 or x.ndim != 3:
        return False
    B, T, C = x.shape
    if T not in _ATTN_SMALL_T:
        return False
    return bool(_exp.lib().afldm_attn_small_fused_supported(B, T, C, int(heads)))


def attn_small_fused(x, w_qkv, bias_qkv, heads, scale, out=None):
    """q | k | v projection + attention of the 8x8 / 4x4 levels in ONE launch; x [B, T, C] = the GroupNorm-ed tokens."""
    _dev(x, "x")
    B, T, C = x.shape
    if out is None:
        out = torch.empty_like(x)
    tok = _begin()
    check(_exp.lib().afldm_attn_small_fused(ptr(x), ptr(w_qkv), ptr(bias_qkv), ptr(out), B, T, C, int(heads), float(scale), _code(x),
                                     stream_ptr()), "attn_small_fused")
    d = C // heads
    _end(tok, "attn_small", 2.0 * B * T * 3 * C * C + 4.0 * B * heads * T * T * d, (2 * B * T * C + 3 * C * C) * x.element_size())
    return out


# ----------------------------------------------------------------------------- DDIM
def ddim_step(x, eps_nhwc, coef, step_idx, advance=False, out=None):
    """x NCHW fp32, eps NHWC dtype; coef float[4*nsteps] and step_idx int32[1] on device."""
    _dev(x, "x"); _dev(eps_nhwc, "eps")
    B, C, H, W = x.shape
    if out is None:
        out = torch.empty_like(x)
    check(lib.afldm_ddim_step(ptr(x), ptr(eps_nhwc), ptr(out), ptr(coef), ptr(step_idx), int(advance), B, C, H, W,
                              _code(eps_nhwc), stream_ptr()), "ddim_step")
    return out


def ddim_step_flat(x, eps, coefs, out=None):
    """x, eps: same-shape contiguous fp32 CUDA tensors; coefs = 4 python floats."""
    _dev(x, "x"); _dev(eps, "eps")
    assert x.dtype == eps.dtype == torch.float32 and x.shape == eps.shape
    if out is None:
        out = torch.empty_like(x)
    check(lib.afldm_ddim_step_flat(ptr(x), ptr(eps), ptr(out), *[float(c) for c in coefs], x.numel(), stream_ptr()),
          "ddim_step_flat")
    return out


def select_timestep(tvals, step_idx, t_out, pre_advance=False):
    check(lib.afldm_select_timestep(ptr(tvals), ptr(step_idx), ptr(t_out), int(pre_advance), stream_ptr()),
          "select_timestep")
    return t_out


def select_step_row(tvals, step_idx, t_out, table, row_out, pre_advance=False):
    """afldm_select_step_row: select_timestep + row_out <- table[step] in one launch."""
    assert table.is_contiguous() and row_out.is_contiguous() and table[0].numel() == row_out.numel()
    nbytes = row_out.numel() * row_out.element_size()
    check(lib.afldm_select_step_row(ptr(tvals), ptr(step_idx), ptr(t_out), int(pre_advance), ptr(table), ptr(row_out),
                                    nbytes, stream_ptr()), "select_step_row")
    return t_out
