"""The 2x2 level of the alias-free UNet2DModel as ONE cooperative launch (csrc/trunk.hip, afldm_trunk_run).

Reference: diffusers UNet2DModel.forward over down_blocks[-1] (DownBlock2D, no downsampler) -> mid_block (resnet, attention,
resnet) -> the resnets of up_blocks[0] (UpBlock2D, skip concatenations) of reference configs/ldm/model_unet.json after the
alias-free surgery (af_api.py:70-83).  This module is the HOST side: it packs the level's weights into the kernel's 192 x 192
blocks once per model, lays out a workspace per batch size and writes the PROGRAM of phases (GEMM partial products /
reduce + norm + activation / attention) the kernel's persistent workgroups walk between grid barriers.  The level's input and
output tensors and the step's time-embedding row are arguments of each run (the program refers to them by tag), so one program
serves eager calls and captured graphs alike.
"""
import ctypes
import os

import torch

from . import _exp, ops
from ._lib import check, stream_ptr

# OFF by default: built, correct, deterministic - and 2x SLOWER than the 51 launches it replaces (profiles/r05/trunk_coop.txt:
# a phase costs >= 7 us on this chip - two memory-side round trips + a grid barrier - against ~7 us per stand-alone launch,
# and the split-K slabs it exchanges are 4x the launches'); AFLDM_TRUNK=1 enables it (tests do).
_ENABLED = os.environ.get("AFLDM_TRUNK", "0") == "1"
TB = 192
PH_GEMM, PH_RED, PH_ATTN = 1, 2, 3
EXT_IN, EXT_OUT = 0, 1


class GemmJob(ctypes.Structure):
    _fields_ = [("A", ctypes.c_uint64), ("W", ctypes.c_uint64), ("slab", ctypes.c_uint64), ("a_ld", ctypes.c_int),
                ("rows", ctypes.c_int), ("nsplit", ctypes.c_int), ("ksplit", ctypes.c_int), ("slab_ld", ctypes.c_int),
                ("slab0", ctypes.c_int), ("slab_stride", ctypes.c_longlong)]


class RedJob(ctypes.Structure):
    _fields_ = [("slab", ctypes.c_uint64), ("src", ctypes.c_uint64), ("bias", ctypes.c_uint64), ("bias2", ctypes.c_uint64),
                ("residual", ctypes.c_uint64), ("out_raw", ctypes.c_uint64), ("gamma", ctypes.c_uint64), ("beta", ctypes.c_uint64),
                ("out_act", ctypes.c_uint64), ("slab_stride", ctypes.c_longlong), ("nslab", ctypes.c_int), ("temb_off", ctypes.c_int),
                ("cpg", ctypes.c_int), ("mode", ctypes.c_int), ("B", ctypes.c_int), ("C", ctypes.c_int), ("eps", ctypes.c_float),
                ("pad", ctypes.c_int)]


class AttnJob(ctypes.Structure):
    _fields_ = [("slab", ctypes.c_uint64), ("bias", ctypes.c_uint64), ("out", ctypes.c_uint64), ("slab_stride", ctypes.c_longlong),
                ("nslab", ctypes.c_int), ("B", ctypes.c_int), ("C", ctypes.c_int), ("heads", ctypes.c_int), ("scale", ctypes.c_float),
                ("pad", ctypes.c_int)]


class _Jobs(ctypes.Union):
    _fields_ = [("g", GemmJob * 3), ("r", RedJob * 2), ("a", AttnJob)]


class Phase(ctypes.Structure):
    _fields_ = [("type", ctypes.c_int), ("njobs", ctypes.c_int), ("u", _Jobs)]


def _ext(idx, offset=0):
    return (0xE << 60) | (idx << 56) | offset


def _p(t):
    return 0 if t is None else t.data_ptr()


def _blocks(w2d):
    """[N, K] (row-major, K contiguous) -> contiguous 192 x 192 blocks [N/192][K/192][192][192]."""
    N, K = w2d.shape
    return w2d.reshape(N // TB, TB, K // TB, TB).permute(0, 2, 1, 3).contiguous()


def eligible(unet, x):
    """True when the level `x` [B, 2, 2, C] enters is the topology / dtype / size the cooperative kernel covers."""
    from .af_modules.af_blocks import WarpedNonlinearity
    from .models import blocks as Bk
    if not _ENABLED or x.dtype != torch.bfloat16 or x.ndim != 4 or x.shape[1] != 2 or x.shape[2] != 2 or x.shape[0] > 64:
        return False
    if lib_missing() or _BLOCKED:
        return False
    down, mid, up = unet.down_blocks[-1], unet.mid_block, unet.up_blocks[0]
    C = x.shape[-1]
    if (getattr(down, "has_attention", False) or down.downsamplers is not None or getattr(up, "has_attention", False)
            or len(mid.attentions) != 1 or mid.attentions[0] is None or len(mid.resnets) != 2 or C % TB):
        return False
    attn = mid.attentions[0]
    if type(attn.processor) is not Bk.AttnProcessor2_0 or attn.group_norm is None or C // attn.heads > 32 or attn.inner_dim != C:
        return False
    for r in list(down.resnets) + list(mid.resnets) + list(up.resnets):
        if (not isinstance(r.nonlinearity, WarpedNonlinearity) or not r.nonlinearity.fused_silu or r.out_channels != C
                or r.time_emb_proj is None or TB % (r.norm2.num_channels // r.norm2.num_groups)
                or r.in_channels % TB or TB % (r.in_channels // r.norm1.num_groups) or (r.in_channels // C) not in (1, 2)
                or tuple(r.conv1.kernel_size) != (3, 3)):
            return False
        if r.in_channels != C and (r.conv_shortcut is None or C % (r.in_channels // r.norm1.num_groups)):
            return False
    return True


def lib_missing():
    return not _exp.available()


# The cooperative launch needs the GPU to itself (spin grid barriers between resident workgroups): an engine that runs its batch
# as parallel graph branches blocks it for the process - two persistent kernels would starve each other's barriers.
_BLOCKED = set()


def block(reason):
    _BLOCKED.add(reason)


class LowResTrunk:
    """Program + packed weights + workspace of the 2x2 level for one (model, batch size)."""

    def __init__(self, unet, B, C, temb_offsets):
        from .models import blocks as Bk
        dev, dt = unet.device, torch.bfloat16
        assert ctypes.sizeof(Phase) == _exp.lib().afldm_trunk_phase_bytes(), "Phase layout differs from csrc/trunk.hip"
        self.B, self.C = B, C
        self.keep = []
        P4 = 4 * C
        down, mid, up = unet.down_blocks[-1], unet.mid_block, unet.up_blocks[0]
        order = list(unet._resnets_in_order())
        self.res = list(down.resnets) + list(mid.resnets) + list(up.resnets)
        idx = [next(i for i, r in enumerate(order) if r is m) for m in self.res]
        self.first_slice = idx[0]
        toff = {id(m): temb_offsets[i] - temb_offsets[idx[0]] for m, i in zip(self.res, idx)}

        def new(*shape, dtype=dt):
            t = torch.empty(shape, dtype=dtype, device=dev)
            self.keep.append(t)
            return t

        def f32(t):
            t = t.detach().to(device=dev, dtype=torch.float32).contiguous()
            self.keep.append(t)
            return t

        def dense_blocks(conv, c_parts):
            """blocks of the dense-over-the-plane form of a 3x3 conv, one block tensor per input tensor of the virtual concat"""
            w2, _ = Bk.packed_conv_dense2x2(conv, dt, c_parts[0], c_parts[1] if len(c_parts) > 1 else 0)
            w2 = w2.reshape(w2.shape[0], -1)
            outs, o = [], 0
            for cp in c_parts:
                blk = _blocks(w2[:, o:o + 4 * cp])
                self.keep.append(blk)
                outs.append(blk)
                o += 4 * cp
            return outs

        def lin_blocks(w2d):
            blk = _blocks(w2d.detach().to(device=dev, dtype=dt))
            self.keep.append(blk)
            return blk

        # workspace
        A_h, A_s, A2 = new(B, P4), new(B, P4), new(B, P4)
        HN, O = new(B, P4), new(B, P4)
        nslab_max = 32
        slabs = new(max(nslab_max * B * P4, 4 * 4 * B * 3 * C), dtype=torch.float32)
        SS = B * P4                                            # elements between slabs of a [B][4C] / [4B][C] layer
        U, D = ops.filter_matrices(2, dev)
        self.U, self.D = U, D
        phases = []

        def gemm(jobs):
            ph = Phase()
            ph.type, ph.njobs = PH_GEMM, len(jobs)
            for k, (A, a_ld, rows, W, slab_ld, slab0, stride) in enumerate(jobs):
                g = ph.u.g[k]
                g.A, g.W, g.slab = A, _p(W), _p(slabs)
                g.a_ld, g.rows, g.nsplit, g.ksplit = a_ld, rows, W.shape[0], W.shape[1]
                g.slab_ld, g.slab0, g.slab_stride = slab_ld, slab0, stride
            phases.append(ph)

        def red(jobs):
            ph = Phase()
            ph.type, ph.njobs = PH_RED, len(jobs)
            for k, j in enumerate(jobs):
                r = ph.u.r[k]
                r.slab, r.src = _p(slabs) if j.get("nslab", 0) else 0, j.get("src", 0)
                r.bias, r.bias2 = _p(j.get("bias")), _p(j.get("bias2"))
                r.residual, r.out_raw = j.get("residual", 0), j.get("out_raw", 0)
                norm = j.get("norm")
                if norm is not None:
                    mod, c0, cpg = norm
                    g_, b_ = Bk.packed_norm(mod)
                    r.gamma, r.beta = g_.data_ptr() + 4 * c0, b_.data_ptr() + 4 * c0
                    r.cpg, r.eps = cpg, float(mod.eps)
                    self.keep += [g_, b_]
                r.mode = j.get("mode", 0)
                r.out_act = j.get("out_act", 0)
                r.slab_stride, r.nslab, r.temb_off = SS, j.get("nslab", 0), j.get("temb_off", -1)
                r.B, r.C = B, C
            phases.append(ph)

        def norm1_jobs(nxt, h_raw, skip_raw):
            """activation jobs for the next resnet's norm1 over (h | skip): (job fields for the h part, [skip job])"""
            if nxt is None:
                return dict(mode=0), []
            cin = nxt.in_channels
            cpg = cin // nxt.norm1.num_groups
            hpart = dict(norm=(nxt.norm1, 0, cpg), mode=2, out_act=_p(A_h))
            extra = []
            if cin != C:
                extra = [dict(src=skip_raw, norm=(nxt.norm1, C, cpg), mode=2, out_act=_p(A_s))]
            return hpart, extra

        # skips consumed by up_blocks[0]: (level input, down.resnets[0] out, down.resnets[1] out), popped from the end
        x_in = _ext(EXT_IN)
        raw = x_in                                             # raw input of the current resnet
        skips = [x_in]
        seq = [("down", m) for m in down.resnets] + [("mid0", mid.resnets[0]), ("attn", mid.attentions[0]), ("mid1", mid.resnets[1])] + \
              [("up", m) for m in up.resnets]
        # first activation: norm1 of the first resnet on the level input
        first = down.resnets[0]
        red([dict(src=x_in, norm=(first.norm1, 0, first.in_channels // first.norm1.num_groups), mode=2, out_act=_p(A_h))])
        n_up_left = len(up.resnets)
        for k, (kind, m) in enumerate(seq):
            nxt_kind, nxt = seq[k + 1] if k + 1 < len(seq) else (None, None)
            last = nxt is None
            if kind == "attn":
                attn = m
                wq = torch.cat([attn.to_q.weight, attn.to_k.weight, attn.to_v.weight], 0)
                bq = f32(torch.cat([attn.to_q.bias, attn.to_k.bias, attn.to_v.bias], 0))
                Wqkv, Wo = lin_blocks(wq), lin_blocks(attn.to_out[0].weight)
                gemm([(_p(HN), C, 4 * B, Wqkv, 3 * C, 0, 4 * B * 3 * C)])
                ph = Phase()
                ph.type, ph.njobs = PH_ATTN, 1
                a = ph.u.a
                a.slab, a.bias, a.out, a.slab_stride = _p(slabs), _p(bq), _p(O), 4 * B * 3 * C
                a.nslab, a.B, a.C, a.heads, a.scale = Wqkv.shape[1], B, C, attn.heads, float(attn.scale)
                phases.append(ph)
                gemm([(_p(O), C, 4 * B, Wo, C, 0, SS)])
                y = new(B, P4)
                hpart, extra = norm1_jobs(nxt, None, None)
                red([dict(nslab=Wo.shape[1], bias=f32(attn.to_out[0].bias), residual=raw, out_raw=_p(y), **hpart)] + extra)
                raw = _p(y)
                continue
            cin = m.in_channels
            concat = cin != C
            skip = skips.pop() if kind == "up" else None
            if concat:
                w1a, w1b = dense_blocks(m.conv1, (C, cin - C))
                gemm([(_p(A_h), P4, B, w1a, P4, 0, SS), (_p(A_s), P4, B, w1b, P4, w1a.shape[1], SS)])
                ns1 = w1a.shape[1] + w1b.shape[1]
            else:
                (w1,) = dense_blocks(m.conv1, (C,))
                gemm([(_p(A_h), P4, B, w1, P4, 0, SS)])
                ns1 = w1.shape[1]
            red([dict(nslab=ns1, bias=f32(m.conv1.bias), temb_off=toff[id(m)],
                      norm=(m.norm2, 0, C // m.norm2.num_groups), mode=2, out_act=_p(A2))])
            (w2,) = dense_blocks(m.conv2, (C,))
            jobs = [(_p(A2), P4, B, w2, P4, 0, SS)]
            ns2, bias2, residual = w2.shape[1], None, raw
            if m.conv_shortcut is not None:
                wsc = m.conv_shortcut.weight.detach().reshape(C, cin)
                wa, wb = lin_blocks(wsc[:, :C]), lin_blocks(wsc[:, C:])
                jobs += [(raw, C, 4 * B, wa, C, ns2, SS), (skip, C, 4 * B, wb, C, ns2 + wa.shape[1], SS)]
                ns2 += wa.shape[1] + wb.shape[1]
                bias2, residual = f32(m.conv_shortcut.bias), 0
            gemm(jobs)
            y_out = _ext(EXT_OUT) if last else _p(new(B, P4))
            if nxt_kind == "attn":
                hpart, extra = dict(norm=(nxt.group_norm, 0, C // nxt.group_norm.num_groups), mode=1, out_act=_p(HN)), []
            else:
                nskip = skips[-1] if (nxt_kind == "up" and skips) else None
                hpart, extra = norm1_jobs(nxt, y_out, nskip)
            red([dict(nslab=ns2, bias=f32(m.conv2.bias), bias2=bias2, residual=residual, out_raw=y_out, **hpart)] + extra)
            raw = y_out
            if kind == "down":
                skips.append(y_out)
        assert not skips, len(skips)
        buf = b"".join(bytes(ph) for ph in phases)
        self.nphases = len(phases)
        self.program = torch.frombuffer(bytearray(buf), dtype=torch.uint8).to(dev)
        self.sync = ops.new_sync_buffer(dev)

    def run(self, x, temb_ptr, temb_stride):
        """x: the level's input [B, 2, 2, C] (bf16, contiguous) -> its output [B, 2, 2, C]."""
        assert x.shape[0] == self.B and x.shape[-1] == self.C and x.is_contiguous()
        out = torch.empty_like(x)
        sync = ops._SYNC_OVERRIDE if ops._SYNC_OVERRIDE is not None else self.sync
        tok = ops._begin()
        check(_exp.lib().afldm_trunk_run(self.program.data_ptr(), self.nphases, ops.ptr(x), ops.ptr(out), temb_ptr, int(temb_stride),
                                      ops.ptr(self.U), ops.ptr(self.D), ops.ptr(sync), sync.numel() * 4, stream_ptr()), "trunk_run")
        ops._end(tok, "trunk_2x2", 0.0, 0.0)
        return out


def get(unet, x, temb_offsets):
    cache = unet.__dict__.setdefault("_afldm_cache", {})
    key = ("trunk2", x.shape[0], x.shape[-1])
    if key not in cache:
        cache[key] = LowResTrunk(unet, x.shape[0], x.shape[-1], temb_offsets)
    return cache[key]
