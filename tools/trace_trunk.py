#!/usr/bin/env python
"""Phase stamps of the cooperative 2x2-level launch (afldm_trunk_trace): per phase, workgroup 0's time in the phase body and in
the grid barrier behind it.  B=<batch>"""
import os
import sys

R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, R)
sys.path.insert(0, os.path.join(R, "tests"))
import torch  # noqa: E402

from afldm_amd import _exp, _lib, ops, trunk  # noqa: E402
from test_gpu_r02 import build_unet  # noqa: E402

B = int(os.environ.get("B", "64"))
unet, _, _ = build_unet("ffhq", torch.bfloat16)
x = torch.randn(B, 4, 32, 32, generator=torch.Generator().manual_seed(B)).cuda()
for _ in range(2):
    unet(x, 501)
tr = next(v for k, v in unet._afldm_cache.items() if isinstance(k, tuple) and k[0] == "trunk2")
buf = torch.zeros(tr.nphases, 4, dtype=torch.int64, device="cuda")
_exp.lib().afldm_trunk_trace(buf.data_ptr())
unet(x, 501)
torch.cuda.synchronize()
_exp.lib().afldm_trunk_trace(None)
t = buf.cpu().double()
names = {1: "GEMM", 2: "RED ", 3: "ATTN"}
prog = bytes(tr.program.cpu().numpy())
psz = _exp.lib().afldm_trunk_phase_bytes()
tot_body = tot_bar = 0.0
for i in range(tr.nphases):
    ph = trunk.Phase.from_buffer_copy(prog[i * psz:(i + 1) * psz])
    nxt = (t[i + 1, 0] - t[i, 3]) if i + 1 < tr.nphases else 0.0
    print(f"phase {i:2d} {names[ph.type]} jobs {ph.njobs}: body + store drain {t[i, 1] - t[i, 0]:8.0f} | release fence + wg barrier {t[i, 2] - t[i, 1]:8.0f} | "
          f"arrive, poll, acquire {t[i, 3] - t[i, 2]:8.0f} | to next {nxt:6.0f} ticks")
print(f"B={B}: {tr.nphases} phases, first start -> last barrier {t[-1, 3] - t[0, 0]:.0f} ticks (s_memtime: 100 MHz constant clock when the shader clock register is not selected; see bench box memtime_ticks_per_us)")
