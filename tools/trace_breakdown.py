#!/usr/bin/env python
"""Per-kernel / per-shape time breakdown of ONE denoise step from a rocprofv3 kernel trace CSV."""
import collections
import csv
import re
import sys

rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
idx = [i for i, r in enumerate(rows) if "k_select_timestep" in r["Kernel_Name"] or "k_select_step_row" in r["Kernel_Name"]]
step = rows[idx[-2]:idx[-1]]


def short(n):
    if "k_conv3h" in n:
        m = re.search(r"Li(\d+)ELi(\d+)ELi(\d+)ELi(\d+)ELi(\d+)E", n) or re.search(r"<\w+, (\d+), (\d+), (\d+), (\d+), (\d+)", n)
        return f"conv3h_{m.group(1)}x{m.group(3)}w{m.group(2)}c{int(m.group(4)) * int(m.group(5))}"
    if "k_igemm3" in n:
        return "igemm3_128x192p"
    if "k_igemm" in n:
        m = re.search(r"Li(\d+)ELi(\d+)ELi\d+ELi\d+ELi\d+E(?:Li(\d+)E)?", n) or re.search(r"<\w+, (\d+), (\d+), \d+, \d+, \d+(?:, (\d+))?", n)
        return f"igemm{'2' if 'igemm2' in n else ''}_{m.group(1)}x{m.group(2)}" + (f"s{m.group(3)}" if m.group(3) else "")
    for k in ["k_attn_fused", "k_attn", "k_dense2_gn_act", "k_skinny", "k_af_act_plane", "k_af_act_p8", "k_af_act_slabs", "k_resample_small",
              "k_resample_plane", "k_conv_out_fused", "gn_partial", "gn_apply", "af_act_mfma", "af_act_kron", "af_act_small", "splitk", "axis_contract",
              "cin4", "small_cout", "silu", "ddim", "nchw", "timestep", "select", "advance"]:
        if k in n:
            return k
    return n[:30]


agg = collections.OrderedDict()
fam = collections.Counter()
for r in step:
    d = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
    nm = short(r["Kernel_Name"])
    k = (nm, int(r["Grid_Size_X"]) // int(r["Workgroup_Size_X"]), int(r["Grid_Size_Z"]))
    a = agg.setdefault(k, [0, 0.0])
    a[0] += 1
    a[1] += d
    fam[nm.split("_")[0] if nm.startswith("igemm") else nm] += d
tot = sum(v[1] for v in agg.values())
span = (int(step[-1]["End_Timestamp"]) - int(step[0]["Start_Timestamp"])) / 1e3
print(f"step: {len(step)} kernels, sum {tot:.1f} us, span {span:.1f} us")
for k, v in fam.most_common():
    print(f"  {k:16s} {v:8.1f} us  {100*v/tot:5.1f}%")
print()
for k, v in sorted(agg.items(), key=lambda kv: -kv[1][1])[:int(sys.argv[2]) if len(sys.argv) > 2 else 40]:
    print(f"{k[0]:18s} wgs={k[1]:6d} z={k[2]:2d} n={v[0]:3d} total={v[1]:8.1f}us avg={v[1]/v[0]:7.1f}")
