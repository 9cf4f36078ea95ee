#!/usr/bin/env python
"""Ordered kernel timeline of ONE denoise step from a rocprofv3 kernel trace CSV (name, grid, duration, gap)."""
import csv
import re
import sys

rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
idx = [i for i, r in enumerate(rows) if "k_select_timestep" in r["Kernel_Name"] or "k_select_step_row" in r["Kernel_Name"]]
step = rows[idx[-2]:idx[-1]]
prev_end = None
for r in step:
    n = r["Kernel_Name"]
    m = re.search(r"afldm\d+(k_[a-z0-9_]+)", n)
    short = m.group(1) if m else n[:24]
    t = re.search(r"I(DF16b|f)((?:Li\d+E|Lb\dE)*)", n)
    targs = "" if not t else ",".join(re.findall(r"L[ib](\d+)E", t.group(2)))
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    gap = 0 if prev_end is None else (s - prev_end) / 1e3
    prev_end = e
    wgs = int(r["Grid_Size_X"]) // int(r["Workgroup_Size_X"])
    print(f"{short:22s} <{targs:28s}> wgs={wgs:5d} z={r['Grid_Size_Z']:>2s} {(e - s) / 1e3:8.1f} us  gap {gap:5.1f}")
