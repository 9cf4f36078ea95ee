#!/usr/bin/env python
"""Build afldm_amd/lib/libafldm_<tag>.so from the kernel sources of a git revision (default HEAD), with build.py's own
per-file flags, for same-box A/B timing of uncommitted kernel changes:
    python tools/build_rev_lib.py [rev] [tag]   ->   AFLDM_LIB=afldm_amd/lib/libafldm_<tag>.so python bench.py ...
Only files that differ from the working tree are recompiled; the rest reuses the objects of the current build."""
import os, subprocess, sys, tempfile
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from afldm_amd import build as B

rev = sys.argv[1] if len(sys.argv) > 1 else "HEAD"
tag = sys.argv[2] if len(sys.argv) > 2 else "REV"
B.build(verbose=False)
top = tempfile.mkdtemp(prefix="afldm_rev_")
tmp = os.path.join(top, "afldm_amd", "csrc")             # common.hpp includes "../../include/afldm_hip.h"
os.makedirs(tmp)
os.makedirs(os.path.join(top, "include"))
open(os.path.join(top, "include", "afldm_hip.h"), "wb").write(subprocess.check_output(["git", "show", f"{rev}:include/afldm_hip.h"], cwd=ROOT))
hipcc = B._hipcc()
for h in ("common.hpp", "conv_common.hpp"):
    open(os.path.join(tmp, h), "wb").write(subprocess.check_output(["git", "show", f"{rev}:afldm_amd/csrc/{h}"], cwd=ROOT))
objs = []
for s in B.SOURCES:
    cur = open(os.path.join(B.CSRC, s), "rb").read()
    try:
        old = subprocess.check_output(["git", "show", f"{rev}:afldm_amd/csrc/{s}"], cwd=ROOT)
    except subprocess.CalledProcessError:
        print("not in", rev, ":", s); continue
    hdr_same = all(open(os.path.join(B.CSRC, h), "rb").read() == open(os.path.join(tmp, h), "rb").read()
                   for h in ("common.hpp", "conv_common.hpp"))
    if old == cur and hdr_same:
        objs.append(os.path.join(B.OUT_DIR, s.replace(".hip", ".o")))
        continue
    src, obj = os.path.join(tmp, s), os.path.join(tmp, s.replace(".hip", ".o"))
    open(src, "wb").write(old)
    vg = s in B.VGPR_FORM
    wt = ["-DAFLDM_WT=1"] if s in B.WRITE_THROUGH else []
    cmd = [hipcc] + B.FLAGS + (B.VGPR_FORM_FLAGS if vg else []) + wt + ["-c", src, "-o", obj]
    subprocess.check_call(cmd)
    print("compiled", s, "from", rev)
    objs.append(obj)
out = os.path.join(B.OUT_DIR, f"libafldm_{tag}.so")
subprocess.check_call([hipcc, f"--offload-arch={B.ARCH}", "-shared", "-fPIC", "-o", out] + objs)
print("built", out)
