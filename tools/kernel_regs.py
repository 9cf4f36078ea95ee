"""Per-kernel register / LDS / spill figures of a hipcc object or shared library (gfx950 code object metadata).
usage: python tools/kernel_regs.py <file.o|.so> [name-filter]"""
import re
import subprocess
import sys
import tempfile
import os

LLVM = "/opt/rocm/lib/llvm/bin"


def code_objects(path):
    tmp = tempfile.mkdtemp()
    out = os.path.join(tmp, "dev.co")
    kind = "o" if path.endswith(".o") else "o"
    r = subprocess.run([f"{LLVM}/clang-offload-bundler", "--unbundle", f"--type={kind}", f"--input={path}",
                        "--targets=hipv4-amdgcn-amd-amdhsa--gfx950", f"--output={out}"], capture_output=True, text=True)
    if r.returncode != 0 or not os.path.exists(out) or os.path.getsize(out) == 0:
        # a host object with an embedded fat binary: pull .hip_fatbin out first
        fb = os.path.join(tmp, "fatbin")
        subprocess.run([f"{LLVM}/llvm-objcopy", "-O", "binary", "--only-section=.hip_fatbin", path, fb], check=True)
        subprocess.run([f"{LLVM}/clang-offload-bundler", "--unbundle", "--type=o", f"--input={fb}",
                        "--targets=hipv4-amdgcn-amd-amdhsa--gfx950", f"--output={out}"], check=True)
    return out


def main():
    path = sys.argv[1]
    flt = sys.argv[2] if len(sys.argv) > 2 else ""
    co = code_objects(path)
    txt = subprocess.run([f"{LLVM}/llvm-readelf", "--notes", co], capture_output=True, text=True).stdout
    # the metadata is YAML-ish text: split per kernel on ".name:"
    blocks = re.split(r"\n\s*- \.agpr_count:", txt)
    rows = []
    for b in blocks[1:]:
        b = ".agpr_count:" + b
        def g(key):
            m = re.search(rf"\.{key}:\s*(\S+)", b)
            return m.group(1) if m else "?"
        name = g("name")
        if flt and flt not in name:
            continue
        dem = subprocess.run(["c++filt", name], capture_output=True, text=True).stdout.strip()
        rows.append((dem[:110], g("vgpr_count"), g("agpr_count"), g("sgpr_count"), g("vgpr_spill_count"), g("sgpr_spill_count"),
                     g("group_segment_fixed_size"), g("private_segment_fixed_size"), g("max_flat_workgroup_size")))
    print(f"{'kernel':110s} vgpr agpr sgpr vspill sspill lds scratch wg")
    for r in rows:
        print(f"{r[0]:110s} " + " ".join(str(x) for x in r[1:]))


if __name__ == "__main__":
    main()
