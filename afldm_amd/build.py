"""Build libafldm_hip.so (gfx950) in-tree with hipcc.  `python -m afldm_amd.build [--force]`.

hipcc cross-compiles without a GPU; the resulting .so travels to the GPU box with the repo
snapshot (it is git-ignored, not gpurun-ignored)."""
import concurrent.futures
import os
import shutil
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
OUT_DIR = os.path.join(HERE, "lib")
LIB = os.path.join(OUT_DIR, "libafldm_hip.so")
SOURCES = ["api.hip", "misc.hip", "gn.hip", "af.hip", "sep.hip", "conv.hip", "conv3h.hip", "attn.hip", "attnf.hip", "fir.hip", "lin.hip", "skinny.hip", "convout.hip", "dense2.hip"]
# Experiments (include/afldm_hip_experimental.h): built, bit-identical to the launches they replace, measured slower - a library of
# their own that links against the product library and that the default path never loads
EXP_LIB = os.path.join(OUT_DIR, "libafldm_exp.so")
EXP_SOURCES = ["actconv.hip", "trunk.hip", "attns.hip"]
ARCH = "gfx950"
FLAGS = [f"--offload-arch={ARCH}", "-O3", "-std=c++17", "-fPIC", "-Wall", "-Wno-unused-function"]
# Kernels whose MFMA accumulators are post-processed by VALU code (softmax, SiLU, GroupNorm affine):
# keep the accumulators in architectural VGPRs.  By default the compiler parks them in AGPRs and
# pays a v_accvgpr_read/write per element around every VALU use (144 of them per 64-key chunk of
# the attention loop: as many cycles as the exponentials).  AFLDM_VGPR_FORM=all|none overrides.
VGPR_FORM = {"attn.hip", "attnf.hip", "attns.hip", "af.hip", "sep.hip"}       # (actconv.hip: A/B below)
VGPR_FORM_FLAGS = ["-mllvm", "-amdgpu-mfma-vgpr-form"]
# Sources whose OUTPUT tensors are stored write-through (sc1: st16_out in common.hpp).  Measured in the step, same box
# (profiles/r02/write_through_ab.txt): the convolution epilogues gain (nothing is left dirty in the XCD L2s for the
# end-of-kernel release to write back in front of the next launch), every other kernel family loses.
WRITE_THROUGH = {"conv.hip", "conv3h.hip"}


def _hipcc():
    for c in (os.environ.get("HIPCC"), shutil.which("hipcc"), "/opt/rocm/bin/hipcc"):
        if c and os.path.exists(c):
            return c
    raise RuntimeError("hipcc not found (need ROCm >= 7.0 for gfx950)")


def _stale(target, deps):
    if not os.path.exists(target):
        return True
    t = os.path.getmtime(target)
    return any(os.path.getmtime(d) > t for d in deps)


def _fingerprint(cmd, deps):
    """sha256 over the compile command and the CONTENTS of the source and its headers: an object is reused only when
    this matches the fingerprint stored next to it (a changed flag or header rebuilds it; timestamps play no part)."""
    import hashlib
    h = hashlib.sha256(" ".join(cmd).encode())
    for d in deps:
        h.update(open(d, "rb").read())
    return h.hexdigest()


def build(force=False, verbose=True):
    os.makedirs(OUT_DIR, exist_ok=True)
    hipcc = _hipcc()
    headers = [os.path.join(CSRC, h) for h in ("common.hpp", "conv_common.hpp", "conv3h_tile.hpp", "conv3h_body.inc", "af_plane.hpp",
                                               "af_plane_passes.inc")] + [os.path.join(HERE, "..", "include", "afldm_hip.h")]
    exp_headers = headers + [os.path.join(HERE, "..", "include", "afldm_hip_experimental.h")]
    def command(src, obj):
        mode = os.environ.get("AFLDM_VGPR_FORM", "")
        vg = mode == "all" or (mode != "none" and os.path.basename(src) in VGPR_FORM)
        wt = ["-DAFLDM_WT=1"] if os.path.basename(src) in WRITE_THROUGH and os.environ.get("AFLDM_NO_WT") is None else []
        return [hipcc] + FLAGS + (VGPR_FORM_FLAGS if vg else []) + wt + ["-c", src, "-o", obj]

    objs, exp_objs, jobs = [], [], []
    for s in SOURCES + EXP_SOURCES:
        src = os.path.join(CSRC, s)
        obj = os.path.join(OUT_DIR, s.replace(".hip", ".o"))
        (exp_objs if s in EXP_SOURCES else objs).append(obj)
        fp = _fingerprint([os.path.basename(c) if os.sep in c else c for c in command(src, obj)],
                          [src] + (exp_headers if s in EXP_SOURCES else headers))
        stamp = obj + ".sha256"
        same = os.path.exists(obj) and os.path.exists(stamp) and open(stamp).read().strip() == fp
        if force or not same:
            jobs.append((src, obj, fp))

    def cc(job):
        src, obj, fp = job
        cmd = command(src, obj)
        r = subprocess.run(cmd, capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError(f"hipcc failed for {src}:\n{r.stdout}\n{r.stderr}")
        with open(obj + ".sha256", "w") as f:
            f.write(fp + "\n")
        return src

    if jobs:
        with concurrent.futures.ThreadPoolExecutor(max_workers=min(len(jobs), os.cpu_count() or 4)) as ex:
            for done in ex.map(cc, jobs):
                if verbose:
                    print("[afldm_amd.build] compiled", os.path.basename(done), flush=True)
    if force or _stale(LIB, objs):
        cmd = [hipcc, f"--offload-arch={ARCH}", "-shared", "-fPIC", "-o", LIB] + objs
        r = subprocess.run(cmd, capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError(f"link failed:\n{r.stdout}\n{r.stderr}")
        # hipcc can silently drop a kernel template (seen with ROCm 7.2): refuse a library whose own
        # kernels are unresolved instead of failing at dlopen time on the GPU box
        nm = subprocess.run(["nm", "-D", "--undefined-only", LIB], capture_output=True, text=True)
        missing = [l.split()[-1] for l in nm.stdout.splitlines() if "afldm" in l]
        if missing:
            os.remove(LIB)
            raise RuntimeError("libafldm_hip.so has unresolved afldm symbols: " + ", ".join(missing[:4]))
        if verbose:
            print("[afldm_amd.build] linked", LIB, flush=True)
    if force or _stale(EXP_LIB, exp_objs + [LIB]):
        # the experiments call into the product library (afldm_conv2d, afldm_af_act, conv3h_plan, the error string): link against it
        cmd = [hipcc, f"--offload-arch={ARCH}", "-shared", "-fPIC", "-o", EXP_LIB] + exp_objs + [f"-L{OUT_DIR}", "-lafldm_hip", "-Wl,-rpath,$ORIGIN"]
        r = subprocess.run(cmd, capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError(f"link of libafldm_exp.so failed:\n{r.stdout}\n{r.stderr}")
        if verbose:
            print("[afldm_amd.build] linked", EXP_LIB, flush=True)
    try:
        build_aql(force, verbose)
    except Exception as e:          # a diagnostic library (AQL packet view): the product never loads it
        if verbose:
            print(f"[afldm_amd.build] libafldm_aql.so not built ({e}); only tools/aql_*.py need it", flush=True)
    return LIB


AQL_LIB = os.path.join(OUT_DIR, "libafldm_aql.so")


def build_aql(force=False, verbose=True):
    """libafldm_aql.so (csrc/aqlq.cpp): host-only C++ on the ROCr tools interface - the AQL packet view of the step."""
    src = os.path.join(CSRC, "aqlq.cpp")
    if not (force or _stale(AQL_LIB, [src])):
        return AQL_LIB
    rocm = os.environ.get("ROCM_PATH", "/opt/rocm")
    cmd = ["g++", "-O2", "-std=c++17", "-fPIC", "-shared", "-fvisibility=hidden", "-DAMD_INTERNAL_BUILD",
           f"-I{rocm}/include", f"-I{rocm}/include/hsa", src, "-o", AQL_LIB]
    r = subprocess.run(cmd, capture_output=True, text=True)
    if r.returncode != 0:
        raise RuntimeError(f"g++ failed for {src}:\n{r.stdout}\n{r.stderr}")
    if verbose:
        print("[afldm_amd.build] linked", AQL_LIB, flush=True)
    return AQL_LIB


if __name__ == "__main__":
    build(force="--force" in sys.argv)
