#!/usr/bin/env python
"""In-situ variant tuning: micro-benchmarks of one convolution in a loop do not predict its time
inside the step (inputs L2/MALL-hot from the producer, neighbours competing for the same caches);
this runs the whole denoise step (bench.py, HIP graph) with one shape's tile variant / split-K
overridden (AFLDM_CONV_OVERRIDE) and reports the step time per candidate.

  python tools/tune_insitu.py            # all shapes below
  python tools/tune_insitu.py S1 S3      # a subset
"""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
# name: (M, Cout, KS, Ct), candidates "variant/splitk" (first = current policy)
SHAPES = {
    "S1 L32 3x3 192->192": ((65536, 192, 3, 192), ["29/1", "12/1", "26/1", "33/1"]),
    "S1b L32 3x3 576->192": ((65536, 192, 3, 576), ["29/1", "26/1", "33/1"]),
    "S1c L32 3x3 384->192": ((65536, 192, 3, 384), ["29/1", "26/1", "33/1"]),
    "S1d L32 3x3 384->384": ((65536, 384, 3, 384), ["29/1", "26/1", "33/1"]),
    "S2 L16 3x3 384->384": ((16384, 384, 3, 384), ["33/1", "29/1", "30/1", "22/1"]),
    "S2b L16 3x3 768->384": ((16384, 384, 3, 768), ["33/1", "29/1", "30/1"]),
    "S3 L8 3x3 384->384": ((4096, 384, 3, 384), ["32/1", "33/4", "33/2", "31/2", "11/1"]),
    "S4 L32 qkv 192->576": ((65536, 576, 1, 192), ["29/1", "31/1", "30/1", "12/1"]),
    "S4b L16 qkv 384->1152": ((16384, 1152, 1, 384), ["31/1", "29/1", "30/1", "33/1"]),
    "S5 L4 3x3 768->768": ((1024, 768, 3, 768), ["33/8", "33/4", "31/4", "31/8", "24/4"]),
    "S6 L2 3x3 768->768": ((256, 768, 3, 768), ["32/4", "32/8", "11/4", "31/8"]),
    "S7 L32 to_out 192->192": ((65536, 192, 1, 192), ["29/1", "31/1", "12/1"]),
    "S9 L2 dense 3072->3072": ((64, 3072, 1, 3072), ["32/4", "32/8", "32/2", "32/12", "31/8", "33/8"]),
    "S9b L2 dense 6144->3072": ((64, 3072, 1, 6144), ["32/4", "32/8", "32/12", "31/8", "33/8"]),
    "T1 L4 qkv 768->2304": ((1024, 2304, 1, 768), ["32/1", "31/1", "30/1", "33/1", "32/2", "29/1"]),
    "T2 L4 out 768->768": ((1024, 768, 1, 768), ["32/1", "31/1", "32/2", "11/1"]),
    "T3 L8 qkv 384->1152": ((4096, 1152, 1, 384), ["32/1", "31/1", "30/1", "29/1", "33/1"]),
    "T4 L8 out 384->384": ((4096, 384, 1, 384), ["32/1", "31/1", "30/1"]),
    "T5 L16 out 384->384": ((16384, 384, 1, 384), ["31/1", "30/1", "29/1", "32/1"]),
    "S8 L8 3x3 1152->384": ((4096, 384, 3, 1152), ["33/4", "33/2", "32/1", "29/4"]),
}


def run(override):
    env = dict(os.environ)
    if override:
        env["AFLDM_CONV_OVERRIDE"] = override
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--steps", "30", "--warmup", "5",
                          "--no-cpu-baseline", "--no-roofline"], env=env, capture_output=True, text=True)
    line = [l for l in out.stdout.splitlines() if l.startswith("{")]
    return json.loads(line[-1])["ms_per_step"] if line else float("nan")


def main():
    sel = sys.argv[1:]
    base = [run(None) for _ in range(2)]
    print(f"baseline ms/step: {base}", flush=True)
    for name, ((M, Co, KS, Ct), cands) in SHAPES.items():
        if sel and not any(name.startswith(s) for s in sel):
            continue
        res = []
        for c in cands:
            v, sk = c.split("/")
            res.append((c, run(f"{M}:{Co}:{KS}:{Ct}={v}/{sk}")))
        print(f"{name:26s} " + "  ".join(f"v{c}: {t:.3f}" for c, t in res), flush=True)


if __name__ == "__main__":
    main()
