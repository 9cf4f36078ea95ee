#!/bin/bash
# Timing decomposition of the LDS-DMA implicit-GEMM kernel: AFLDM_CONV_DBG bit 0 skips the DMA
# issue, bit 1 skips the fragment-read + MFMA phase (results are garbage; timing only).
for dbg in 0 1 2 3; do
  echo "== AFLDM_CONV_DBG=$dbg"
  AFLDM_CONV_DBG=$dbg CONV_ALL=1 CONV_VARIANTS=${CONV_VARIANTS:-4,12,21,22,23} CONV_SHAPES=${CONV_SHAPES:-L32 384+192,L16 384+384} \
    python tools/bench_kernels.py conv 2>&1 | grep -v BAD | grep -v "^    "
done
