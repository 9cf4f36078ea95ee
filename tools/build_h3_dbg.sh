#!/bin/bash
# Timing-decomposition builds of conv3h.hip: libafldm_h3_<NAME>.so = the standard library with conv3h.hip compiled with
# -DAFLDM_H3_<NAME> (NOLOAD: MFMAs without fragment reads, NOMMA: fragment reads without MFMAs; garbage results).
# Load with AFLDM_LIB=afldm_amd/lib/libafldm_h3_<NAME>.so
set -e
cd "$(dirname "$0")/.."
python -m afldm_amd.build > /dev/null
L=afldm_amd/lib
for f in "$@"; do
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wall -Wno-unused-function -DAFLDM_WT=1 -DAFLDM_H3_$f -c afldm_amd/csrc/conv3h.hip -o /tmp/h3_$f.o
  objs=""
  for o in api misc gn af sep conv conv3h attn fir lin; do
    if [ $o = conv3h ]; then objs="$objs /tmp/h3_$f.o"; else objs="$objs $L/$o.o"; fi
  done
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o $L/libafldm_h3_$f.so $objs
  echo built $L/libafldm_h3_$f.so
done
