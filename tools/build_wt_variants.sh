#!/bin/bash
# A/B builds: libafldm_wt_<file>.so = the standard library with ONE source compiled with -DAFLDM_WT=1 (write-through
# output stores); load with AFLDM_LIB=afldm_amd/lib/libafldm_wt_<file>.so
set -e
cd "$(dirname "$0")/.."
python -m afldm_amd.build > /dev/null
L=afldm_amd/lib
for f in "$@"; do
  extra=""
  case $f in af|attn|attnf|attns|sep) extra="-mllvm -amdgpu-mfma-vgpr-form";; esac
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wall -Wno-unused-function $extra -DAFLDM_WT=1 -c afldm_amd/csrc/$f.hip -o /tmp/wt_$f.o
  objs=""
  for o in api misc gn af sep conv conv3h actconv trunk attn attnf attns fir lin skinny convout; do
    if [ $o = $f ]; then objs="$objs /tmp/wt_$f.o"; else objs="$objs $L/$o.o"; fi
  done
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o $L/libafldm_wt_$f.so $objs
  echo built $L/libafldm_wt_$f.so
done
