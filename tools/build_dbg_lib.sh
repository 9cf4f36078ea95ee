#!/bin/bash
# Timing-decomposition builds: tools/build_dbg_lib.sh <source stem> <MACRO> [<MACRO> ...] ->
# afldm_amd/lib/libafldm_<stem>_<MACRO>.so = the standard library with <stem>.hip compiled with -D<MACRO>
# (e.g. conv3h AFLDM_H3_NOLOAD, skinny AFLDM_SK_NOX).  Garbage results; load with AFLDM_LIB=...
set -e
cd "$(dirname "$0")/.."
python -m afldm_amd.build > /dev/null
L=afldm_amd/lib
f=$1; shift
for m in "$@"; do
  wt=""; case $f in conv|conv3h) wt="-DAFLDM_WT=1";; esac
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wall -Wno-unused-function $wt -D$m -c afldm_amd/csrc/$f.hip -o /tmp/dbg_${f}_$m.o
  objs=""
  for o in $(ls $L/*.o | xargs -n1 basename | sed 's/\.o$//'); do
    if [ $o = $f ]; then objs="$objs /tmp/dbg_${f}_$m.o"; else objs="$objs $L/$o.o"; fi
  done
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o $L/libafldm_${f}_$m.so $objs
  echo built $L/libafldm_${f}_$m.so
done
