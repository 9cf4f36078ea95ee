#!/bin/bash
# Same-box A/B of ONE source file against an older revision of it:
#   tools/build_alt_lib.sh <source stem> <git rev> [tag]  ->  afldm_amd/lib/libafldm_<tag>.so = today's library with <stem>.hip of <rev>
# (e.g. af HEAD~1 afold).  Load with AFLDM_LIB=afldm_amd/lib/libafldm_<tag>.so; tools/ab_lib.sh runs the pair in the step.
set -e
cd "$(dirname "$0")/.."
python -m afldm_amd.build > /dev/null
L=afldm_amd/lib
f=$1; rev=$2; tag=${3:-${f}_alt}
mkdir -p /tmp/alt_$tag
git show $rev:afldm_amd/csrc/$f.hip > /tmp/alt_$tag/$f.hip
cp afldm_amd/csrc/*.hpp /tmp/alt_$tag/
mkdir -p /tmp/alt_$tag/../include 2>/dev/null || true
extra=""; case $f in attn|af|sep) extra="-mllvm -amdgpu-mfma-vgpr-form";; esac
wt=""; case $f in conv|conv3h) wt="-DAFLDM_WT=1";; esac
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wall -Wno-unused-function $extra $wt -I afldm_amd/csrc -I include -c /tmp/alt_$tag/$f.hip -o /tmp/alt_$tag/$f.o
objs=""
for o in api misc gn af sep conv conv3h attn fir lin skinny convout; do
  if [ $o = $f ]; then objs="$objs /tmp/alt_$tag/$f.o"; else objs="$objs $L/$o.o"; fi
done
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o $L/libafldm_$tag.so $objs
echo built $L/libafldm_$tag.so
