#!/bin/bash
# Cut builds of the refinement-stage kernel (the method of DESIGN 6: what the per-unit skeleton costs without its two phases).  Builds three variants of the library into
# tools/exp/_r3me/ (run here, they travel to the GPU box): cut 1 = no first pass, cut 2 = no second pass + distortion, cut 3 = neither (records, tables, barriers, position
# lists, cost hand-over only).  On the GPU box:  for c in 0 1 2 3; do VVHIP_LIB=$PWD/tools/exp/_r3me/libvvenc_hip_cut$c.so python tools/me_parts.py 2,5,15; done
set -e
cd "$(dirname "$0")/.."
mkdir -p tools/exp/_r3me
for c in 1 2 3; do
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -fvisibility=hidden -Wno-unused-function -DVVHIP_ME_CUT=$c -c vvenc_amd/csrc/me.hip -o tools/exp/_r3me/me_cut$c.o
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o tools/exp/_r3me/libvvenc_hip_cut$c.so $(ls vvenc_amd/csrc/*.o | grep -v "/me.o") tools/exp/_r3me/me_cut$c.o
done
cp vvenc_amd/libvvenc_hip.so tools/exp/_r3me/libvvenc_hip_cut0.so
