#!/bin/bash
# what do the near-empty launches of leg C (phase B: neighbour tests + fixed-point resolution, 2 launches per level) cost a GOP cycle?  The cycle with a development build of the
# library (-DVVHIP_DEV_KNOBS, tools/exp/_ab/libvvenc_hip_dev.so) with and without $VVHIP_MCTF_NO_PHASE_B (results wrong by construction, timing only)
cd "$(dirname "$0")/../.."
export VVHIP_LIB=$PWD/tools/exp/_ab/libvvenc_hip_dev.so
for rep in 1 2; do
  for knob in 0 2 1; do
    if [ $knob != 0 ]; then export VVHIP_MCTF_NO_PHASE_B=$knob; else unset VVHIP_MCTF_NO_PHASE_B; fi
    echo "== no_phase_b=$knob"
    python tools/mctf_overlap.py 2>&1 | grep -E "^(ab|mctf one lane|both one lane) "
  done
done
