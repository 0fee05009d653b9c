# round 5: where a TU wave's time goes — single-wave latencies, list lengths, phase cuts; base (round-4 kernel) vs new
for lib in base new; do
  L=$PWD/vvenc_amd/libvvenc_hip.so; [ $lib = base ] && L=$PWD/vvenc_amd/libvvenc_hip_base.so
  for mix in 64:1 64:256 64:955 64:2048 64:3072 32:1 32:1024 32:2133 32:4096 16:600 8:800 4:600 64:955,32:2133,16:600,8:800,4:600; do
    echo -n "$lib "; VVHIP_LIB=$L python tools/tu_mix.py $mix --reps 100 2>&1 | tail -1
  done
done
for ph in 1 2 3 4 5 6 7; do echo -n "new phases<=$ph "; VVHIP_TU_PHASES=$ph python tools/tu_mix.py 32:1 --reps 100 2>&1 | tail -1; done
for ph in 1 2 3 4 5 6 7; do echo -n "new phases<=$ph "; VVHIP_TU_PHASES=$ph python tools/tu_mix.py 32:2133 --reps 100 2>&1 | tail -1; done
for rw in 2048 3072 4096 6144; do echo -n "new resident $rw "; VVHIP_TU_RESIDENT_WAVES=$rw python tools/tu_mix.py 64:955,32:2133,16:600,8:800,4:600 --reps 100 2>&1 | tail -1; done
