for m in 64:955,32:2133,16:600 64:955,32:2133,8:800 64:955,32:2133,4:600 64:955,32:2133,16:600,8:800 32:2133,16:600,8:800 32:2133,16:600,8:800,4:600; do python tools/tu_mix.py $m 2>/dev/null; done
echo "--- resident waves 100000 (one tile per wave)"
for m in 64:955,32:2133,16:600,8:800,4:600 64:955,32:2133; do VVHIP_TU_RESIDENT_WAVES=100000 python tools/tu_mix.py $m 2>/dev/null; done
echo "--- one-launch off"
for m in 64:955,32:2133,16:600,8:800,4:600; do VVHIP_TU_ONE_LAUNCH_TILES=0 python tools/tu_mix.py $m 2>/dev/null; done
