python -m pytest tests/test_gpu_corners.py -q -m gpu -k "all_zero" 2>&1 | tail -3
for amp in 300 1; do for mix in 32:2133 32:4096 32:8192 64:955 64:2048 16:2048 8:2048; do python tools/tu_mix.py $mix --reps 100 --amp $amp 2>&1 | tail -1; done; done
