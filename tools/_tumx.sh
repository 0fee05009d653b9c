timeout 800 python -m pytest tests/test_gpu_parity.py -m gpu -q -x -k "tu or fused or workload or golden" 2>&1 | tail -3
for r in 1 2; do echo "== mx repeat=$r"; VVHIP_TU_REPEAT=$r timeout 300 python tools/kbench.py 100 2>&1 | grep -E "TU fused|TU merged|whole step \(" ; done
