"""Per-phase timing of the fused TU kernels: runs tools/kbench.py with the kernels cut after phase k (VVHIP_TU_PHASES).  The cut changes the results, so the
knob only exists in a library built with `make -C vvenc_amd/csrc FLAGS+=-DVVHIP_DEV_KNOBS` (development aid, not part of the product build)."""
import sys, os, subprocess, re
for ph in [1,2,3,4,5,6,7,0]:
    env=dict(os.environ, VVHIP_TU_PHASES=str(ph))
    out=subprocess.run([sys.executable,"tools/kbench.py","30"],env=env,capture_output=True,text=True).stdout
    print("phaseLimit",ph, [re.search(r"([\d.]+) us", l).group(1) for l in out.splitlines() if l.startswith("TU fused")])
