import sys, os, subprocess, re
for ph in [1,2,3,4,5,6,7,0]:
    env=dict(os.environ, VVHIP_TU_PHASES=str(ph))
    out=subprocess.run([sys.executable,"tools/kbench.py","30"],env=env,capture_output=True,text=True).stdout
    print("phaseLimit",ph, [re.search(r"([\d.]+) us", l).group(1) for l in out.splitlines() if l.startswith("TU fused")])
