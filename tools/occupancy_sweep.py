#!/usr/bin/env python3
"""k_env_step time vs resident workgroups per CU (walker mix, 8192 envs): SGRL_LDS_PAD inflates the LDS request.
Answers: what would one wave per SIMD cost?  (the price of packing two environments into one wavefront at equal LDS)"""
import os, subprocess, sys
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for pad in (0, 3000, 7000, 13000, 20500, 33000):
    env = dict(os.environ, SGRL_LDS_PAD=str(pad), WARM="150")
    out = subprocess.run([sys.executable, os.path.join(REPO, "tools", "quick_bench.py"), "1024", "10"], env=env, capture_output=True, text=True).stdout
    lds = [l for l in out.splitlines() if "lds_bytes" in l]
    ms = [l for l in out.splitlines() if "hip-event" in l]
    print("pad %6d | %s | %s" % (pad, lds[0] if lds else "?", ms[0] if ms else out[-300:]), flush=True)
