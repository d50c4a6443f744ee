"""k_env_step of several BUILDS of the library on the same batches (each build in its own process: SGRL_HIP_LIB selects the
shared object).  usage: variant_probe.py lib1.so[@ENV=v...][,lib2.so...] [w7|mix|w2|hopper|humanoid|cwhh...]"""
import os, subprocess, sys
REPO = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
if len(sys.argv) > 1 and sys.argv[1] != "--child":
    libs = sys.argv[1].split(",")
    batches = sys.argv[2].split(",") if len(sys.argv) > 2 else ["w7", "mix"]
    for lib in libs:                       # "<path or 'product'>[@ENV=value]..." : a build, optionally with environment settings
        for b in batches:
            env = dict(os.environ)
            parts = lib.split("@")
            for kv in parts[1:]:
                k, v = kv.split("=", 1)
                env[k] = v
            if parts[0] != "product":
                env["SGRL_HIP_LIB"] = os.path.join(REPO, parts[0])
            r = subprocess.run([sys.executable, os.path.abspath(__file__), "--child", b], env=env, capture_output=True, text=True)
            print("%-40s %-4s %s" % (lib, b, (r.stdout.strip().splitlines() or [r.stderr[-300:]])[-1]), flush=True)
    sys.exit(0)
sys.path.insert(0, REPO)
import torch
from sgrl_amd import mjcf
from sgrl_amd.vec_env import BatchedModularVecEnv
which = sys.argv[2]
walkers = sorted(n for n in mjcf.list_assets() if "walker" in n)
if which == "mix":
    names, per = walkers, 1024
elif which == "hopper":
    names, per = sorted(n for n in mjcf.list_assets() if "hopper" in n), 4096 // 3
elif which == "humanoid":
    names, per = sorted(n for n in mjcf.list_assets() if "humanoid" in n), 512
elif which == "cwhh":
    held = {"3d_walker_3_left_knee_right_knee", "3d_walker_6_right_foot", "3d_humanoid_7_left_leg", "3d_humanoid_8_right_knee",
            "3d_cheetah_11_leftbkneen_rightffoot", "3d_cheetah_12_tail_leftffoot"}
    names, per = sorted(n for n in mjcf.list_assets() if n not in held), 356
else:
    L = which[1:]
    names, per = [n for n in walkers if "walker_%s_" % L in n][:1], 8192
env = BatchedModularVecEnv(names, per, seed=1, device="cuda:0")
env.reset_device()
for _ in range(150):
    a = (torch.rand((env.num_envs, env.action_max_len), device="cuda") * 2 - 1).contiguous()
    env.step_device(a)
torch.cuda.synchronize()
ms = [env.time_steps(a, 10) for _ in range(3)]
cnt = env.get_counters()
print("ms per launch %s (lds %d B, groups %d) slab-solve envs %d dropped-row envs %d" % (
    " ".join("%.3f" % m for m in ms), env.lds_bytes, env.launch_groups, int(((cnt[:, 3] >> 16) > 0).sum()), int((cnt[:, 2] > 0).sum())))
