#!/usr/bin/env python3
"""Who is wrong in tests/test_parity_matrix_gpu.py[cheetah]: the engine, or oracle envs stepped from a thread pool?"""
import os, sys
REPO = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, REPO); sys.path.insert(0, os.path.join(REPO, "tests"))
import numpy as np, torch
from concurrent.futures import ThreadPoolExecutor
from sgrl_amd import mjcf
from sgrl_amd.vec_env import BatchedModularVecEnv
from oracle import physics_ref
POOL = ThreadPoolExecutor(max_workers=16)
names = sorted(n for n in mjcf.list_assets() if "cheetah" in n) + ["3d_cheetah_v2_14_full"]
if len(sys.argv) > 1 and sys.argv[1] == "nov2":
    names = names[:-1]
env = BatchedModularVecEnv(names, 1, seed=5, device="cuda:0")
env.enable_f64_outputs(); env.reset_device()
def mk():
    out = []
    for i in range(env.num_envs):
        ib, fb = env._blobs[env.env_morph[i]]
        oe = physics_ref.OracleEnv(physics_ref.OracleModel(ib, fb), seed=5, env_id=i); oe.reset(); out.append(oe)
    return out
pooled, serial = mk(), mk()
rng = np.random.RandomState(0)
for t in range(12):
    rec, cnt = env.get_records()
    for i, oe in enumerate(serial):
        m = env.models[env.env_morph[i]]
        rec[i, :m.nq] = oe.qpos; rec[i, m.nq:m.nq + m.nv] = oe.qvel
        rec[i, m.nq + m.nv:m.nq + m.nv + 2] = oe.torso_xy_stale; rec[i, m.nq + m.nv + 2:m.nq + m.nv + 4] = oe.target
        cnt[i, 0], cnt[i, 1] = oe.counters[0], oe.counters[1]
    env.set_records(rec, cnt)
    a = rng.uniform(-1, 1, size=(env.num_envs, env.action_max_len)).astype(np.float32)
    env.step_device(torch.from_numpy(a).cuda(), auto_reset=False); torch.cuda.synchronize()
    rec2, cnt2 = env.get_records()
    ods_p = list(POOL.map(lambda ia: ia[1].step(a[ia[0]].astype(np.float64), auto_reset=False), enumerate(pooled)))
    ods_s = [oe.step(a[i].astype(np.float64), auto_reset=False) for i, oe in enumerate(serial)]
    for i in range(env.num_envs):
        q, v, xy, tg = env.state_of(rec2, i)
        e_gs = np.abs(q - serial[i].qpos).max(); e_ps = np.abs(pooled[i].qpos - serial[i].qpos).max()
        flag = "" if max(e_gs, e_ps) < 1e-7 else "  <<<<"
        if flag or t == 0:
            print("t %2d env %2d %-38s |gpu-serial| %.2e |pooled-serial| %.2e done %s/%s%s" % (t, i, names[i], e_gs, e_ps, ods_s[i][2], ods_p[i][2], flag), flush=True)
        for oe, od in ((serial[i], ods_s[i]), (pooled[i], ods_p[i])):
            if od[2]:
                oe.counters[1] += 1; oe.reset()
