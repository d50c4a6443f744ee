#!/usr/bin/env python3
"""Teacher-forced one-step deviation engine vs oracle for every cheetah morphology, alone and inside the 11-morphology batch."""
import os, sys
REPO = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, REPO)
import numpy as np, torch
from sgrl_amd import mjcf
from sgrl_amd.vec_env import BatchedModularVecEnv
from oracle import physics_ref

def probe(names, per=1, steps=6, tag=""):
    env = BatchedModularVecEnv(names, per, seed=5, device="cuda:0")
    env.enable_f64_outputs()
    env.reset_device()
    oes = []
    for i in range(env.num_envs):
        ib, fb = env._blobs[env.env_morph[i]]
        oe = physics_ref.OracleEnv(physics_ref.OracleModel(ib, fb), seed=5, env_id=i); oe.reset(); oes.append(oe)
    rng = np.random.RandomState(0)
    worst = np.zeros(env.num_envs)
    for t in range(steps):
        rec, cnt = env.get_records()
        for i, oe in enumerate(oes):
            m = env.models[env.env_morph[i]]
            rec[i, :m.nq] = oe.qpos; rec[i, m.nq:m.nq + m.nv] = oe.qvel
            rec[i, m.nq + m.nv:m.nq + m.nv + 2] = oe.torso_xy_stale; rec[i, m.nq + m.nv + 2:m.nq + m.nv + 4] = oe.target
            cnt[i, 0], cnt[i, 1] = oe.counters[0], oe.counters[1]
        env.set_records(rec, cnt)
        a = rng.uniform(-1, 1, size=(env.num_envs, env.action_max_len)).astype(np.float32)
        env.step_device(torch.from_numpy(a).cuda(), auto_reset=False)
        torch.cuda.synchronize()
        rec2, cnt2 = env.get_records()
        for i, oe in enumerate(oes):
            o, r, d, info = oe.step(a[i].astype(np.float64), auto_reset=False)
            q, v, xy, tg = env.state_of(rec2, i)
            worst[i] = max(worst[i], np.abs(q - oe.qpos).max(), np.abs(v - oe.qvel).max() / (1 + np.abs(oe.qvel).max()))
            if d:
                oe.counters[1] += 1; oe.reset()
    for i in range(env.num_envs):
        print("%s %-40s nv %2d lds-of-batch %6d groups %d worst %.2e diag %08x" % (tag, names[env.env_morph[i]], env.models[env.env_morph[i]].nv,
              env.lds_bytes, env.launch_groups, worst[i], int(cnt2[i, 3]) & 0xffffffff), flush=True)
    env.close()

names = sorted(n for n in mjcf.list_assets() if "cheetah" in n)
for n in names:
    probe([n], tag="alone ")
probe(names, tag="batch ")
probe(names[::-1], tag="batchR")
