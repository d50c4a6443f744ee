import json, os, sys, time
REPO = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, REPO)
import torch
from sgrl_amd import mjcf
from sgrl_amd.td3 import default_train_args
from sgrl_amd.train_loop import DeviceTrainer
HELD_OUT = {"3d_walker_3_left_knee_right_knee", "3d_walker_6_right_foot", "3d_humanoid_7_left_leg", "3d_humanoid_8_right_knee",
            "3d_cheetah_11_leftbkneen_rightffoot", "3d_cheetah_12_tail_leftffoot"}
names = sorted(n for n in mjcf.list_assets() if n not in HELD_OUT)
tr = DeviceTrainer(names, 64, args=default_train_args(), seed=1, device="cuda:0", max_buffer_size=50000, graph_updates=True, tune_gemms=False)
tr.warmup(40)
tr.update_after_round(max_iters=4)
torch.cuda.synchronize()
ag, gr = tr.agent, tr.graphed
for k, name in enumerate(names):
    ag.change_morphology(tr.graph_dicts[k])
    torch.cuda.synchronize(); t0 = time.time(); tc = 0.0
    for it in range(10):
        c0 = time.time()
        batch = tr.buffers[k].sample(tr.batch_size, generator=tr.gen)
        c1 = time.time()
        gr.update(k, tr.graph_dicts[k], tr.ro.env.num_limbs[k], batch, it)
        tc += time.time() - c0
    cpu = time.time() - t0
    torch.cuda.synchronize()
    print("%-40s L=%2d  wall %.1f ms/update  cpu-issue %.1f ms/update" % (name, tr.ro.env.num_limbs[k], (time.time() - t0) * 100, cpu * 100))
# component timing (CPU issue cost, no syncs in between) for one morphology
k = names.index("3d_walker_7_full"); L = tr.ro.env.num_limbs[k]
ag.change_morphology(tr.graph_dicts[k])
sl = gr.slots[k]
def cpu_cost(fn, n=20):
    torch.cuda.synchronize(); t0 = time.time()
    for i in range(n): fn(i)
    c = (time.time() - t0) / n * 1e3
    torch.cuda.synchronize(); w = (time.time() - t0) / n * 1e3
    return "cpu %.2f ms  wall %.2f ms" % (c, w)
print("sample          ", cpu_cost(lambda i: tr.buffers[k].sample(tr.batch_size, generator=tr.gen)))
b = tr.buffers[k].sample(tr.batch_size, generator=tr.gen)
print("load            ", cpu_cost(lambda i: gr._load(sl, b)))
print("replay          ", cpu_cost(lambda i: sl["graphs"][i % 2].replay()))
print("update (graphed)", cpu_cost(lambda i: gr.update(k, tr.graph_dicts[k], L, b, i)))
import cProfile, pstats
pr = cProfile.Profile(); pr.enable()
for i in range(10): gr.update(k, tr.graph_dicts[k], L, tr.buffers[k].sample(tr.batch_size, generator=tr.gen), i)
pr.disable(); torch.cuda.synchronize()
pstats.Stats(pr).sort_stats("cumulative").print_stats(12)
