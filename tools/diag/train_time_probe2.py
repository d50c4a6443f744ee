import json, os, sys, time
REPO = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, REPO)
import torch
def tiny_graph_cost(tag):
    x = torch.zeros(1024, device="cuda")
    s = torch.cuda.Stream(); s.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(s):
        for _ in range(3): x.add_(1.0)
    torch.cuda.current_stream().wait_stream(s)
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        for _ in range(1000): x.add_(1.0)
    g.replay(); torch.cuda.synchronize()
    t0 = time.time()
    for _ in range(10): g.replay()
    c = (time.time() - t0) / 10 * 1e3
    torch.cuda.synchronize()
    print("%-46s 1000-node graph replay: cpu %.2f ms  wall %.2f ms" % (tag, c, (time.time() - t0) / 10 * 1e3), flush=True)
tiny_graph_cost("clean process")
from sgrl_amd import mjcf
from sgrl_amd.td3 import default_train_args
from sgrl_amd.train_loop import DeviceTrainer
names = ["3d_walker_7_full", "3d_hopper_3_shin", "3d_humanoid_9_full"]
tr = DeviceTrainer(names, 64, args=default_train_args(), seed=1, device="cuda:0", max_buffer_size=50000, graph_updates=True, tune_gemms=False)
tiny_graph_cost("after DeviceTrainer construction")
tr.warmup(40)
tiny_graph_cost("after warmup (collection)")
tr.update_after_round(max_iters=4)
torch.cuda.synchronize()
tiny_graph_cost("after first graphed update round")
gr = tr.graphed
for k in range(len(names)):
    sl = gr.slots[k]
    torch.cuda.synchronize(); t0 = time.time()
    for i in range(10): sl["graphs"][i % 2].replay()
    c = (time.time() - t0) / 10 * 1e3
    torch.cuda.synchronize()
    print("%-30s update replay cpu %.1f ms wall %.1f ms" % (names[k], c, (time.time() - t0) / 10 * 1e3), flush=True)
    for flag in (0, 1):
        torch.cuda.synchronize(); t0 = time.time()
        for i in range(10): sl["graphs"][flag].replay()
        torch.cuda.synchronize()
        print("    flag %d only: wall %.1f ms" % (flag, (time.time() - t0) / 10 * 1e3), flush=True)
print("env vars:", {k: v for k, v in os.environ.items() if k.startswith(("HIP", "HSA", "ROC", "AMD", "GPU", "PYTORCH"))})
sl = gr.slots[0]
torch.cuda.synchronize()
for i in range(4): sl["graphs"][1].replay()
torch.cuda.synchronize()
