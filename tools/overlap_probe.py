#!/usr/bin/env python3
"""Diagnostic: do the latency-bound physics kernel and the MFMA-bound SET GEMMs overlap when two half-batches are
software-pipelined on two streams?  Prints env-steps/s for one 8192-env pipeline vs two 4096-env pipelines."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from bench import WALKERS
from sgrl_amd.rollout import Rollout
from sgrl_amd.set_policy import make_policy
dev = "cuda:0"
torch.manual_seed(1)
policy = make_policy(device=dev).eval()

def run(ros, streams, steps=30, preroll=150):
    for ro in ros:
        ro.reset()
        for _ in range(preroll): ro.step(ro.random_actions())
    torch.cuda.synchronize()
    def it():
        for ro, st in zip(ros, streams):
            with torch.cuda.stream(st):
                obs, rew, done, _ = ro.step(ro.random_actions())
                ro.policy_forward(obs)
    for _ in range(3): it()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(steps): it()
    torch.cuda.synchronize(); dt = time.perf_counter() - t0
    n = sum(ro.env.num_envs for ro in ros)
    return n * steps / dt, dt / steps * 1e3

one = Rollout(WALKERS, 1024, policy=policy, seed=1, device=dev, hold_weights=True)
v, ms = run([one], [torch.cuda.current_stream()])
print("single pipeline 8192 envs: %.0f env-steps/s (%.2f ms/step)" % (v, ms))
del one
import copy
halves = [Rollout(WALKERS, 512, policy=policy if k == 0 else copy.deepcopy(policy), seed=1, device=dev, rank=k, hold_weights=True) for k in range(2)]
v, ms = run(halves, [torch.cuda.Stream(), torch.cuda.Stream()])
print("two pipelines 2 x 4096 envs on two streams: %.0f env-steps/s (%.2f ms per 8192 env-steps)" % (v, ms))
v, ms = run(halves, [torch.cuda.current_stream(), torch.cuda.current_stream()])
print("two pipelines 2 x 4096 envs on ONE stream: %.0f env-steps/s (%.2f ms per 8192 env-steps)" % (v, ms))
