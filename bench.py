#!/usr/bin/env python3
"""bench.py -- headline benchmark of the rollout hot path on MI355X.

Metric (BASELINE.json): env-steps/sec (whole job) + SET-actor forward us/step on the 3d_walker 8-variant mix at
8192 envs per GPU (config 3).  One "step" = one pass of the hot path over the whole batch:
    random U(-1,1) actions (device RNG)  ->  engine step (4 x mj_step, obs scatter, reward/done, auto-reset)
    ->  batched SET actor forward on the new observations  ->  [N > 1: replay-block gather to rank 0 over RCCL]
Inputs are resident in HBM when the timed region starts.  Weak scaling: every rank owns its own 8192 envs.

    python bench.py --gpus 1 --steps 20 --warmup 3
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P \
        bench.py --gpus N --steps K --warmup W
Rank 0 prints ONE JSON line.
"""
import argparse
import json
import os
import sys
import time

REPO = os.path.dirname(os.path.abspath(__file__))
if REPO not in sys.path:
    sys.path.insert(0, REPO)

WALKERS = sorted(["3d_walker_2_right_leg_left_knee", "3d_walker_3_left_leg_right_foot", "3d_walker_3_left_knee_right_knee",
                  "3d_walker_4_right_knee_left_foot", "3d_walker_5_foot", "3d_walker_5_left_knee",
                  "3d_walker_6_right_foot", "3d_walker_7_full"])
HBM_PEAK_GBS = 8000.0


def algorithmic_bytes_per_env_step(env):
    """HBM bytes one env-step must move: state record in + out (f64), counters, action row (f32),
    observation row (f32, padded to obs_max_len -- that is what is written), reward/done/dist/truncated."""
    import numpy as np
    total = 0
    for m, c in zip(env.models, env.counts):
        rec = 8 * (m.nq + m.nv + 4)
        total += c * (2 * rec + 2 * 16 + 4 * env.action_max_len + 4 * env.obs_max_len + 4 + 1 + 4 + 1)
    return total / float(sum(env.counts))


def _cpu_sample(names, seed, per_morph, steps, cores, pool_cls, mp_ctx):
    """One bounded sample: `per_morph` oracle envs per walker variant stepped `steps` times on the host cores, then the
    PyTorch-CPU SEPolicy forward batched per morphology (B = per_morph) for the same number of env-steps."""
    import torch
    from sgrl_amd.set_policy import make_policy
    from sgrl_amd import graph as G, mjcf
    jobs = [(n, i, seed, steps) for n in names for i in range(per_morph)]
    t0 = time.time()
    with pool_cls(max_workers=min(cores, len(jobs)), mp_context=mp_ctx) as pool:
        done = list(pool.map(_cpu_worker, jobs, timeout=120))
    t_env = time.time() - t0
    torch.set_num_threads(min(cores, 16))
    pol = make_policy(use_hip=False).eval()
    t1 = time.time()
    with torch.no_grad():
        for n in names:
            m = mjcf.load_asset(n)
            pol.change_morphology(G.getGraphDict(m.parents, ["pre", "inlcrs", "postlcrs"], [], device=torch.device("cpu")))
            x = torch.randn(per_morph, 41 * m.num_limbs)
            for _ in range(steps):
                pol(x)
    t_set = time.time() - t1
    return sum(done), t_env, t_set


def _reference_shaped(names, seed, steps, pool_cls, mp_ctx):
    """The reference's own shape (reference src/trainer.py:173-200): n = #morphologies environments, one worker process
    each (SubprocVecEnv), and per time step one B = 1 `select_action` forward per environment on the main process."""
    import torch
    from sgrl_amd.set_policy import make_policy
    from sgrl_amd import graph as G, mjcf
    jobs = [(n, 0, seed, steps) for n in names]
    t0 = time.time()
    with pool_cls(max_workers=len(jobs), mp_context=mp_ctx) as pool:
        done = list(pool.map(_cpu_worker, jobs, timeout=120))
    t_env = time.time() - t0
    pol = make_policy(use_hip=False).eval()
    gds = [(G.getGraphDict(mjcf.load_asset(n).parents, ["pre", "inlcrs", "postlcrs"], [], device=torch.device("cpu")),
            mjcf.load_asset(n).num_limbs) for n in names]
    t1 = time.time()
    with torch.no_grad():
        for _ in range(steps):
            for gd, L in gds:
                pol.change_morphology(gd)
                pol(torch.randn(1, 41 * L))
    t_set = time.time() - t1
    return sum(done) / (t_env + t_set), t_env, t_set


def cpu_baseline(names, seed):
    """the CPU oracle (oracle/physics.c, FP64, same algorithm) + PyTorch-CPU SEPolicy on the host
    cores over a bounded sample of the same workload, median of 3 samples, plus the "reference-shaped" figure (8 envs,
    B = 1 forwards).  kind = "port": the reference's own stack (MuJoCo 2.1 + gym) is not installable on the box."""
    import multiprocessing as mp
    from concurrent.futures import ProcessPoolExecutor
    from oracle import physics_ref
    cores = os.cpu_count() or 1
    per_morph, steps = 16, 80
    physics_ref.lib()   # build/load the checker ONCE in the parent; the forked workers inherit it
    ctx = mp.get_context("fork")
    samples = []
    for rep in range(3):
        n_steps, t_env, t_set = _cpu_sample(names, seed + rep, per_morph, steps, cores, ProcessPoolExecutor, ctx)
        samples.append((n_steps / (t_env + t_set), n_steps / t_env, t_env, t_set))
    samples.sort()
    med = samples[1]
    ref_rate, ref_env, ref_set = _reference_shaped(names, seed, 100, ProcessPoolExecutor, ctx)
    nproc = min(cores, per_morph * len(names))
    return {"value": round(med[0], 1), "unit": "env-steps/s", "cores": nproc, "kind": "port",
            "sample": "median of 3 samples of %d envs (%d per walker variant) x %d steps: oracle/physics.c FP64 step on %d "
                      "processes (%.1f s) + PyTorch-CPU SEPolicy forward B=%d per morphology (%.1f s); %d steps per sample replace "
                      "BASELINE.md section 3's T = 1000 so that the default run fits the driver's clock" % (
                          per_morph * len(names), per_morph, steps, nproc, med[2], per_morph, med[3], steps),
            "samples": [round(x[0], 1) for x in samples],
            "env_only_steps_per_s": round(med[1], 1),
            "reference_shaped": {"value": round(ref_rate, 1), "unit": "env-steps/s", "cores": len(names) + 1,
                                 "sample": "%d envs (one per walker variant, one process each) x 100 steps (%.2f s) + one B=1 "
                                           "SEPolicy forward per env per step on the main process (%.1f s): the shape of "
                                           "reference trainer.py:173-200" % (len(names), ref_env, ref_set)}}


def _cpu_worker(job):
    import numpy as np
    from oracle import physics_ref
    from sgrl_amd import mjcf, model_pack
    from sgrl_amd.env_spec import env_spec_for
    name, idx, seed, steps = job
    m = mjcf.load_asset(name)
    ib, fb = model_pack.pack_model(m, spec=env_spec_for(name))
    env = physics_ref.OracleEnv(physics_ref.OracleModel(ib, fb), seed=seed, env_id=idx)
    env.reset()
    rng = np.random.RandomState(idx)
    for _ in range(steps):
        env.step(rng.uniform(-1, 1, size=3 * m.num_limbs))
    return steps


# multiply-accumulates the SET forward EXECUTES per limb node: with the Gram matrix taken over its blocked lower triangle (K = 576 instead of 1024 on the
# seven Gram-fed layers), with the projections as the zero-padded stacked GEMM operands the kernels really run, and after
# folding ng_out / g_out into the value projections (those two GEMMs per layer no longer exist)
def set_executed_flops_per_node():
    layer = (3 * 128 * 32) + 576 * 256 + 256 * 128 + 256 * 768 + 3 * 128 * 256                                     # attention
    layer += (3 * 128 * 64) + 576 * 256 + 256 * 128 + 2 * (256 * 256) + 256 * 128 + 256 * 1024 + 3 * 32 * 32 + 3 * 32 * 128
    # head: decoder_g is folded through linear2_m (the 1024-wide product is a 32-wide one: include/sgrl_set.h)
    head = (3 * 144 * 64) + 576 * 128 + 128 * 128 + 160 * 128 + 128 * 128 + 256 * 256 + 256 * 32 + 3 * 32
    return 2 * (3 * layer + head)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--envs-per-morph", type=int, default=1024)
    ap.add_argument("--seed", type=int, default=1)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--force-collectives", action="store_true",
                    help="diagnostic: run the N > 1 code path (process group, replay gather, barriers, max-over-ranks) with "
                         "WORLD_SIZE=1 under torch.distributed.run -- what a 1-GPU box can exercise of it")
    ap.add_argument("--set-forward-only", action="store_true",
                    help="internal: time the SET forward alone on this process's settings and print {\"ms_per_forward\": x} "
                         "(the parent bench starts it as a child with SGRL_SET_GEMM=f32 for the exact-f32 comparison)")
    ap.add_argument("--regions", type=int, default=3, help="timed regions of --steps steps each; the median is reported")
    ap.add_argument("--preroll", type=int, default=200,
                    help="untimed rollout steps run during set-up so that episodes are desynchronised and the timed "
                         "steps see the stationary mix of flight / stance / fallen states (not warm-up of the code)")
    args = ap.parse_args()
    import faulthandler
    faulthandler.dump_traceback_later(900, exit=True)   # never hang a GPU box: dump stacks and exit

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    # The CPU baseline forks worker processes: it must run BEFORE this process touches the GPU (a process that
    # has initialised HIP must neither fork-and-use nor exec).  Rank 0 at N=1 only.
    cpu_base = None
    if args.set_forward_only:
        args.no_cpu_baseline = True
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        cpu_base = cpu_baseline(WALKERS, args.seed)
    import torch
    import torch.distributed as dist
    if world != args.gpus:
        if rank == 0:
            sys.stderr.write("warning: --gpus %d but WORLD_SIZE=%d; using WORLD_SIZE\n" % (args.gpus, world))
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X: torch.cuda.is_available() is False (no CPU fallback)")
    torch.cuda.set_device(local_rank)
    dev = "cuda:%d" % local_rank
    multi = world > 1 or args.force_collectives
    if multi:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group(backend="nccl", device_id=torch.device(dev))

    from sgrl_amd.rollout import Rollout, ReplayGather
    from sgrl_amd.set_policy import make_policy
    torch.manual_seed(args.seed)
    policy = make_policy(device=dev).eval()          # random-init weights of the reference architecture
    # the policy's weights never change during this rollout: the actor packs them once (include/sgrl_set.h sgrl_set_hold_weights; the
    # training loop holds them the same way between two rounds of updates)
    ro = Rollout(WALKERS, args.envs_per_morph, policy=policy, seed=args.seed, device=dev, rank=rank, hold_weights=True)
    env = ro.env
    n_local = env.num_envs
    if args.set_forward_only:
        ro.reset()
        for _ in range(50):
            ro.step(ro.random_actions())
        ms = ro.actor.time_forward(env.obs, ro.policy_actions, 5)
        ms = min(ms, ro.actor.time_forward(env.obs, ro.policy_actions, 5))
        print(json.dumps({"ms_per_forward": round(ms, 4), "SGRL_SET_GEMM": os.environ.get("SGRL_SET_GEMM", "")}))
        return
    gather = ReplayGather(n_local, env.obs_max_len, env.action_max_len, dev, depth=2) if multi else None
    # N > 1: the learner rank (0) also INGESTS the N gathered blocks of every step into its per-morphology replay rings
    # (reference common/buffer.py:75-84 x N x envs per rank): one sgrl_ingest_rows launch per block, no host synchronisation
    # (rollout.TransitionSink).  The blocks of step t are ingested during step t + 1, once their gather has completed.
    sink = None
    if multi:
        from sgrl_amd.replay import DeviceReplayBuffer
        from sgrl_amd.rollout import TransitionSink
        buffers = None
        if rank == 0:
            cap = max(4 * world * args.envs_per_morph, 65536)
            buffers = [DeviceReplayBuffer(41 * L, 3 * L, cap, device=dev) for L in env.num_limbs]
        sink = TransitionSink(env.env_morph, env.num_limbs, env.obs_max_len, env.action_max_len, device=dev, buffers=buffers, dst=0)
    state = {"pending": None, "ingest": True}

    def ingest_pending():
        if state["pending"] is not None:
            slot, blocks = state["pending"]
            gather._wait(slot)             # the gather that filled these blocks (issued one step ago)
            if state["ingest"] and rank == 0:
                sink.ingest(blocks)
            state["pending"] = None

    def one_step():
        a = ro.random_actions()
        if gather is not None:
            gather.stage_obs(env.obs)      # the row's observation half, before the step overwrites env.obs in place
        obs, rew, done, _ = ro.step(a)
        ro.policy_forward(obs)
        if gather is not None:
            ingest_pending()               # last step's blocks: their transfer ran under this step's kernels
            gather.pack(None, a, obs, rew, done, morph_id=sink.env_morph)
            slot = gather._k % gather.depth
            blocks = gather.push(wait=False)   # in flight over xGMI while the next step runs; its block is reused two steps on
            state["pending"] = (slot, blocks) if blocks is not None else None

    ro.reset()
    for _ in range(args.preroll):          # synthetic-state preparation: reach the stationary episode mix
        ro.step(ro.random_actions())
    for _ in range(args.warmup):
        one_step()

    def barrier():
        if gather is not None:
            ingest_pending()
            gather.drain()                 # every replay block of the timed steps has arrived (and is ingested) before the clock stops
        if multi:
            dist.barrier()
        torch.cuda.synchronize()

    def timed_region():
        barrier()
        t0 = time.perf_counter()
        for _ in range(args.steps):
            one_step()
        barrier()
        dt = time.perf_counter() - t0
        if multi:
            tt = torch.tensor([dt], dtype=torch.float64, device=dev)
            dist.all_reduce(tt, op=dist.ReduceOp.MAX)
            dt = float(tt.item())
        return dt

    # `regions` timed regions of exactly `steps` steps each (barrier + synchronize on both sides, max over ranks); the MEDIAN is
    # what `value` / `ms_per_step` report, all samples are listed
    samples = [timed_region() for _ in range(max(1, args.regions))]
    dt = sorted(samples)[len(samples) // 2]
    dt_no_ingest = None
    if multi:                              # the same region with the learner's ingest switched off: what the ingest costs
        state["ingest"] = False
        dt_no_ingest = timed_region()
        state["ingest"] = True

    # per-kernel timings with HIP events on the launch stream (rank 0 only; not part of the timed region above)
    extra = {}
    if rank == 0:
        # the step kernel INSIDE the rollout's flow: one more region of `steps` steps with a HIP event pair around every sgrl_step
        # (on torch's current stream, the stream sgrl_step launches on) -- fresh random actions, the forward between two steps
        evs = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(args.steps)]
        if gather is not None:             # rank-local only: NO collective may run in this rank-0 block (the other ranks are not here)
            ingest_pending()
            gather.drain()
        torch.cuda.synchronize()
        for e0, e1 in evs:
            a = ro.random_actions()
            if gather is not None:
                gather.stage_obs(env.obs)
            e0.record()
            obs, rew, done, _ = ro.step(a)
            e1.record()
            ro.policy_forward(obs)
            if gather is not None:
                gather.pack(None, a, obs, rew, done, morph_id=sink.env_morph)
        torch.cuda.synchronize()
        ms_step_in_rollout = sum(e0.elapsed_time(e1) for e0, e1 in evs) / len(evs)
        a = ro.random_actions()
        ms_step = env.time_steps(a, 10)
        ms_set = ro.actor.time_forward(env.obs, ro.policy_actions, 5)
        bytes_step = algorithmic_bytes_per_env_step(env)
        achieved = bytes_step * n_local / (ms_step * 1e-3) / 1e9
        traffic, traffic_src = None, None
        pmc = os.path.join(REPO, "profiles", "pmc_traffic.json")
        if os.path.exists(pmc):   # FETCH_SIZE/WRITE_SIZE of k_env_step from separate rocprofv3 --pmc passes of this command
            with open(pmc) as f:
                pj = json.load(f)
            if pj.get("envs_per_gpu") == n_local:
                traffic = pj.get("k_env_step_bytes_per_launch")
                traffic_src = "profiles/pmc_traffic.json: FETCH_SIZE + WRITE_SIZE per launch from two rocprofv3 --pmc passes " \
                              "of this command (%s); not re-measured in this run" % pj.get("build", "see profiles/README.md")
        extra["roofline"] = {"bound": "hbm", "kernel": "k_env_step", "achieved": round(achieved, 3), "peak": HBM_PEAK_GBS,
                             "unit": "GB/s", "frac": round(achieved / HBM_PEAK_GBS, 6), "traffic": traffic, "traffic_source": traffic_src,
                             "algorithmic_bytes_per_launch": int(bytes_step * n_local),
                             "ms_per_launch": round(ms_step, 4),
                             "ms_per_launch_in_rollout": round(ms_step_in_rollout, 4),
                             "achieved_in_rollout": round(bytes_step * n_local / (ms_step_in_rollout * 1e-3) / 1e9, 3),
                             "dispatches_per_launch": env.launch_groups, "fixed_dimension_kernels": env.fixed_dim_groups,
                             "note": "latency/VALU-bound FP64 rigid-body kernel: ~2 KB of HBM traffic per env-step "
                                     "against ~1e6 FP64 operations; see DESIGN.md (roofline).  One launch = one "
                                     "sgrl_step = dispatches_per_launch concurrent step-kernel dispatches (one per kernel "
                                     "family / LDS occupancy class; this workload: ONE dispatch of the walker family's "
                                     "fixed-dimension kernel k_env_step_spec, csrc/step_spec.hip); ms_per_launch (what `achieved` and "
                                     "`frac` are made of) is the HIP-event mean of ten back-to-back launches on ONE repeated action after "
                                     "the run; ms_per_launch_in_rollout is the HIP-event mean around every sgrl_step of one more region "
                                     "of `steps` rollout steps (fresh actions, the forward between two steps): the time ms_per_step is "
                                     "made of, and what the rocprofv3 kernel trace of this command averages"}
        # the VALU view of the same kernel (it is issue / latency bound, not HBM bound): SQ counters of a separate
        # rocprofv3 --pmc pass of this command, committed under profiles/
        sq = os.path.join(REPO, "profiles", "sq_pmc.json")
        if os.path.exists(sq):
            with open(sq) as f:
                sj = json.load(f)
            if sj.get("envs_per_gpu") == n_local:
                extra["roofline_valu"] = sj.get("k_env_step")
                rv = extra["roofline_valu"]
                if rv and "valu_issue_fraction" in rv and "active_lane_fraction" in rv:
                    # the share of the chip's FP64 lane-slots the kernel fills: cycles a SIMD issues a vector instruction x lanes live in it
                    rv["fp64_lane_slot_fraction"] = round(rv["valu_issue_fraction"] * rv["active_lane_fraction"], 4)
                    rv["source"] = "profiles/sq_pmc.json (build %s): separate rocprofv3 --pmc passes of this command; not re-measured in this run" % sj.get("build")
        nodes = ro.actor.num_nodes
        ex = set_executed_flops_per_node()
        # outside the timed region: the forward on this run's last observations in BOTH product forms (two-piece f16 = what was
        # timed, three-piece bf16 = f32's full exponent range); the largest difference between the two action tensors
        forms_diff = None
        if os.environ.get("SGRL_SET_GEMM", "")[:1] not in ("b", "f"):
            a_h = ro.actor.forward_batch(env.obs, act_ld=env.action_max_len).clone()
            ro.actor.gemm_form(ro.actor.FORM_BF16X6)
            a_b = ro.actor.forward_batch(env.obs, act_ld=env.action_max_len)
            ro.actor.gemm_form(0)
            forms_diff = float((a_h - a_b).abs().max())
        # the same forward WITHOUT the weight hold (k_pack + k_encode_rows at the top of every call: what round 3 timed, and what a
        # caller that cannot promise constant weights pays)
        ms_unheld = None
        if getattr(ro, "holds_weights", False):
            ro.actor.hold_weights(False)
            ms_unheld = ro.actor.time_forward(env.obs, ro.policy_actions, 5)
            ro.actor.hold_weights(True)
        extra["set_actor"] = {"ms_per_forward": round(ms_set, 4), "us_per_env_step": round(ms_set * 1e3 / n_local, 4),
                              "ms_per_forward_weights_not_held": None if ms_unheld is None else round(ms_unheld, 4),
                              "weights_pack_ms_hoisted": None if ms_unheld is None else round(ms_unheld - ms_set, 4),
                              "nodes": nodes, "nominal_flops_per_node": 10.07e6, "executed_flops_per_node": ex,
                              "tflops_nominal": round(nodes * 10.07e6 / (ms_set * 1e-3) / 1e12, 2),
                              "tflops_executed": round(nodes * ex / (ms_set * 1e-3) / 1e12, 2),
                              "mfma_f16_dense_peak_tflops": 2500.0,
                              "frac_of_f16_mfma_peak_executed_x3": round(3 * nodes * ex / (ms_set * 1e-3) / 2.5e15, 4),
                              "product_form": os.environ.get("SGRL_SET_GEMM", "f16x3"),
                              "fused_chains": os.environ.get("SGRL_SET_CHAIN", "1") != "0",
                              "product_launches_per_forward_by_construction": 25 if os.environ.get("SGRL_SET_CHAIN", "1") != "0" else 42,
                              "weights_held": bool(getattr(ro, "holds_weights", False)),
                              "row_scale_tile_repeats": ro.actor.scale_redos(reset=False),
                              "max_action_diff_between_product_forms": forms_diff,
                              "note": "nominal = the reference's dense layer sizes; executed = what the kernels run with the symmetric Gram "
                                      "matrix taken over the 36 4x4 blocks of its lower triangle (K = 576 instead of 1024; the "
                                      "operand is generated inside the GEMM, never stored) and the attention output "
                                      "projections folded into the value projections.  The GEMMs are float32 products (f32 in, f32 out, error "
                                      "against float64 at or below an f32 FMA chain's: tests/test_split_products_gpu.py, tools/chain_lab.hip r) carried by the 16-bit matrix "
                                      "cores: every operand row is scaled by a power of two into f16's range (exact, undone in the epilogue: "
                                      "float32's exponent range, nothing clamped) and cut into two f16 pieces, three matrix instructions per "
                                      "product block (gemm_f32.h; SGRL_SET_GEMM=bf16x6 selects the three-piece bf16 form); back-to-back products "
                                      "run as one kernel each (chain_f16.h), decoder_g is folded through linear2_m; weights_held: the rollout "
                                      "promised constant weights, so the forward packs them once, not per call (sgrl_set_hold_weights: what the "
                                      "training loop does between two rounds of updates); row_scale_tile_repeats = workgroups that had to "
                                      "repeat a tile with exact row maxima (0 = every sampled estimate held).  "
                                      "frac_of_f16_mfma_peak_executed_x3 = 3 matrix instructions per product block x executed flops / time / the "
                                      "dense f16 peak (2.5 PF): the share of the matrix pipe's peak the forward's matrix work amounts to; "
                                      "weights_pack_ms_hoisted = what the weight hold moved out of the timed forward (ms_per_forward_weights_not_held "
                                      "- ms_per_forward); product_launches_per_forward_by_construction: the count of set_actor.hip forward(), "
                                      "checked against the kernel trace in profiles/<tag>_kernel_stats.csv, not measured in this run"}
        # the exact-f32 forward next to the two-piece one: a child process with SGRL_SET_GEMM=f32 (plain products on
        # v_mfma_f32_32x32x2_f32, the reference's arithmetic; generated-operand products bf16 x 6) -- outside the timed region
        exact = None
        if world == 1 and not args.force_collectives and os.environ.get("SGRL_BENCH_NO_CHILD", "") != "1":
            import subprocess
            try:
                r = subprocess.run([sys.executable, os.path.abspath(__file__), "--set-forward-only", "--envs-per-morph", str(args.envs_per_morph)],
                                   env=dict(os.environ, SGRL_SET_GEMM="f32", SGRL_BENCH_NO_CHILD="1"), capture_output=True, text=True, timeout=300)
                exact = json.loads([l for l in r.stdout.splitlines() if l.startswith("{")][-1])["ms_per_forward"]
            except Exception as e:       # diagnostics only: never fail the bench line over it
                exact = "unavailable: %r" % (e,)
        ro.actor.gemm_form(ro.actor.FORM_BF16X6)
        ms_b6 = ro.actor.time_forward(env.obs, ro.policy_actions, 5)
        ro.actor.gemm_form(0)
        extra["set_actor"]["ms_per_forward_bf16x6"] = round(ms_b6, 4)
        extra["set_actor"]["ms_per_forward_exact_f32"] = exact
        if isinstance(exact, float):
            extra["set_actor"]["env_steps_per_s_with_exact_f32_forward"] = round(n_local / ((ms_step + exact) * 1e-3), 1)
            extra["set_actor"]["exact_f32_note"] = ("child process with SGRL_SET_GEMM=f32: plain products on v_mfma_f32_32x32x2_f32 "
                                                    "(exact f32), generated-operand products (Gram, equivariant) bf16 x 6; the rate "
                                                    "is n_envs / (k_env_step ms_per_launch + that forward)")
        # replay ingest on the learner, priced on this GPU (outside the timed region at N = 1): one block of this rank's rows
        try:
            from sgrl_amd.replay import DeviceReplayBuffer
            from sgrl_amd.rollout import TransitionSink, ReplayGather as _RG
            bufs = [DeviceReplayBuffer(41 * L, 3 * L, 65536, device=dev) for L in env.num_limbs]
            sk = sink if (sink is not None and sink.buffers is not None) else TransitionSink(
                env.env_morph, env.num_limbs, env.obs_max_len, env.action_max_len, device=dev, buffers=bufs, dst=0)
            g1 = sk.gather if sk is not sink else gather
            blk = g1.pack(env.obs, ro.actions, env.obs, env.rew, env.done, morph_id=sk.env_morph)
            for _ in range(3):
                sk.ingest([blk])
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(10):
                sk.ingest([blk])
            e1.record()
            torch.cuda.synchronize()
            ms_ing = e0.elapsed_time(e1) / 10
            # the learner's ingest at N = 8 as it would run: the eight ranks' blocks arrive in ONE contiguous tensor (ReplayGather
            # recv_flat) and are ingested by one sgrl_ingest_block call = three launches, whatever N -- measured here on 8 copies
            flat8 = blk.repeat(8, 1).contiguous()
            bufs8 = [DeviceReplayBuffer(41 * L, 3 * L, 8 * n_local, device=dev) for L in env.num_limbs]
            sk8 = TransitionSink(env.env_morph, env.num_limbs, env.obs_max_len, env.action_max_len, device=dev, buffers=bufs8, dst=0)
            for _ in range(3):
                assert sk8._ingest_block_hip(flat8)
            e0.record()
            for _ in range(10):
                sk8._ingest_block_hip(flat8)
            e1.record()
            torch.cuda.synchronize()
            ms_ing8 = e0.elapsed_time(e1) / 10
            extra["replay_ingest"] = {"ms_ingest_per_block": round(ms_ing, 4), "rows_per_block": n_local,
                                      "projected_learner_ms_per_step_at_8_gpus": round(ms_ing8, 3),
                                      "note": "learner-side ingest (all rows stored): sgrl_ingest_block = k_ingest_keys + k_ingest_slots + k_ingest_rows, "
                                              "slots on the device, no host synchronisation.  ms_ingest_per_block: one rank's block; "
                                              "projected_learner_ms_per_step_at_8_gpus: MEASURED on this GPU on a contiguous tensor of "
                                              "8 such blocks (what the learner receives at N = 8: one call, three launches), inside the "
                                              "timed region of an N > 1 run"}
        except Exception as e:
            extra["replay_ingest"] = {"error": repr(e)}
        rec, cnt = env.get_records()
        extra["row_overflow_envs"] = int((cnt[:, 2] > 0).sum())
        if cpu_base is not None:
            extra["cpu_baseline"] = cpu_base
    if rank == 0:
        total_envs = n_local * world
        out = {
            "metric": "env-steps/sec (whole node) + SET-actor fwd us/step, 3d_walker mix @8192 envs/GPU",
            "value": round(total_envs * args.steps / dt, 1),
            "unit": "env-steps/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(dt / args.steps * 1e3, 4),
            "ms_per_step_samples": [round(x / args.steps * 1e3, 4) for x in samples], "timed_regions": len(samples),
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "f64 (dynamics) / f32 (SET actor, obs)", "data": "synthetic",
            "config": {"workload": "3D_Walker++ 8 variants x %d envs per GPU (config 3), random U(-1,1) actions, "
                                   "auto-reset, SET actor forward on every step, random-init weights" % args.envs_per_morph,
                       "envs_per_gpu": n_local, "obs_max_len": env.obs_max_len, "action_max_len": env.action_max_len,
                       "replay_gather": "torch.distributed.gather (RCCL) of %d B/rank/step, in flight during the next step (2 blocks in turn)" % gather.bytes_per_step()
                       if gather is not None else "none (single rank)"},
        }
        if dt_no_ingest is not None:
            out["learner_ingest"] = {"in_timed_region": True, "blocks_per_step_on_rank0": world,
                                     "ms_per_step_without_ingest": round(dt_no_ingest / args.steps * 1e3, 4),
                                     "value_without_ingest": round(total_envs * args.steps / dt_no_ingest, 1)}
        out.update(extra)
        print(json.dumps(out))
    if multi:
        dist.barrier()                     # rank 0's post-run kernel timings are done before any rank tears NCCL down
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
