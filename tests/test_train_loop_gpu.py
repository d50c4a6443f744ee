"""Config 5 assembled on the device (sgrl_amd/train_loop.py): Rollout -> RoundCollector -> ReplayGather -> per-morphology
DeviceReplayBuffer -> TD3 update -> next round.  a15 / a16 are checked ON THE GPU against a per-environment Python replay
of the reference's loop (reference src/trainer.py:173-236) fed with the step outputs the engine actually produced -- the
same technique as the evaluator test in test_dropin_loop_gpu.py."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu

NAMES = ["3d_cheetah_10_tail_leftbleg", "3d_hopper_3_shin", "3d_humanoid_7_left_arm", "3d_walker_4_right_knee_left_foot"]
PER = [2, 3, 2, 3]


def _trainer(max_episode_steps=40, seed=3, lag_flag=False):
    """lag_flag=False: the immediate round-finished flag, which the step-by-step re-enactments below follow (the trainer's default
    reads it one step late: test_lagged_round_flag_stores_the_same_rows)."""
    import torch
    from sgrl_amd.td3 import default_train_args
    from sgrl_amd.train_loop import DeviceTrainer
    assert torch.cuda.is_available()
    args = default_train_args(max_episode_steps=max_episode_steps, batch_size=16)
    return torch, DeviceTrainer(NAMES, PER, args=args, seed=seed, device="cuda:0", max_buffer_size=256, lag_flag=lag_flag)


class _Tape(object):
    """Records what went into TransitionSink.push (host copies), step by step."""

    def __init__(self, trainer):
        self.rows = []
        sink = trainer.sink
        orig = sink.push

        def push(prev_obs, action, next_obs, reward, done):
            self.rows.append(tuple(t.detach().cpu().numpy().copy() for t in (prev_obs, action, next_obs, reward, done)))
            return orig(prev_obs, action, next_obs, reward, done)
        sink.push = push


def _reenact(tape_rows, env_morph, limbs, max_steps, round_starts):
    """reference trainer.py:205-236 per env, scalar code; `round_starts` = step indices at which a new round began."""
    n = len(env_morph)
    rows = [[] for _ in limbs]
    done_list, steps = [False] * n, [0] * n
    finished_at = []
    for t, (obs, act, nxt, rew, done) in enumerate(tape_rows):
        if t in round_starts:
            done_list, steps = [False] * n, [0] * n
        for i in range(n):
            curr = bool(done[i])
            done_bool = float(curr)
            if steps[i] + 1 == max_steps:
                done_bool, curr = 0.0, True
            if not done_list[i]:
                steps[i] += 1
                L = limbs[env_morph[i]]
                rows[env_morph[i]].append((obs[i, :41 * L], act[i, :3 * L], nxt[i, :41 * L], rew[i], done_bool))
                done_list[i] = done_list[i] or curr
        if all(done_list):
            finished_at.append(t)
    return rows, finished_at


def test_collection_rounds_fill_the_per_morphology_buffers_like_the_reference_loop():
    torch, tr = _trainer()
    tape = _Tape(tr)
    env = tr.ro.env
    round_starts, finished = {0}, []
    # warm-up with uniform actions (trainer.py:90-138), then policy rounds with exploration noise (:173-196)
    for t in range(60):
        if tr.collect_step(random_actions=True):
            finished.append(len(tape.rows) - 1)
            tr.begin_round()
            round_starts.add(len(tape.rows))
    for t in range(60):
        if tr.collect_step():
            finished.append(len(tape.rows) - 1)
            tr.begin_round()
            round_starts.add(len(tape.rows))
    rows, fin_ref = _reenact(tape.rows, list(env.env_morph), env.num_limbs, 40, round_starts)
    assert finished == fin_ref and len(finished) >= 2
    assert tr.tot_env_steps == sum(len(r) for r in rows) == tr.sink.stored
    for k, L in enumerate(env.num_limbs):
        st = tr.buffers[k].state_arrays()
        total, cap = len(rows[k]), 256
        assert st["curr"] == total % cap and st["max_sample_size"] == min(total, cap) and total > 20
        for j in range(max(0, total - cap), total):
            o, a, nx, rw, d = rows[k][j]
            p = j % cap
            assert np.array_equal(st["obs_buffer"][p], o) and np.array_equal(st["action_buffer"][p], a), (k, j)
            assert np.array_equal(st["next_obs_buffer"][p], nx) and st["reward_buffer"][p] == rw and st["done_buffer"][p] == d
        # padding slots and the first three (torso) action slots: the reference stores what the driver produced
        assert st["obs_buffer"].shape[1] == 41 * L and st["action_buffer"].shape[1] == 3 * L
    # time-limit rows were stored with done = 0 (trainer.py:209-212) and terminal ones with done = 1
    alld = np.concatenate([tr.buffers[k].state_arrays()["done_buffer"][:tr.buffers[k].max_sample_size] for k in range(4)])
    assert (alld == 1).any() and (alld == 0).any()
    # policy rounds: the stored actions are clipped exploration-noised policy outputs, zero beyond 3 L in the padded tensor
    a_last = tape.rows[-1][1]
    assert np.abs(a_last).max() <= 1.0
    for i, k in enumerate(env.env_morph):
        assert (a_last[i, 3 * env.num_limbs[k]:] == 0).all()


def test_training_round_updates_the_policy_and_the_rollout_follows():
    torch, tr = _trainer(max_episode_steps=30, seed=5)
    tr.warmup(80)
    assert all(b.max_sample_size >= 16 for b in tr.buffers)
    obs = tr.ro.env.obs.clone()
    before = [p.detach().clone() for p in tr.agent.actor.parameters()]
    a0 = tr.ro.policy_forward(obs).clone()
    out = tr.train_round(max_steps=200, max_iters=3)
    assert out["per_morph_iter"] >= 1 and out["steps"] >= 1
    assert tr.rounds == 1
    for name in NAMES:
        loss = tr.last_losses[name]
        assert np.isfinite(float(loss["loss/critic_loss"]))
    moved = max(float((p - q).abs().max()) for p, q in zip(tr.agent.actor.parameters(), before))
    assert moved > 0                                           # policy_freq = 2: it = 0 (and 2) updated the actor
    # the batched HIP actor of the rollout reads the optimizer's in-place updates: same numbers as the PyTorch path now
    a1 = tr.ro.policy_forward(obs).clone()
    assert float((a1 - a0).abs().max()) > 0
    env = tr.ro.env
    pol = tr.agent.actor
    pol.use_hip = False
    with torch.no_grad():
        for k, sl in enumerate(env.morph_slices):
            L = env.num_limbs[k]
            pol.change_morphology(tr.graph_dicts[k])
            ref = pol(obs[sl, :41 * L])
            assert float((ref - a1[sl, :3 * L]).abs().max()) < 2e-5, NAMES[k]
    pol.use_hip = True
    # targets track the online networks by tau (agent.py:185-187)
    d_on = sum(float((p - q).abs().sum()) for p, q in zip(tr.agent.actor.parameters(), before))
    d_tg = sum(float((p - q).abs().sum()) for p, q in zip(tr.agent.actor_target.parameters(), before))
    assert 0 < d_tg < d_on
    assert tr.ro.env.row_overflow_envs() == 0


def test_graphed_updates_equal_eager_updates():
    """td3.GraphedUpdates: the update replayed from hipGraphs (HIP target kernels + autograd + Adam captured together) moves
    the parameters like the eager update from the same start, on two morphologies, with and without the actor step."""
    import copy
    import torch
    from oracle.formula import synth_obs
    from sgrl_amd import graph as G, mjcf
    from sgrl_amd.td3 import Agent, GraphedUpdates, default_train_args
    args = default_train_args(batch_size=12)
    torch.manual_seed(2)
    eager = Agent(args, device="cuda:0")
    graphed = Agent(args, device="cuda:0")
    graphed.load_state_dict(copy.deepcopy(eager.state_dict()))
    gu = GraphedUpdates(graphed, 12)
    for opt in (eager.actor_optimizer, eager.critic_optimizer):
        for g in opt.param_groups:
            g["capturable"] = True            # same Adam arithmetic on both sides
    eager.models2train(); graphed.models2train()
    morphs = []
    for name in ("3d_walker_5_foot", "3d_hopper_3_shin"):
        m = mjcf.load_asset(name)
        gd = G.getGraphDict(m.parents, ["pre", "inlcrs", "postlcrs"], [], device=torch.device("cuda:0"))
        L = m.num_limbs
        rng = np.random.RandomState(L)
        batch = {"obs": torch.from_numpy(synth_obs(L, 12, 5).astype(np.float32)).cuda(),
                 "next_obs": torch.from_numpy(synth_obs(L, 12, 6).astype(np.float32)).cuda(),
                 "action": torch.from_numpy(rng.uniform(-1, 1, size=(12, 3 * L)).astype(np.float32)).cuda(),
                 "reward": torch.from_numpy(rng.normal(1, 0.5, size=(12, 1)).astype(np.float32)).cuda(),
                 "done": torch.zeros(12, 1).cuda()}
        morphs.append((name, gd, L, batch))
    start = [p.detach().clone() for p in eager.critic.parameters()]
    # identical noise on both sides: policy_noise = 0 makes the draw irrelevant
    args.policy_noise = 0.0
    for key, (name, gd, L, batch) in enumerate(morphs):          # warm-up = 2 real updates per morphology, mirrored eagerly
        gu.warm(key, gd, L, batch, iters=2)
        eager.change_morphology(gd)
        for it in range(2):
            eager.update(batch, it, noise=torch.zeros(12, 3 * L, device="cuda"))
    for it in range(4):                                            # capture at first use, then replays
        for key, (name, gd, L, batch) in enumerate(morphs):
            out = gu.update(key, gd, L, batch, it)
            eager.change_morphology(gd)
            ref = eager.update(batch, it, noise=torch.zeros(12, 3 * L, device="cuda"))
            assert abs(float(out["loss/critic_loss"]) - float(ref["loss/critic_loss"])) < 2e-3 * abs(float(ref["loss/critic_loss"])) + 1e-5, (it, name)
            assert ("loss/actor_loss" in out) == ("loss/actor_loss" in ref)
    assert set(gu.slots[0]["graphs"]) == {0, 1}
    for nm in ("actor", "critic", "actor_target", "critic_target"):
        for p, q, s0 in zip(getattr(graphed, nm).parameters(), getattr(eager, nm).parameters(),
                            start if nm == "critic" else getattr(eager, nm).parameters()):
            assert float((p - q).abs().max()) < 2e-4 * (1 + float(q.abs().max())), nm
    moved = max(float((p - s0).abs().max()) for p, s0 in zip(graphed.critic.parameters(), start))
    assert moved > 1e-4


def test_graphed_update_survives_a_workspace_regrowth():
    """A morphology captured BEFORE a larger one makes the SET handles regrow their workspaces must be captured again, not
    replayed into freed memory (td3.GraphedUpdates workspace stamp)."""
    from oracle.formula import synth_obs
    from sgrl_amd import graph as G, mjcf
    from sgrl_amd.rollout import TRAV
    from sgrl_amd.td3 import Agent, GraphedUpdates, default_train_args
    import torch
    dev = torch.device("cuda:0")
    torch.manual_seed(0)
    agent = Agent(default_train_args(), device=dev)
    agent.models2train()
    gr = GraphedUpdates(agent, 100)
    data = {}
    for k, name in enumerate(["3d_hopper_3_shin", "3d_cheetah_14_full"]):
        m = mjcf.load_asset(name)
        L = m.num_limbs
        gd = G.getGraphDict(m.parents, TRAV, [], device=dev)
        batch = {"obs": torch.from_numpy(synth_obs(L, 100, 1).astype(np.float32)).to(dev), "next_obs": torch.from_numpy(synth_obs(L, 100, 2).astype(np.float32)).to(dev),
                 "action": torch.rand(100, 3 * L, device=dev) * 2 - 1, "reward": torch.randn(100, 1, device=dev), "done": torch.zeros(100, 1, device=dev)}
        data[k] = (gd, L, batch)
        gr.warm(k, gd, L, batch, iters=1)
        for it in range(2):
            gr.update(k, gd, L, batch, it)          # captured right after ITS warm-up: the second morphology regrows the workspaces
    stamp_small = dict(gr.slots[0]["stamp"])
    gd, L, batch = data[0]
    for it in range(4):
        out = gr.update(0, gd, L, batch, it)        # must re-capture, not fault
    torch.cuda.synchronize()
    assert gr.slots[0]["stamp"][0] != stamp_small[0]
    assert np.isfinite(float(out["loss/critic_loss"]))


def test_fused_ingest_writes_exactly_what_the_row_by_row_path_writes():
    """TransitionSink.ingest on the GPU: one launch per gathered block (include/sgrl.h sgrl_ingest_rows, slots from a prefix sum of
    the store flags) against the morphology-by-morphology indexed copies of the CPU path -- every ring buffer bit for bit, the
    write pointers and fill levels, over several blocks with random store flags, including a ring that wraps and a block that
    stores nothing; a block that would wrap a ring onto itself falls back to the row-by-row path."""
    import torch
    from sgrl_amd import rollout
    from sgrl_amd.replay import DeviceReplayBuffer
    dev = torch.device("cuda:0")
    limbs = [2, 7, 4, 5]
    per = [40, 30, 50, 20]
    env_morph = sum(([k] * n for k, n in enumerate(per)), [])
    n, omax, amax = len(env_morph), 41 * 7, 3 * 7
    caps = [1000, 64, 1000, 25]                       # ring 1 wraps after a few blocks, ring 3 is smaller than one full block of its rows... almost
    def make():
        bufs = [DeviceReplayBuffer(41 * L, 3 * L, max_buffer_size=c, device=dev) for L, c in zip(limbs, caps)]
        return rollout.TransitionSink(env_morph, limbs, omax, amax, device=dev, buffers=bufs), bufs
    g = torch.Generator().manual_seed(4)
    blocks = []
    for t in range(9):
        row = torch.randn(n, 2 * omax + amax + 4, generator=g)
        store = (torch.rand(n, generator=g) < (0.0 if t == 3 else 0.7)).float()
        if t == 6:
            store[:] = 1.0                            # 20 rows for ring 3 (capacity 25): fits; ring 1 gets 30 rows of 64
        row[:, 2 * omax + amax + 2] = store
        row[:, 2 * omax + amax + 3] = torch.tensor(env_morph, dtype=torch.float32)
        blocks.append(row.to(dev))
    out = {}
    for fused in (False, True):
        rollout.FUSED_INGEST = fused
        sink, bufs = make()
        for b in blocks:
            sink.ingest([b])
        torch.cuda.synchronize()
        out[fused] = ([(x.obs_buffer.clone(), x.action_buffer.clone(), x.next_obs_buffer.clone(), x.reward_buffer.clone(), x.done_buffer.clone(),
                        x.curr, x.max_sample_size) for x in bufs], sink.stored)
    rollout.FUSED_INGEST = True
    assert out[True][1] == out[False][1] > 0
    for a, b in zip(out[False][0], out[True][0]):
        assert a[5:] == b[5:]
        for ta, tb in zip(a[:5], b[:5]):
            assert torch.equal(ta, tb)
    # more stored rows of one morphology than its ring holds, in ONE block: the fused path declines, the result is still the reference's
    tiny = [DeviceReplayBuffer(41 * L, 3 * L, max_buffer_size=8, device=dev) for L in limbs]
    ref = [DeviceReplayBuffer(41 * L, 3 * L, max_buffer_size=8, device=dev) for L in limbs]
    s1 = rollout.TransitionSink(env_morph, limbs, omax, amax, device=dev, buffers=tiny)
    rollout.FUSED_INGEST = False
    s0 = rollout.TransitionSink(env_morph, limbs, omax, amax, device=dev, buffers=ref)
    s0.ingest([blocks[6]])
    rollout.FUSED_INGEST = True
    s1.ingest([blocks[6]])
    for x, y in zip(tiny, ref):
        assert torch.equal(x.obs_buffer, y.obs_buffer) and (x.curr, x.max_sample_size) == (y.curr, y.max_sample_size)


def test_whole_gather_ingest_equals_block_by_block():
    """The learner at N > 1: the gather arrives in ONE contiguous [N * rows, row] tensor whose per-rank blocks are
    views (rollout.ReplayGather recv_flat); TransitionSink.ingest recognises them and runs sgrl_ingest_block ONCE over the whole
    tensor (k_ingest_keys + k_ingest_slots + k_ingest_rows, three launches whatever N).  Rings, write pointers and fill levels must equal what the
    block-by-block ingest of the same views writes, over several steps, with random store flags, a ring that wraps, and more
    rows than one 256-row chunk per morphology (ranks of rows span chunks)."""
    import torch
    from sgrl_amd import rollout
    from sgrl_amd.replay import DeviceReplayBuffer
    dev = torch.device("cuda:0")
    limbs = [2, 7, 4, 5]
    per = [300, 130, 350, 20]
    env_morph = sum(([k] * n for k, n in enumerate(per)), [])
    n, omax, amax, world = len(env_morph), 41 * 7, 3 * 7, 3
    caps = [100000, 2500, 100000, 4000]                 # ring 1 wraps after a few steps; every ring holds one whole gather
    g = torch.Generator().manual_seed(9)
    steps = []
    for t in range(6):
        flat = torch.randn(world * n, 2 * omax + amax + 4, generator=g)
        flat[:, 2 * omax + amax + 2] = (torch.rand(world * n, generator=g) < (0.0 if t == 2 else 0.8)).float()
        flat[:, 2 * omax + amax + 3] = torch.tensor(env_morph * world, dtype=torch.float32)
        steps.append(flat.to(dev))
    out = {}
    for whole in (False, True):
        bufs = [DeviceReplayBuffer(41 * L, 3 * L, max_buffer_size=c, device=dev) for L, c in zip(limbs, caps)]
        sink = rollout.TransitionSink(env_morph, limbs, omax, amax, device=dev, buffers=bufs)
        for flat in steps:
            views = list(flat.split(n, dim=0))
            if whole:
                sink.ingest(views)                      # views of one tensor, rank order: one call over the whole gather
            else:
                for v in views:
                    sink.ingest([v.clone()])            # clones: no common base, block by block
        torch.cuda.synchronize()
        out[whole] = ([(x.obs_buffer.clone(), x.action_buffer.clone(), x.next_obs_buffer.clone(), x.reward_buffer.clone(), x.done_buffer.clone(),
                        x.curr, x.max_sample_size) for x in bufs], sink.stored)
    assert out[True][1] == out[False][1] > 0
    for a, b in zip(out[False][0], out[True][0]):
        assert a[5:] == b[5:]
        for ta, tb in zip(a[:5], b[:5]):
            assert torch.equal(ta, tb)


def test_lagged_round_flag_stores_the_same_rows():
    """DeviceTrainer's default reads the round-finished flag one step late (rollout.TransitionSink lag_flag: no host synchronisation
    per collection step).  The late step's rows carry store = False everywhere, so warm-up rounds driven by the same seeds fill the
    replay rings with exactly the rows of the immediate flag, end every round one step later, and report the same statistics."""
    torch, a = _trainer(max_episode_steps=25, seed=9, lag_flag=False)
    _, b = _trainer(max_episode_steps=25, seed=9, lag_flag=True)
    assert b.sink.lag_flag and not a.sink.lag_flag
    ends = {"a": [], "b": []}
    stats = {"a": [], "b": []}
    for key, tr in (("a", a), ("b", b)):
        for t in range(1, 28 if key == "a" else 29):      # exactly one round: every environment is done by step 25 (time limit)
            if tr.collect_step(random_actions=True):
                ends[key].append(t)
                stats[key].append((tr.sink.collector.episode_timesteps.clone(), tr.sink.collector.episode_reward.clone()))
                break
    assert len(ends["a"]) == 1 and ends["b"] == [ends["a"][0] + 1]
    assert torch.equal(stats["a"][0][0], stats["b"][0][0]) and torch.equal(stats["a"][0][1], stats["b"][0][1])
    for ba, bb in zip(a.buffers, b.buffers):
        assert ba.max_sample_size == bb.max_sample_size and ba.curr == bb.curr and ba.max_sample_size > 0
        n = ba.max_sample_size
        for name in ("obs_buffer", "action_buffer", "next_obs_buffer", "reward_buffer", "done_buffer"):
            assert torch.equal(getattr(ba, name)[:n], getattr(bb, name)[:n]), name


def test_fused_round_record_equals_the_tensor_form():
    """include/sgrl.h sgrl_round_record (one launch) against RoundCollector's tensor operations (the statement of the reference's rule,
    trainer.py:205-232) on random reward / done sequences incl. the time limit: store masks, stored `done`, episode statistics and the
    round-finished flag agree step by step, exactly."""
    import torch
    from sgrl_amd import rollout
    n, T = 1000, 12
    g = torch.Generator(device="cuda").manual_seed(4)
    a = rollout.RoundCollector(n, max_episode_steps=T, device="cuda:0")
    b = rollout.RoundCollector(n, max_episode_steps=T, device="cuda:0")
    fin_a = fin_b = False
    for t in range(T + 3):
        rew = torch.randn(n, device="cuda", generator=g)
        done = (torch.rand(n, device="cuda", generator=g) < 0.08).to(torch.uint8)
        rollout.FUSED_RECORD = True
        sa, da, fin_a = a.record(rew, done)
        sa, da = sa.clone(), da.clone()
        rollout.FUSED_RECORD = False
        try:
            sb, db, fin_b = b.record(rew, done)
        finally:
            rollout.FUSED_RECORD = True
        assert torch.equal(sa, sb) and torch.equal(da, db) and fin_a == fin_b, t
        assert torch.equal(a.done_list, b.done_list) and torch.equal(a.episode_timesteps, b.episode_timesteps)
        assert torch.equal(a.episode_reward, b.episode_reward) and torch.equal(a._reward_buf, b._reward_buf)
    assert fin_a and a.per_morph_iter() == b.per_morph_iter()
