"""TD3 agent behind the reference's `Agent` surface (reference src/agent.py:20-215).

Same constructor argument (`args` namespace), attributes (`actor`, `actor_target`, `critic`, `critic_target`, the two
Adam optimizers), methods (`update`, `try_update_target_network`, `select_action`, `change_morphology`, `models2eval`,
`models2train`) and `state_dict()` keys as the reference, so a `save.pth` written by either side loads on the other
(sgrl_amd/snapshot.py).  The update is ordinary PyTorch-ROCm autograd through the differentiable path of
`SEPolicy` / `SECritic` (set_policy.py); everything evaluated under `torch.no_grad()` on the GPU -- the target action
and the twin target Q values (agent.py:126-148), `select_action` (agent.py:189-198) -- runs on the HIP kernels
(csrc/set_actor.hip), which read the parameters' live storage on every call and therefore follow the in-place Polyak
updates (`target_param.data.copy_`, common/functional.py:7-10).

Pinned by tests/golden/td3_update.npz, produced by executing the reference's own `Agent.update`
(tools/capture_golden_update.py)."""
import os
import types

import contextlib
import gc

import torch
import torch.nn as nn
import torch.nn.functional as F

from . import train_ops
from .set_policy import SECritic, SEPolicy, default_args


def soft_update_network(source_network, target_network, tau):
    """target <- tau * source + (1 - tau) * target for every parameter (reference common/functional.py:7-10), written through
    `.data` like the reference does.  On the GPU the ~300 tensors of a network pair are updated by two multi-tensor launches
    instead of four small kernels per tensor (the per-tensor loop was 40 % of all launches of a TD3 update); per element the
    same two products and one sum (the second product inside the addition)."""
    with torch.no_grad():
        targets = [p.data for p in target_network.parameters()]
        sources = [p.data for p in source_network.parameters()]
        if _TABLE_OPT and targets and all(t.is_cuda and t.dtype == torch.float32 and t.is_contiguous() and s.is_contiguous() and
                                          s.dtype == torch.float32 and s.device == t.device and s.numel() == t.numel()
                                          for t, s in zip(targets, sources)):
            import ctypes
            from . import train_ops
            dev = targets[0].device
            ent = _table("lerp", [(t.data_ptr(), s.data_ptr(), 0, 0, 0, t.numel()) for t, s in zip(targets, sources)], dev)
            if ent is not None:
                L = _optim_lib()
                train_ops._check(L, L.sgrl_optim_lerp(ctypes.c_void_p(ent["dev_tab"].data_ptr()), ctypes.c_void_p(ent["dev_chunks"].data_ptr()),
                                                      ent["n_chunks"], float(tau), ctypes.c_void_p(torch.cuda.current_stream(dev).cuda_stream)),
                                 "sgrl_optim_lerp")
                _touched(list(target_network.parameters()))
                return
        if targets and targets[0].is_cuda:
            torch._foreach_mul_(targets, 1 - tau)
            torch._foreach_add_(targets, sources, alpha=tau)
        else:
            for t, l in zip(targets, sources):
                t.copy_(tau * l + (1 - tau) * t)


# False: the foreach chain (bit-compatible with torch.optim.Adam's multi-tensor path) instead of the fused kernel
_FUSED_ADAM = hasattr(torch, "_fused_adam_")
# Gradient clipping + Adam, and the soft target update, over a device TABLE of tensor addresses (csrc/train_gemm.hip k_opt_*,
# include/sgrl_train.h sgrl_optim_*): three launches and one instead of torch's ~50 multi-tensor launches of ~18 us per update.
_TABLE_OPT = True
_tables = {}          # key (kind, address tuple) -> dict(dev table, dev chunks, pinned copies, scratch): see _table()
_capture_owner = None  # whoever is capturing a hipGraph right now (GraphedUpdates: (id, morphology key, flag)); see release_tables()


def _touched(params):
    """A kernel of this library wrote these tensors through raw pointers: bump their version counters, as an in-place PyTorch
    operation would have, so that whoever fingerprints parameters (rollout.Rollout's held weight pack, autograd's saved-tensor
    checks) sees the write.  Host-side only: 12 us for 300 tensors."""
    torch.autograd.graph.increment_version(params)


def _optim_lib():
    import ctypes
    from . import train_ops
    L = train_ops._L()
    if not getattr(L, "_sgrl_optim_bound", False):
        vp, ci, cf, cd = ctypes.c_void_p, ctypes.c_int, ctypes.c_float, ctypes.c_double
        L.sgrl_optim_chunk.restype = ci
        L.sgrl_optim_clip_adam.argtypes = [vp, ci, vp, ci, cd, cd, cd, cd, cf, vp, vp]
        L.sgrl_optim_lerp.argtypes = [vp, vp, ci, cf, vp]
        L._sgrl_optim_bound = True
    return L


_spare_pinned = {}    # (dtype, shape) -> pinned host tensors allocated OUTSIDE captures (hipHostMalloc is not permitted inside one)


def _stock_pinned(n=4):
    """Before a capture: every table size seen so far has at least n spare pinned buffers."""
    if torch.cuda.is_current_stream_capturing():
        return
    for (dt, shape), pool in _spare_pinned.items():
        while len(pool) < n:
            pool.append(torch.empty(shape, dtype=dt).pin_memory())


def _pinned_copy(t):
    """A pinned copy of the small CPU tensor t; during a capture from the stock (None when it is empty: the caller falls back)."""
    pool = _spare_pinned.setdefault((t.dtype, tuple(t.shape)), [])
    if torch.cuda.is_current_stream_capturing():
        if not pool:
            return None
        buf = pool.pop()
    else:
        buf = torch.empty(t.shape, dtype=t.dtype).pin_memory()
        _stock_pinned(4)
    buf.copy_(t)
    return buf


def _table(kind, rows, device):
    """Device copy of a tensor table (rows of six int64: five addresses and the element count) and its chunk list, cached by
    content.  The upload is a copy from PINNED host memory on the current stream: inside a hipGraph capture it becomes a memcpy
    node of the graph, so entries made during a capture are kept (with their host buffers) for as long as the process lives."""
    key = (kind, device.index, tuple(rows))
    ent = _tables.get(key)
    capturing = torch.cuda.is_current_stream_capturing()
    if ent is not None:
        if capturing:
            ent["captured"] = True                 # a graph now points at this entry's device buffers: kept while an owner lives
            ent["owners"].add(_capture_owner)
        elif not ent["eager_valid"]:               # made during a capture (its upload exists only as a node of that graph)
            ent["dev_tab"].copy_(ent["host_tab"], non_blocking=True)
            ent["dev_chunks"].copy_(ent["host_chunks"], non_blocking=True)
            ent["eager_valid"] = True
        return ent
    chunk = int(_optim_lib().sgrl_optim_chunk())
    chunks = [(i, o) for i, r in enumerate(rows) for o in range(0, r[5], chunk)]
    host_tab = _pinned_copy(torch.tensor(rows, dtype=torch.int64))
    host_chunks = _pinned_copy(torch.tensor(chunks, dtype=torch.int32))
    if host_tab is None or host_chunks is None:
        return None                                  # capturing with no pinned buffer in stock: the caller takes torch's path
    ent = {"host_tab": host_tab, "host_chunks": host_chunks, "n": len(rows), "n_chunks": len(chunks), "captured": capturing,
           "dev_tab": torch.empty_like(host_tab, device=device), "dev_chunks": torch.empty_like(host_chunks, device=device),
           "scratch": torch.zeros(1 + len(chunks), dtype=torch.float32, device=device), "eager_valid": not capturing,
           "owners": {_capture_owner} if capturing else set()}
    ent["dev_tab"].copy_(host_tab, non_blocking=True)
    ent["dev_chunks"].copy_(host_chunks, non_blocking=True)
    if not capturing and len(_tables) > 64:          # eager callers whose gradient addresses keep changing: bounded cache
        for k in [k for k, v in _tables.items() if not v["captured"]][:32]:
            del _tables[k]
    _tables[key] = ent
    return ent


def release_tables(owner):
    """The graph(s) captured under `owner` are gone (GraphedUpdates dropped them to capture again): table entries that only they
    pointed at become ordinary cache entries again -- evictable -- instead of living as long as the process."""
    for ent in _tables.values():
        if owner in ent["owners"]:
            ent["owners"].discard(owner)
            if not ent["owners"]:
                ent["captured"] = False


def clip_and_step(opt, max_norm):
    """torch.nn.utils.clip_grad_norm_(params, max_norm) (max_norm > 0) followed by opt.step() for a capturable torch.optim.Adam
    (reference agent.py:161-164, 174-177) on the optimizer's OWN state tensors -- as three launches over a table of the tensors'
    addresses where everything is float32 on one GPU, through clip_grad_norm_ + adam_step otherwise."""
    params = [p for g in opt.param_groups for p in g["params"] if p.grad is not None]
    ok = _TABLE_OPT and len(opt.param_groups) == 1 and params and all(
        p.is_cuda and p.dtype == torch.float32 and p.is_contiguous() and p.grad.is_contiguous() and p.grad.dtype == torch.float32
        and p.grad.device == p.device for p in params)
    group = opt.param_groups[0]
    ok = ok and group.get("capturable", False) and group.get("weight_decay", 0) == 0 and not group.get("amsgrad", False) and \
        not group.get("maximize", False)
    if not ok:
        if max_norm and max_norm > 0:
            torch.nn.utils.clip_grad_norm_([p for g in opt.param_groups for p in g["params"]], max_norm)
        return adam_step(opt)
    host = opt.__dict__.setdefault("_sgrl_step_class", {})
    rows = []
    for p in params:
        st = opt.state[p]
        if len(st) == 0:            # torch.optim.Adam's lazy state initialisation (capturable: the counter lives on the device)
            st["step"] = torch.zeros((), dtype=torch.float32, device=p.device)
            st["exp_avg"] = torch.zeros_like(p, memory_format=torch.preserve_format)
            st["exp_avg_sq"] = torch.zeros_like(p, memory_format=torch.preserve_format)
            host[id(p)] = 0
        elif id(p) not in host:
            host[id(p)] = int(st["step"].item())
        host[id(p)] += 1
        rows.append((p.data_ptr(), p.grad.data_ptr(), st["exp_avg"].data_ptr(), st["exp_avg_sq"].data_ptr(), st["step"].data_ptr(), p.numel()))
    dev = params[0].device
    ent = _table("adam", rows, dev)
    if ent is None:
        for p in params:
            host[id(p)] -= 1                        # adam_step counts the step itself
        if max_norm and max_norm > 0:
            torch.nn.utils.clip_grad_norm_([p for g in opt.param_groups for p in g["params"]], max_norm)
        return adam_step(opt)
    import ctypes
    from . import train_ops
    L = _optim_lib()
    beta1, beta2 = group["betas"]
    lr = group["lr"]
    rc = L.sgrl_optim_clip_adam(ctypes.c_void_p(ent["dev_tab"].data_ptr()), ent["n"], ctypes.c_void_p(ent["dev_chunks"].data_ptr()),
                                ent["n_chunks"], float(lr), float(beta1), float(beta2), float(group["eps"]),
                                float(max_norm) if (max_norm and max_norm > 0) else 0.0, ctypes.c_void_p(ent["scratch"].data_ptr()),
                                ctypes.c_void_p(torch.cuda.current_stream(dev).cuda_stream))
    train_ops._check(L, rc, "sgrl_optim_clip_adam")
    _touched(params)


def adam_step(opt):
    """`opt.step()` for a torch.optim.Adam whose groups are `capturable` (step counters on the device: GraphedUpdates), on the
    optimizer's OWN state tensors (state_dict unchanged).  torch's capturable multi-tensor path divides every parameter's
    tensors by per-parameter 0-dim bias-correction tensors, which its foreach kernels cannot batch: ~3 small launches per
    parameter, ~900 per TD3 update, a third of the update's launches.  Parameters that have always been stepped together share
    one step count, so here the two bias corrections are computed ONCE per such class (from its first counter; the classes
    are kept on the host: a parameter that skipped a step -- no gradient -- forms its own) and applied as single scalar
    tensors: a dozen multi-tensor launches per optimizer.  Same formula, term for term, as torch.optim.Adam (no weight decay,
    no amsgrad, not maximising -- the reference's optimizers, agent.py:96-115)."""
    for group in opt.param_groups:
        if not group.get("capturable", False) or group.get("weight_decay", 0) != 0 or group.get("amsgrad", False) or \
                group.get("maximize", False):
            return opt.step()
    host = opt.__dict__.setdefault("_sgrl_step_class", {})       # id(param) -> number of steps taken, as far as the host knows
    for group in opt.param_groups:
        params = [p for p in group["params"] if p.grad is not None]
        if not params:
            continue
        beta1, beta2 = group["betas"]
        classes = {}
        for p in params:
            st = opt.state[p]
            if len(st) == 0:        # torch.optim.Adam's lazy state initialisation (capturable: the counter lives on the device)
                st["step"] = torch.zeros((), dtype=torch.float32, device=p.device)
                st["exp_avg"] = torch.zeros_like(p, memory_format=torch.preserve_format)
                st["exp_avg_sq"] = torch.zeros_like(p, memory_format=torch.preserve_format)
                host[id(p)] = 0
            elif id(p) not in host:  # state loaded from a checkpoint: read the counter once (never during a capture)
                host[id(p)] = int(st["step"].item())
            classes.setdefault(host[id(p)], []).append(p)
            host[id(p)] += 1
        with torch.no_grad():
            for members in classes.values():
                grads = [p.grad for p in members]
                steps = [opt.state[p]["step"] for p in members]
                exp_avgs = [opt.state[p]["exp_avg"] for p in members]
                exp_avg_sqs = [opt.state[p]["exp_avg_sq"] for p in members]
                torch._foreach_add_(steps, 1)
                if _FUSED_ADAM and members[0].is_cuda:
                    # the whole update of the class in one multi-tensor kernel (torch's own `fused=True` Adam kernel: per-
                    # parameter step counters are read on the device, so nothing here divides by 0-dim tensors): three launches
                    # instead of a dozen per class.  Same formula; its float32 operation order differs from the foreach chain
                    # below in the last bit (tests/test_td3_update.py holds both to the reference's Agent.update).
                    torch._fused_adam_([p.data for p in members], grads, exp_avgs, exp_avg_sqs, [], steps, amsgrad=False,
                                       lr=group["lr"], beta1=beta1, beta2=beta2, weight_decay=0.0, eps=group["eps"],
                                       maximize=False, grad_scale=None, found_inf=None)
                    continue
                torch._foreach_lerp_(exp_avgs, grads, 1 - beta1)
                torch._foreach_mul_(exp_avg_sqs, beta2)
                torch._foreach_addcmul_(exp_avg_sqs, grads, grads, 1 - beta2)
                step = steps[0]
                neg_step_size = (group["lr"] / (1 - beta1 ** step)).neg()
                bias_correction2_sqrt = (1 - beta2 ** step).sqrt()
                denom = torch._foreach_sqrt(exp_avg_sqs)
                torch._foreach_div_(denom, bias_correction2_sqrt)
                torch._foreach_add_(denom, group["eps"])
                torch._foreach_div_(denom, neg_step_size)
                torch._foreach_addcdiv_([p.data for p in members], exp_avgs, denom)


def default_train_args(**over):
    """The reference's TD3 / SET hyper-parameters (reference arguments.py:55-160, configs/default.py:9-12).

    Two batch sizes, as in the reference: `agent_batch_size` = 256 is what an update SAMPLES (configs/default.py:61 "trainer":
    {"agent_batch_size": 256}, passed as **args['trainer'] at main.py:164-171, read at trainer.py:289-291
    `self.env_buffer[name].sample(self.agent_batch_size)`); `batch_size` = 100 (arguments.py:66-68 --batch_size) only reaches the
    policies' constructor argument, which none of them uses (agent.py:42)."""
    a = default_args()
    a.agent_batch_size = 256
    a.actor_type = a.critic_type = "set"
    a.limb_obs_size, a.limb_action_size = 41, 3
    a.msg_dim, a.batch_size, a.max_action, a.max_children = 32, 100, 1.0, 3
    a.disable_fold, a.td, a.bu = True, False, False
    a.lr, a.discount, a.policy_noise, a.noise_clip, a.policy_freq = 1e-4, 0.99, 0.2, 0.5, 2
    a.expl_noise, a.grad_clipping_value, a.max_episode_steps = 0.126, 0.1, 1000
    a.agent = types.SimpleNamespace(target_smoothing_tau=0.005, reward_scale=1.0)
    for k, v in over.items():
        setattr(a, k, v)
    return a


class Agent(nn.Module):
    def __init__(self, args, device=None, use_hip=True):
        super().__init__()
        self.args = args
        atype, ctype = getattr(args, "actor_type", "set"), getattr(args, "critic_type", "set")
        if atype not in ("set", "swat", "smp") or ctype not in ("set", "swat", "smp"):
            raise NotImplementedError("actor / critic types 'set' (HIP fast path), 'swat' and 'smp' (PyTorch) are built; "
                                      "'mlp' is not (SURVEY 8 f4)")
        self.networks = {}
        from .smp_policy import ActorGraphPolicy, CriticGraphPolicy
        from .swat_policy import CriticStructurePolicy, StructurePolicy

        def actor():
            if atype == "swat":
                return StructurePolicy(args.limb_obs_size, args.limb_action_size, args.msg_dim, args.batch_size, args.max_action,
                                       args.max_children, args.disable_fold, args.td, args.bu, args, device=device)
            if atype == "smp":
                return ActorGraphPolicy(args.limb_obs_size, args.limb_action_size, args.msg_dim, args.batch_size, args.max_action,
                                        args.max_children, args.disable_fold, args.td, args.bu, args, device=device)
            return SEPolicy(args.limb_obs_size, args.limb_action_size, args.msg_dim, args.batch_size, args.max_action,
                            args.max_children, args.disable_fold, args.td, args.bu, args, device=device, use_hip=use_hip)

        def critic():
            if ctype == "swat":
                return CriticStructurePolicy(args.limb_obs_size, args.limb_action_size, args.msg_dim, args.batch_size,
                                             args.max_children, args.disable_fold, args.td, args.bu, args, device=device)
            if ctype == "smp":
                return CriticGraphPolicy(args.limb_obs_size, args.limb_action_size, args.msg_dim, args.batch_size,
                                         args.max_children, args.disable_fold, args.td, args.bu, args, device=device)
            return SECritic(args.limb_obs_size, args.limb_action_size, args.msg_dim, args.batch_size, args.max_children,
                            args.disable_fold, args.td, args.bu, args, device=device, use_hip=use_hip)
        self.actor, self.actor_target = actor(), actor()
        self.critic, self.critic_target = critic(), critic()
        soft_update_network(self.actor, self.actor_target, 1.0)
        soft_update_network(self.critic, self.critic_target, 1.0)
        self.actor_optimizer = torch.optim.Adam(self.actor.parameters(), lr=args.lr)
        self.critic_optimizer = torch.optim.Adam(self.critic.parameters(), lr=args.lr)
        self.models2eval()
        self.tot_update_count = 0
        self.target_smoothing_tau = args.agent.target_smoothing_tau
        self.reward_scale = args.agent.reward_scale

    @property
    def device(self):
        return next(self.actor.parameters()).device

    def update(self, data_batch, it, noise=None, lazy_stats=False, skip_unused_critic_grads=False):
        """One TD3 step on a batch of ONE morphology (reference agent.py:117-183).  `noise` replaces the N(0, policy_noise)
        draw of agent.py:128 (tests: so that a run can be compared with the reference number for number; graph capture: a
        static buffer refilled before every replay).  lazy_stats: keep the two reward statistics as device tensors instead
        of `.item()` floats (no host sync -- required inside a hipGraph capture)."""
        reward_batch, target_Q = self.update_targets(data_batch, noise)
        current_Q1, current_Q2 = self.update_critic_forward(data_batch)
        return self.update_finish(data_batch, it, reward_batch, target_Q, current_Q1, current_Q2, lazy_stats, skip_unused_critic_grads)

    # The three parts of an update.  The first two -- the no-grad target chain (actor_target -> critic_target) and the critics'
    # forward with autograd -- need nothing from each other: GraphedUpdates records them as separate graphs and replays them
    # side by side on two streams (both are chains of small latency-bound launches that leave most of the chip idle).
    def update_targets(self, data_batch, noise=None):
        args = self.args
        action_batch, next_obs_batch = data_batch["action"], data_batch["next_obs"]
        reward_batch, done_batch = data_batch["reward"] * self.reward_scale, data_batch["done"]
        with torch.no_grad():
            if noise is None:
                noise = torch.zeros_like(action_batch).normal_(0, args.policy_noise)
            noise = noise.clamp(-args.noise_clip, args.noise_clip)
            next_action = (self.actor_target(next_obs_batch) + noise).clamp(-args.max_action, args.max_action)
            target_Q1, target_Q2 = self.critic_target(next_obs_batch, next_action)    # per-limb values [B, L]
            target_Q = torch.min(target_Q1, target_Q2)
            target_Q = reward_batch + ((1.0 - done_batch) * args.discount * target_Q)  # reward [B, 1] broadcast over limbs
        return reward_batch, target_Q

    def update_critic_forward(self, data_batch):
        return self.critic(data_batch["obs"], data_batch["action"])

    def update_finish(self, data_batch, it, reward_batch, target_Q, current_Q1, current_Q2, lazy_stats=False,
                      skip_unused_critic_grads=False):
        args = self.args
        obs_batch = data_batch["obs"]
        critic_loss = F.mse_loss(current_Q1, target_Q) + F.mse_loss(current_Q2, target_Q)
        self.critic_optimizer.zero_grad()
        # (graphed path) the weight gradients are not on the backward pass's critical path: collected, issued together at the end
        with train_ops.deferred_wgrads(enabled=skip_unused_critic_grads):
            critic_loss.backward()
        clip_and_step(self.critic_optimizer, args.grad_clipping_value)
        rmean, rvar = torch.mean(reward_batch), torch.var(reward_batch)
        # the losses are returned DETACHED: a loss that keeps its autograd graph alive also keeps the parameters' AccumulateGrad
        # nodes -- and the stream they were created on -- alive into the next update; a hipGraph captured on another stream then
        # records a cross-stream hand-over per parameter (a 2 500-node update graph cost 39 ms to launch instead of 19)
        loss_dict = {"loss/critic_loss": critic_loss.detach(), "misc/train_reward_mean": rmean if lazy_stats else rmean.item(),
                     "misc/train_reward_var": rvar if lazy_stats else rvar.item()}
        if it % args.policy_freq == 0:       # delayed policy update
            # The actor loss back-propagates THROUGH the critic, but the critic's own parameter gradients from this pass are
            # never used (reference agent.py:167-176: only actor_optimizer steps; critic_optimizer.zero_grad() discards them
            # at the top of the next update).  skip_unused_critic_grads: do not compute them -- a third of the pass's
            # weight-gradient products; parameters, losses and actions are unchanged, only the stale `critic.*.grad` left behind
            # differs from the reference's (default off: the reference's exact state; GraphedUpdates turns it on).
            critic_params = [p for p in self.critic.parameters() if p.requires_grad] if skip_unused_critic_grads else []
            for p in critic_params:
                p.requires_grad_(False)
            try:
                actor_loss = -self.critic.Q1(obs_batch, self.actor(obs_batch)).mean()
                self.actor_optimizer.zero_grad()
                with train_ops.deferred_wgrads(enabled=skip_unused_critic_grads):
                    actor_loss.backward()
            finally:
                for p in critic_params:
                    p.requires_grad_(True)
            clip_and_step(self.actor_optimizer, args.grad_clipping_value)
            self.try_update_target_network()
            loss_dict.update({"loss/actor_loss": actor_loss.detach()})
        return loss_dict

    def try_update_target_network(self):
        soft_update_network(self.critic, self.critic_target, self.target_smoothing_tau)
        soft_update_network(self.actor, self.actor_target, self.target_smoothing_tau)

    @torch.no_grad()
    def select_action(self, obs, deterministic=False):
        if len(obs.shape) == 1:
            obs = obs[None, ]
        if not isinstance(obs, torch.Tensor):
            obs = torch.as_tensor(obs, dtype=torch.float32).to(self.device)
        return self.actor(obs).cpu().numpy()

    def change_morphology(self, graph):
        self.actor.change_morphology(graph)
        self.actor_target.change_morphology(graph)
        self.critic.change_morphology(graph)
        self.critic_target.change_morphology(graph)

    def models2eval(self):
        for m in (self.actor, self.actor_target, self.critic, self.critic_target):
            m.eval()

    def models2train(self):
        for m in (self.actor, self.actor_target, self.critic, self.critic_target):
            m.train()


def _concurrent_stream(cur, tries=8):
    """A stream that REALLY runs beside `cur`: HIP multiplexes its streams onto a handful of hardware queues and two streams on
    one queue execute strictly one after the other (csrc/stream_pick.h has the story); there is no query for the mapping, so it
    is measured with two spin kernels -- one spin time when the queues differ, two when they are the same.  Rejected candidates
    stay alive until the search ends so that the runtime does not hand the same queue out again."""
    cycles = 400000
    ev = [torch.cuda.Event(enable_timing=True) for _ in range(2)]

    def timed(other):
        best = 1e9
        for _ in range(2):
            if other is not None:
                other.wait_stream(cur)
            ev[0].record(cur)
            torch.cuda._sleep(cycles)
            if other is not None:
                with torch.cuda.stream(other):
                    torch.cuda._sleep(cycles)
                cur.wait_stream(other)
            ev[1].record(cur)
            ev[1].synchronize()
            best = min(best, ev[0].elapsed_time(ev[1]))
        return best
    single = timed(None)
    rejected = []
    for _ in range(tries):
        cand = torch.cuda.Stream()
        if timed(cand) < 1.6 * single:
            return cand
        rejected.append(cand)
    return rejected[0]


@contextlib.contextmanager
def _no_finalizers_during_capture():
    """Collect dead Python cycles NOW and keep the cyclic collector off until the capture has ended.  A collector pass in the
    middle of a capture may finalize an object of an earlier phase that owns device resources (a `HipSetActor`, a vec-env, a
    `DeviceTrainer`: their destructors call hipFree / hipStreamDestroy), and any such call invalidates a global-mode capture
    (`hipErrorStreamCaptureInvalidated` at the next launch).  torch.cuda.graph collected unconditionally in older releases; this one
    only does with torch.compiler.config.force_cudagraph_gc -- found as a test failure that came and went with the number of
    objects pytest had allocated before the capture."""
    gc.collect()
    was = gc.isenabled()
    gc.disable()
    try:
        yield
    finally:
        if was:
            gc.enable()


class GraphedUpdates(object):
    """`Agent.update` captured into hipGraphs (MI355X: the update is ~3 400 small launches -- PyTorch autograd through three
    SET networks at batch 100 plus the HIP target kernels -- and launch-bound when issued eagerly: 50 ms; a replay is
    GPU-bound: 32 ms).  One graph per (morphology, with / without the delayed actor step); the batch is copied into static
    tensors and the target-policy noise redrawn into a static tensor before each replay, so the replayed arithmetic is
    the eager update's.  Everything reachable from `Agent.update` is capturable: the SET kernels are launched on the
    capturing stream (their side-stream fork / join becomes part of the graph), parameters are read live, Adam runs with
    `capturable=True`, the reward statistics stay on the device.

    Capture rules honoured here: every (morphology, batch size) is run eagerly first (so the SET handles have cached the
    batch structure and grown their workspace to its final size before any pointer is baked into a graph)."""

    def __init__(self, agent, batch_size):
        self.agent, self.B = agent, int(batch_size)
        dev = agent.device
        if dev.type != "cuda":
            raise RuntimeError("GraphedUpdates needs the agent on the GPU")
        if not train_ops.ENABLED:
            # round 6: a config-5 run of this combination went NaN in its third round (profiles/r6_takeoff/bisect
            # r6_takeoff_vendor_graphed_*); the vendor path is an A/B arm for EAGER updates, nobody has made its capture sound
            raise RuntimeError("GraphedUpdates replays this library's own training kernels: with SGRL_TRAIN_GEMM=0 (vendor kernels) run the "
                               "updates eagerly (DeviceTrainer(graph_updates=False))")
        for opt in (agent.actor_optimizer, agent.critic_optimizer):
            if opt.state:
                raise RuntimeError("switch to graphed updates before the first optimizer step (Adam's step counters must be device tensors)")
            for g in opt.param_groups:
                g["capturable"] = True
        self.slots = {}            # key -> dict(static tensors, graphs)
        self.warmed = set()
        # True (tools/diag/split_update_graphs_probe.py): target chain and critic forward as graphs of their own, replayed on two
        # streams -- measured 1-2 % faster per update (LAB_LOG round 3), not the default
        self.split = False
        self._cap_stream = self._side = None

    def _slot(self, key, graph, L):
        sl = self.slots.get(key)
        if sl is None:
            dev, B = self.agent.device, self.B
            z = lambda *sh: torch.zeros(sh, dtype=torch.float32, device=dev)
            sl = {"graph": graph, "batch": {"obs": z(B, 41 * L), "action": z(B, 3 * L), "next_obs": z(B, 41 * L), "reward": z(B, 1),
                                            "done": z(B, 1)}, "noise": z(B, 3 * L), "graphs": {}, "out": {}, "stamp": {}}
            self.slots[key] = sl
        return sl

    def _hip_handles(self):
        """The HIP handles whose forwards the update graphs record: the target actor and the two target critics."""
        out = []
        for mod in (self.agent.actor_target, self.agent.critic_target):
            h = getattr(mod, "_hip", None)
            out.extend([h] if hasattr(h, "h") else [getattr(h, "q1", None), getattr(h, "q2", None)])
        return [hh for hh in out if hh is not None]

    def _workspace_stamp(self):
        """What the captured HIP target kernels point into, per handle: the workspace's size and the handle's GENERATION
        (include/sgrl_set.h sgrl_set_generation: bumped whenever the handle frees device memory a recorded forward may
        reference -- a batch structure evicted from its cache, the flat weight buffers of a rebinding, a regrown workspace --
        or is told to change the form of its tile products)."""
        return tuple((int(hh.L.sgrl_set_workspace_bytes(hh.h)), int(hh.L.sgrl_set_generation(hh.h))) for hh in self._hip_handles())

    def _load(self, sl, data_batch):
        for k, t in sl["batch"].items():
            t.copy_(data_batch[k].reshape(t.shape))
        sl["noise"].normal_(0, self.agent.args.policy_noise)

    def warm(self, key, graph, L, data_batch, iters=3, first_it=0):
        """Eager updates on a side stream (PyTorch's capture protocol).  They are REAL updates: the caller passes the first
        `iters` iterations of its schedule (`data_batch` may be a callable returning a fresh batch per iteration, `first_it`
        the index of the first one) instead of adding updates of its own.  Returns the loss dicts."""
        sl = self._slot(key, graph, L)
        self.agent.change_morphology(graph)
        s = torch.cuda.Stream()
        s.wait_stream(torch.cuda.current_stream())
        outs = []
        with torch.cuda.stream(s):
            for it in range(first_it, first_it + iters):
                self._load(sl, data_batch() if callable(data_batch) else data_batch)
                outs.append(self.agent.update(sl["batch"], it, noise=sl["noise"], lazy_stats=True, skip_unused_critic_grads=True))
        torch.cuda.current_stream().wait_stream(s)
        self.warmed.add(key)
        return outs

    def update(self, key, graph, L, data_batch, it):
        """Same contract as Agent.update(data_batch, it) for the morphology `graph` (dict) identified by `key`."""
        if data_batch["obs"].shape[0] != self.B:
            self.agent.change_morphology(graph)         # short batch (buffer not yet filled): eager
            return self.agent.update(data_batch, it)
        sl = self._slot(key, graph, L)
        if key not in self.warmed:
            raise RuntimeError("warm(%r, ...) every morphology before the first graphed update" % (key,))
        flag = 0 if it % self.agent.args.policy_freq == 0 else 1
        self._load(sl, data_batch)
        # A graph bakes the addresses of the SET handles' workspaces.  They only ever grow, and every morphology is run eagerly
        # before any capture -- but a graph captured before a LATER regrowth would fault on replay (a GPU memory fault, not an
        # exception): the workspaces' sizes are compared with those at capture time and such a graph is captured again.
        stamp = self._workspace_stamp()
        owner = (id(self), key, flag)
        if flag in sl["graphs"] and sl["stamp"].get(flag) != stamp:
            del sl["graphs"][flag]
            release_tables(owner)
        global _capture_owner
        if flag not in sl["graphs"] and self.split:
            # Three graphs, all recorded on ONE capture stream (autograd's nodes then belong to one stream: no cross-stream
            # hand-overs inside the backward) but each with its own memory pool: the target chain and the critics' forward are
            # REPLAYED side by side, so neither may reuse a block the other has freed.
            self.agent.change_morphology(graph)
            if self._cap_stream is None:
                self._cap_stream, self._side = torch.cuda.Stream(), _concurrent_stream(torch.cuda.current_stream())
            gs = [torch.cuda.CUDAGraph() for _ in range(3)]
            _stock_pinned(4)
            _capture_owner = owner
            with _no_finalizers_during_capture():
                with torch.cuda.graph(gs[0], stream=self._cap_stream):
                    rb, tq = self.agent.update_targets(sl["batch"], sl["noise"])
                with torch.cuda.graph(gs[1], stream=self._cap_stream):
                    q1, q2 = self.agent.update_critic_forward(sl["batch"])
                with torch.cuda.graph(gs[2], stream=self._cap_stream):
                    sl["out"][flag] = self.agent.update_finish(sl["batch"], flag, rb, tq, q1, q2, lazy_stats=True,
                                                               skip_unused_critic_grads=True)
            _capture_owner = None
            sl["graphs"][flag] = gs
            sl["stamp"][flag] = self._workspace_stamp()
        if flag not in sl["graphs"]:
            self.agent.change_morphology(graph)
            g = torch.cuda.CUDAGraph()
            dump = os.environ.get("SGRL_GRAPH_DUMP")      # diagnostics: <dir> receives one .dot file per captured graph
            if dump:
                g.enable_debug_mode()
            _stock_pinned(4)
            _capture_owner = owner
            try:
                with _no_finalizers_during_capture(), torch.cuda.graph(g):
                    sl["out"][flag] = self.agent.update(sl["batch"], flag, noise=sl["noise"], lazy_stats=True,
                                                        skip_unused_critic_grads=True)
            finally:
                _capture_owner = None
            if dump:
                g.debug_dump(os.path.join(dump, "update_%s_flag%d.dot" % (key, flag)))
            sl["graphs"][flag] = g           # capturing records the work without running it
            sl["stamp"][flag] = self._workspace_stamp()
        g = sl["graphs"][flag]
        if isinstance(g, list):
            cur = torch.cuda.current_stream()
            self._side.wait_stream(cur)          # the batch / noise just loaded on the caller's stream
            g[0].replay()
            with torch.cuda.stream(self._side):
                g[1].replay()
            cur.wait_stream(self._side)
            g[2].replay()
        else:
            g.replay()
        # a replay runs no host code: the writes of its optimizer / soft-update kernels are announced here
        touched = list(self.agent.critic.parameters())
        if flag == 0:
            for mod in (self.agent.actor, self.agent.actor_target, self.agent.critic_target):
                touched.extend(mod.parameters())
        _touched(touched)
        # the capture's output tensors are overwritten by the next replay: hand out copies
        return {k: (v.clone() if torch.is_tensor(v) else v) for k, v in sl["out"][flag].items()}
