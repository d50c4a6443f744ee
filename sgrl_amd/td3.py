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
import types

import torch
import torch.nn as nn
import torch.nn.functional as F

from .set_policy import SECritic, SEPolicy, default_args


def soft_update_network(source_network, target_network, tau):
    """target <- tau * source + (1 - tau) * target, parameter by parameter (reference common/functional.py:7-10)."""
    with torch.no_grad():
        for target_param, local_param in zip(target_network.parameters(), source_network.parameters()):
            target_param.data.copy_(tau * local_param.data + (1 - tau) * target_param.data)


def default_train_args(**over):
    """The reference's TD3 / SET hyper-parameters (reference arguments.py:55-160, configs/default.py:9-12)."""
    a = default_args()
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
        if atype not in ("set", "swat") or ctype not in ("set", "swat"):
            raise NotImplementedError("actor / critic types 'set' (HIP fast path) and 'swat' (PyTorch) are built; "
                                      "'smp' and 'mlp' are not (SURVEY 8 f4)")
        self.networks = {}
        from .swat_policy import CriticStructurePolicy, StructurePolicy

        def actor():
            if atype == "swat":
                return StructurePolicy(args.limb_obs_size, args.limb_action_size, args.msg_dim, args.batch_size, args.max_action,
                                       args.max_children, args.disable_fold, args.td, args.bu, args, device=device)
            return SEPolicy(args.limb_obs_size, args.limb_action_size, args.msg_dim, args.batch_size, args.max_action,
                            args.max_children, args.disable_fold, args.td, args.bu, args, device=device, use_hip=use_hip)

        def critic():
            if ctype == "swat":
                return CriticStructurePolicy(args.limb_obs_size, args.limb_action_size, args.msg_dim, args.batch_size,
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

    def update(self, data_batch, it, noise=None):
        """One TD3 step on a batch of ONE morphology (reference agent.py:117-183).  `noise` (tests only) replaces the
        N(0, policy_noise) draw of agent.py:128 so that a run can be compared with the reference number for number."""
        args = self.args
        obs_batch, action_batch = data_batch["obs"], data_batch["action"]
        next_obs_batch, reward_batch, done_batch = data_batch["next_obs"], data_batch["reward"], data_batch["done"]
        reward_batch = reward_batch * self.reward_scale
        with torch.no_grad():
            if noise is None:
                noise = torch.zeros_like(action_batch).normal_(0, args.policy_noise)
            noise = noise.clamp(-args.noise_clip, args.noise_clip)
            next_action = (self.actor_target(next_obs_batch) + noise).clamp(-args.max_action, args.max_action)
            target_Q1, target_Q2 = self.critic_target(next_obs_batch, next_action)    # per-limb values [B, L]
            target_Q = torch.min(target_Q1, target_Q2)
            target_Q = reward_batch + ((1.0 - done_batch) * args.discount * target_Q)  # reward [B, 1] broadcast over limbs
        current_Q1, current_Q2 = self.critic(obs_batch, action_batch)
        critic_loss = F.mse_loss(current_Q1, target_Q) + F.mse_loss(current_Q2, target_Q)
        self.critic_optimizer.zero_grad()
        critic_loss.backward()
        if args.grad_clipping_value > 0:
            torch.nn.utils.clip_grad_norm_(self.critic.parameters(), args.grad_clipping_value)
        self.critic_optimizer.step()
        loss_dict = {"loss/critic_loss": critic_loss, "misc/train_reward_mean": torch.mean(reward_batch).item(),
                     "misc/train_reward_var": torch.var(reward_batch).item()}
        if it % args.policy_freq == 0:       # delayed policy update
            actor_loss = -self.critic.Q1(obs_batch, self.actor(obs_batch)).mean()
            self.actor_optimizer.zero_grad()
            actor_loss.backward()
            if args.grad_clipping_value > 0:
                torch.nn.utils.clip_grad_norm_(self.actor.parameters(), args.grad_clipping_value)
            self.actor_optimizer.step()
            self.try_update_target_network()
            loss_dict.update({"loss/actor_loss": actor_loss})
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
