"""SAC with Stable-Baselines3's call shape -- the algorithm the reference actually trains (train_agent.py:58-92):

    SAC("MultiInputPolicy", env, policy_kwargs=dict(features_extractor_class=AugmentedNatureCNN,
        share_features_extractor=True, net_arch=[256, 256]), buffer_size=..., batch_size=..., verbose=1,
        tensorboard_log=..., [replay_buffer_class=HerReplayBuffer, replay_buffer_kwargs=dict(n_sampled_goal=4,
        goal_selection_strategy='future', online_sampling=True, max_episode_length=...), learning_starts=...])
    model.learn(total_timesteps, callback=[...]);  model.predict(obs, deterministic=True);  model.save / SAC.load

SB3 defaults are kept (lr 3e-4, tau 0.005, gamma 0.99, train_freq 1, gradient_steps 1, ent_coef 'auto' with target
entropy -|A|, state-dependent log_std clipped to [-20, 2], tanh-squashed Gaussian, twin Q networks with ReLU MLPs, the
shared features extractor trained by the critic loss only). MI355X layout: the replay buffer lives in HBM as uint8
observations (500 k x 20 KB = 10 GB, next observations stored once per step for all envs), sampling, HER relabelling and
the update are device work; with world_size > 1 the gradients of a step are flattened and all-reduced once, like PPO's.
Works with the batched GPU env (GpuVecEnv: tensors in / out) and with the one-env numpy DummyVecEnv.
"""
import io
import math
import os
import time
import zipfile

import numpy as np
import torch as th
import torch.distributed as dist
from torch import nn

from .callbacks import CallbackList

LOG_STD_MIN, LOG_STD_MAX = -20.0, 2.0


def _to_t(x, device, dtype=None):
    t = x.to(device) if isinstance(x, th.Tensor) else th.as_tensor(np.asarray(x), device=device)
    return t if dtype is None else t.to(dtype)


def _mlp(in_dim, arch, out_dim=None):
    layers, d = [], in_dim
    for h in arch:
        layers += [nn.Linear(d, h), nn.ReLU()]
        d = h
    if out_dim is not None:
        layers.append(nn.Linear(d, out_dim))
    return nn.Sequential(*layers), d


class SACPolicy(nn.Module):
    """Actor + twin critics + target critics over one features extractor class (MultiInputPolicy of the reference call)."""

    def __init__(self, observation_space, action_space, features_extractor_class=None, features_extractor_kwargs=None,
                 share_features_extractor=True, net_arch=(256, 256), n_critics=2, normalize_images=True):
        super().__init__()
        if features_extractor_class is None:
            from ..models.feature_extractor import AugmentedNatureCNN as features_extractor_class
        kw = features_extractor_kwargs or {}
        self.normalize_images = normalize_images
        self.share_features_extractor = share_features_extractor
        self.action_dim = int(np.prod(action_space.shape))
        mk = lambda: features_extractor_class(observation_space, **kw)
        self.actor_features = mk()
        self.critic_features = self.actor_features if share_features_extractor else mk()
        self.critic_target_features = mk()
        self._fused_preprocess = bool(getattr(self.actor_features, "accepts_raw_uint8", False))
        fd = self.actor_features.features_dim
        arch = list(net_arch["pi"] if isinstance(net_arch, dict) else net_arch)
        qarch = list(net_arch["qf"] if isinstance(net_arch, dict) else net_arch)
        self.latent_pi, pd = _mlp(fd, arch)
        self.mu = nn.Linear(pd, self.action_dim); self.log_std = nn.Linear(pd, self.action_dim)
        self.q_nets = nn.ModuleList([_mlp(fd + self.action_dim, qarch, 1)[0] for _ in range(n_critics)])
        self.q_targets = nn.ModuleList([_mlp(fd + self.action_dim, qarch, 1)[0] for _ in range(n_critics)])
        self.critic_target_features.load_state_dict(self.critic_features.state_dict())
        self.q_targets.load_state_dict(self.q_nets.state_dict())
        for p in list(self.critic_target_features.parameters()) + list(self.q_targets.parameters()):
            p.requires_grad_(False)

    def _prep(self, obs):
        o = obs["observation"]
        if o.dtype == th.uint8:
            if o.is_cuda and self.normalize_images and getattr(self, "_fused_preprocess", False):
                return {"observation": o}        # AugmentedNatureCNN normalises and lays out raw uint8 in one kernel
            o = o.float()
            if self.normalize_images:
                o = o / 255.0
        return {"observation": o}

    # ---- actor
    def actor_parameters(self):
        ps = list(self.latent_pi.parameters()) + list(self.mu.parameters()) + list(self.log_std.parameters())
        if not self.share_features_extractor:
            ps += list(self.actor_features.parameters())
        return ps

    def critic_parameters(self):
        return list(self.critic_features.parameters()) + list(self.q_nets.parameters())

    def action_log_prob(self, obs, deterministic=False, detach_features=None):
        f = self.actor_features(self._prep(obs))
        if self.share_features_extractor if detach_features is None else detach_features:
            f = f.detach()                                    # SB3: with a shared extractor only the critic loss trains it
        h = self.latent_pi(f)
        mean = self.mu(h); log_std = self.log_std(h).clamp(LOG_STD_MIN, LOG_STD_MAX)
        if deterministic:
            return th.tanh(mean), None
        std = log_std.exp()
        g = mean + std * th.randn_like(mean)
        a = th.tanh(g)
        logp = (-0.5 * ((g - mean) / std) ** 2 - log_std - 0.5 * math.log(2 * math.pi)).sum(-1)
        logp = logp - th.log(1.0 - a * a + 1e-6).sum(-1)      # tanh change of variables (SB3's SquashedDiagGaussian epsilon)
        return a, logp

    # ---- critics
    def q_values(self, obs, actions, target=False):
        fe, nets = (self.critic_target_features, self.q_targets) if target else (self.critic_features, self.q_nets)
        x = th.cat([fe(self._prep(obs)), actions], dim=1)
        return [q(x).squeeze(-1) for q in nets]

    @th.no_grad()
    def polyak(self, tau):
        for src, dst in ((self.critic_features, self.critic_target_features), (self.q_nets, self.q_targets)):
            for p, pt in zip(src.parameters(), dst.parameters()):
                pt.mul_(1.0 - tau).add_(p.detach(), alpha=tau)
            for b, bt in zip(src.buffers(), dst.buffers()):
                bt.copy_(b)


class ReplayBuffer:
    """Transitions of N parallel envs in device memory: uint8 observations [T, N, ...], next observation stored beside it
    (the auto-reset env returns the reset observation after a done, so obs[t + 1] is not the successor there)."""

    def __init__(self, buffer_size, observation_space, action_space, device, n_envs=1, **_):
        self.n_envs, self.device = n_envs, device
        self.T = max(1, buffer_size // n_envs)
        shp = tuple(observation_space["observation"].shape)
        A = int(np.prod(action_space.shape))
        z = lambda *s, dt=th.float32: th.zeros((self.T, n_envs) + s, dtype=dt, device=device)
        self.obs, self.next_obs = z(*shp, dt=th.uint8), z(*shp, dt=th.uint8)
        self.actions, self.rewards, self.dones = z(A), z(), z()
        self.achieved, self.desired, self.next_achieved, self.next_desired = z(2), z(2), z(2), z(2)
        self.pos, self.full = 0, False

    def size(self):
        return (self.T if self.full else self.pos) * self.n_envs

    def stage_obs(self, obs):
        """The pre-step half of add(), written BEFORE the env steps (an env may hand out scratch tensors it rewrites in place:
        robot_env.py:35,223 does, SB3 copies for the same reason); add(None, ...) then completes the row."""
        t = self.pos
        self.obs[t].copy_(obs["observation"])
        if "achieved_goal" in obs:
            self.achieved[t].copy_(obs["achieved_goal"]); self.desired[t].copy_(obs["desired_goal"])

    def add(self, obs, next_obs, action, reward, done, infos=None):
        t = self.pos
        if obs is not None:
            self.stage_obs(obs)
        self.next_obs[t].copy_(next_obs["observation"])
        self.actions[t].copy_(action); self.rewards[t].copy_(reward); self.dones[t].copy_(done)
        if "achieved_goal" in next_obs:
            self.next_achieved[t].copy_(next_obs["achieved_goal"]); self.next_desired[t].copy_(next_obs["desired_goal"])
        self.pos += 1
        if self.pos == self.T:
            self.pos, self.full = 0, True

    def _indices(self, batch_size, generator=None):
        hi = self.T if self.full else self.pos
        t = th.randint(0, hi, (batch_size,), device=self.device, generator=generator)
        e = th.randint(0, self.n_envs, (batch_size,), device=self.device, generator=generator)
        return t, e

    def sample(self, batch_size, generator=None):
        t, e = self._indices(batch_size, generator)
        return dict(obs={"observation": self.obs[t, e]}, next_obs={"observation": self.next_obs[t, e]}, actions=self.actions[t, e],
                    rewards=self.rewards[t, e], dones=self.dones[t, e])


class FlatReplayBuffer:
    """Replay buffer for the time-sliced engine: transitions arrive in batches of varying size (the envs that finished a macro step this
    tick), so storage is one flat ring of `buffer_size` rows and a batch is scattered behind a device-side write pointer -- no host
    sync per tick. Same sample() contract as ReplayBuffer."""

    def __init__(self, buffer_size, observation_space, action_space, device, **_):
        self.cap, self.device = int(buffer_size), device
        shp = tuple(observation_space["observation"].shape); A = int(np.prod(action_space.shape))
        z = lambda *sh, dt=th.float32: th.zeros((self.cap + 1,) + sh, dtype=dt, device=device)      # row cap: dump row of masked scatters
        self.obs, self.next_obs = z(*shp, dt=th.uint8), z(*shp, dt=th.uint8)
        self.actions, self.rewards, self.dones = z(A), z(), z()
        self.ptr = th.zeros(1, dtype=th.int64, device=device)           # rows written so far (monotonic)
        self._host_size = 0

    def add_rows(self, mask, obs, next_obs, action, reward, done):
        """Append the rows where mask is set (all arguments have one row per ready-list slot)."""
        m = mask.long()
        pos = (self.ptr + th.cumsum(m, 0) - 1) % self.cap
        pos = th.where(mask, pos, th.full_like(pos, self.cap))
        self.obs.index_copy_(0, pos, obs); self.next_obs.index_copy_(0, pos, next_obs)
        self.actions.index_copy_(0, pos, action); self.rewards.index_copy_(0, pos, reward); self.dones.index_copy_(0, pos, done)
        self.ptr += m.sum()

    def size(self):
        return min(self._host_size, self.cap)

    def sync_size(self):
        self._host_size = int(self.ptr.item())
        return self.size()

    def sample(self, batch_size, generator=None):
        hi = max(1, self.size())
        i = th.randint(0, hi, (batch_size,), device=self.device, generator=generator)
        return dict(obs={"observation": self.obs[i]}, next_obs={"observation": self.next_obs[i]}, actions=self.actions[i], rewards=self.rewards[i], dones=self.dones[i])


class FlatHerReplayBuffer(FlatReplayBuffer):
    """Hindsight relabelling ('future', n_sampled_goal: train_agent.py:63-69) for the time-sliced collector. Transitions of an env arrive in
    episode order but at arbitrary ring positions, so a row cannot find its episode's later rows by address. Every running episode owns a
    slot of a trajectory store -- the goal achieved after each of its steps, [slot, step, 2] -- and a row remembers (slot, step, generation
    of the slot). An episode that ends publishes its length; its slot is recycled after `n_slots` newer episodes, which the generation tells
    a sampled row (it is then no longer relabelled, like a row of an episode still open -- SB3's online sampling draws from complete
    episodes). Everything is batched tensor work on the ready-list rows of a tick: no host sync, no per-episode loop.
    The relabelled reward swaps the goal-dependent term e^-|dg - ag| (robot_env.py:268-271) exactly as HerReplayBuffer does."""

    def __init__(self, buffer_size, observation_space, action_space, device, n_envs=1, n_sampled_goal=4, goal_selection_strategy="future",
                 online_sampling=True, max_episode_length=400, n_slots=None, **_):
        super().__init__(buffer_size, observation_space, action_space, device)
        if str(goal_selection_strategy).lower().split(".")[-1] != "future":
            raise NotImplementedError("only goal_selection_strategy='future' (the reference's choice) is built")
        self.her_ratio = 1.0 - 1.0 / (n_sampled_goal + 1)
        self.L = int(max_episode_length or 400)
        self.n_envs = int(n_envs)
        # enough slots that a row usually outlives its slot's recycling only when the ring has overwritten it anyway
        self.n_slots = int(n_slots) if n_slots else max(2 * self.n_envs, min(4 * self.n_envs + self.cap // 16, 1 << 16))
        z = lambda *sh, dt=th.float32: th.zeros((self.cap + 1,) + sh, dtype=dt, device=device)
        self.achieved, self.desired, self.next_achieved, self.next_desired = z(2), z(2), z(2), z(2)
        self.row_slot, self.row_step, self.row_gen = z(dt=th.int64), z(dt=th.int64), z(dt=th.int64)
        self.traj_goal = th.zeros(self.n_slots + 1, self.L, 2, device=device)           # slot n_slots: dump
        self.traj_len = th.zeros(self.n_slots + 1, dtype=th.int64, device=device)         # 0 = episode still open
        self.slot_gen = th.zeros(self.n_slots + 1, dtype=th.int64, device=device)
        self.cur_slot = th.arange(self.n_envs + 1, dtype=th.int64, device=device) % self.n_slots   # env n_envs: dump
        self.cur_step = th.zeros(self.n_envs + 1, dtype=th.int64, device=device)
        self.alloc = th.full((1,), self.n_envs % self.n_slots, dtype=th.int64, device=device)      # next slot to hand out
        # Who writes a slot: slots are handed out round-robin, and an episode that outlasts n_slots - n_envs completions of the other envs finds its slot
        # given to a newcomer. The displaced episode then writes to the dump slot for the rest of its life (its earlier rows are already dead: the
        # newcomer bumped the slot's generation), so two envs never mix their goals in one trajectory.
        self.slot_owner = th.full((self.n_slots + 1,), self.n_envs, dtype=th.int64, device=device)
        self.slot_owner[:min(self.n_envs, self.n_slots)] = th.arange(min(self.n_envs, self.n_slots), device=device)

    def add_rows(self, mask, obs, next_obs, action, reward, done, env=None, goals=None):
        """env: env id of every ready-list row (n_envs for masked-out rows); goals = (achieved, desired, next_achieved, next_desired), [rows, 2] each."""
        m = mask.long()
        pos = (self.ptr + th.cumsum(m, 0) - 1) % self.cap
        pos = th.where(mask, pos, th.full_like(pos, self.cap))
        super().add_rows(mask, obs, next_obs, action, reward, done)
        e = th.where(mask, env, th.full_like(env, self.n_envs))
        ag, dg, nag, ndg = goals
        self.achieved.index_copy_(0, pos, ag); self.desired.index_copy_(0, pos, dg); self.next_achieved.index_copy_(0, pos, nag); self.next_desired.index_copy_(0, pos, ndg)
        slot, step = self.cur_slot[e], self.cur_step[e].clamp(max=self.L - 1)
        slot = th.where(mask & (self.slot_owner[slot] == e), slot, th.full_like(slot, self.n_slots))       # masked-out rows and displaced episodes: the dump slot
        self.row_slot.index_copy_(0, pos, slot); self.row_step.index_copy_(0, pos, step); self.row_gen.index_copy_(0, pos, self.slot_gen[slot])
        self.traj_goal[slot, step] = nag                      # the goal achieved AFTER this step: what 'future' hands to earlier steps
        fin = mask & (done > 0)
        self.cur_step.index_copy_(0, e, th.where(fin, th.zeros_like(step), step + 1)); self.cur_step[self.n_envs] = 0
        # finished episodes publish their length and their envs move on to fresh slots (recycled oldest first: generation + 1)
        self.traj_len.index_copy_(0, th.where(fin, slot, th.full_like(slot, self.n_slots)), step + 1); self.traj_len[self.n_slots] = 0
        f = fin.long()
        new = (self.alloc + th.cumsum(f, 0) - 1) % self.n_slots
        new = th.where(fin, new, th.full_like(new, self.n_slots))
        self.slot_gen.index_add_(0, new, f); self.traj_len.index_fill_(0, new, 0); self.slot_gen[self.n_slots] = 0
        self.slot_owner.index_copy_(0, new, th.where(fin, e, th.full_like(e, self.n_envs))); self.slot_owner[self.n_slots] = self.n_envs
        self.cur_slot.index_copy_(0, th.where(fin, e, th.full_like(e, self.n_envs)), new); self.cur_slot[self.n_envs] = self.n_slots
        self.alloc += f.sum()

    @staticmethod
    def her_term(desired, achieved):
        return th.exp(-th.linalg.norm(desired - achieved, dim=-1))

    def sample(self, batch_size, generator=None):
        hi = max(1, self.size())
        i = th.randint(0, hi, (batch_size,), device=self.device, generator=generator)
        slot, step = self.row_slot[i], self.row_step[i]
        L = self.traj_len[slot]
        alive = (self.slot_gen[slot] == self.row_gen[i]) & (L > step)            # the row's episode is complete and its slot not yet recycled
        relabel = (th.rand(batch_size, device=self.device, generator=generator) < self.her_ratio) & alive
        span = (L - 1 - step).clamp(min=0)                                        # 'future': this step ... the episode's last
        k = step + th.minimum((th.rand(batch_size, device=self.device, generator=generator) * (span + 1).float()).long(), span)
        new_goal = self.traj_goal[slot, k.clamp(max=self.L - 1)]
        old = self.her_term(self.next_desired[i], self.next_achieved[i]); new = self.her_term(new_goal, self.next_achieved[i])
        rewards = th.where(relabel, self.rewards[i] - old + new, self.rewards[i])
        desired = th.where(relabel[:, None], new_goal, self.desired[i])
        return dict(obs={"observation": self.obs[i], "achieved_goal": self.achieved[i], "desired_goal": desired},
                    next_obs={"observation": self.next_obs[i], "achieved_goal": self.next_achieved[i], "desired_goal": desired},
                    actions=self.actions[i], rewards=rewards, dones=self.dones[i], relabelled=relabel)


class HerReplayBuffer(ReplayBuffer):
    """Hindsight relabelling with SB3's HerReplayBuffer arguments (train_agent.py:63-69): for n_sampled_goal out of
    n_sampled_goal + 1 sampled transitions the desired goal is replaced by a goal achieved later in the same episode
    ('future') and the reward recomputed. For this env only the `her_buffer` term e^-|dg - ag| of the reward depends on the
    goal (robot_env.py:268-271), so the relabelled reward is the stored one minus the old term plus the new one."""

    def __init__(self, buffer_size, observation_space, action_space, device, n_envs=1, n_sampled_goal=4,
                 goal_selection_strategy="future", online_sampling=True, max_episode_length=None, **_):
        super().__init__(buffer_size, observation_space, action_space, device, n_envs)
        if str(goal_selection_strategy).lower().split(".")[-1] != "future":
            raise NotImplementedError("only goal_selection_strategy='future' (the reference's choice) is built")
        self.her_ratio = 1.0 - 1.0 / (n_sampled_goal + 1)
        self.max_episode_length = max_episode_length
        self.ep_end = th.full((self.T, n_envs), -1, dtype=th.int64, device=device)    # buffer row of the episode's last step, -1 = open
        self._ep_start = th.zeros(n_envs, dtype=th.int64, device=device)              # row where each env's running episode began

    def add(self, obs, next_obs, action, reward, done, infos=None):
        t = self.pos
        self.ep_end[t] = -1
        super().add(obs, next_obs, action, reward, done, infos)
        fin = done > 0
        if bool(fin.any()):
            # close the finished episodes: every row of the episode learns where it ends (rows may wrap around the ring)
            for e in fin.nonzero().flatten().tolist():
                s = int(self._ep_start[e]); rows = th.arange(s, s + ((t - s) % self.T) + 1, device=self.device) % self.T
                self.ep_end[rows, e] = t
                self._ep_start[e] = (t + 1) % self.T
        # rows of old episodes are overwritten oldest first, so the [t, end] range of a surviving row stays intact

    @staticmethod
    def her_term(desired, achieved):
        return th.exp(-th.linalg.norm(desired - achieved, dim=-1))

    def sample(self, batch_size, generator=None):
        t, e = self._indices(batch_size, generator)
        rewards = self.rewards[t, e].clone()
        end = self.ep_end[t, e]
        relabel = (th.rand(batch_size, device=self.device, generator=generator) < self.her_ratio) & (end >= 0)
        # 'future': a uniformly drawn step between this one and the end of its episode (ring distance)
        span = (end - t) % self.T
        off = (th.rand(batch_size, device=self.device, generator=generator) * (span + 1).float()).long().clamp(max=self.T - 1)
        ft = (t + th.minimum(off, span)) % self.T
        new_goal = self.next_achieved[ft, e]
        # the stored reward holds e^-|dg - ag| of the POST-step goals (robot_env.py:176-178, 268-271): take exactly that term out
        old = self.her_term(self.next_desired[t, e], self.next_achieved[t, e]); new = self.her_term(new_goal, self.next_achieved[t, e])
        rewards = th.where(relabel, rewards - old + new, rewards)
        desired = th.where(relabel[:, None], new_goal, self.desired[t, e])
        return dict(obs={"observation": self.obs[t, e], "achieved_goal": self.achieved[t, e], "desired_goal": desired},
                    next_obs={"observation": self.next_obs[t, e], "achieved_goal": self.next_achieved[t, e], "desired_goal": desired},
                    actions=self.actions[t, e], rewards=rewards, dones=self.dones[t, e], relabelled=relabel)


class SAC:
    def __init__(self, policy, env, learning_rate=3e-4, buffer_size=1_000_000, learning_starts=100, batch_size=256, tau=0.005, gamma=0.99,
                 train_freq=1, gradient_steps=1, replay_buffer_class=None, replay_buffer_kwargs=None, ent_coef="auto", target_entropy="auto",
                 policy_kwargs=None, verbose=0, tensorboard_log=None, device=None, seed=None, async_slice=0, async_capacity=None, async_budget_us=0):
        """async_slice > 0: collect over the time-sliced engine (grip_batch_advance) instead of lock-step vector-env steps -- every tick
        gives each env at most `async_slice` calls of physics.step(), the envs that finished are rendered, their transition (previous
        observation and action of that env, reward, done, new observation) goes into a flat replay ring, the actor decides for them, and
        `gradient_steps` updates follow every `train_freq` ticks. Off-policy replay absorbs the asynchrony natively: no lag correction."""
        self.async_slice, self.async_capacity, self.async_budget_us = int(async_slice), async_capacity, int(async_budget_us)
        self.env = env
        self.n_envs = getattr(env, "num_envs", 1)
        self.device = th.device(device) if device is not None else getattr(env, "device", th.device("cuda" if th.cuda.is_available() else "cpu"))
        self.gamma, self.tau, self.batch_size = gamma, tau, batch_size
        self.learning_starts, self.train_freq, self.gradient_steps = learning_starts, train_freq, gradient_steps
        self.verbose, self.tensorboard_log = verbose, tensorboard_log
        self.policy_kwargs = dict(policy_kwargs or {})
        if seed is not None:
            th.manual_seed(seed)
        policy_class = SACPolicy if isinstance(policy, str) else policy
        self.policy = policy_class(env.observation_space, env.action_space, **self.policy_kwargs).to(self.device)
        self.distributed = dist.is_available() and dist.is_initialized() and dist.get_world_size() > 1
        if self.distributed:
            for p in self.policy.parameters():
                dist.broadcast(p.data, src=0)
        self.actor_opt = th.optim.Adam(self.policy.actor_parameters(), lr=learning_rate)
        self.critic_opt = th.optim.Adam(self.policy.critic_parameters(), lr=learning_rate)
        A = self.policy.action_dim
        self.target_entropy = -float(A) if target_entropy == "auto" else float(target_entropy)
        self.auto_ent = isinstance(ent_coef, str) and ent_coef.startswith("auto")
        init = float(ent_coef.split("_")[1]) if self.auto_ent and "_" in ent_coef else 1.0
        self.log_ent_coef = th.log(th.ones(1, device=self.device) * (init if self.auto_ent else float(ent_coef))).requires_grad_(self.auto_ent)
        self.ent_opt = th.optim.Adam([self.log_ent_coef], lr=learning_rate) if self.auto_ent else None
        if self.async_slice > 0:
            # the time-sliced collector scatters varying batches into a flat ring; hindsight relabelling (the reference's HerReplayBuffer call,
            # train_agent.py:57-79) gets the ring with per-episode goal trajectories
            if replay_buffer_class is None or replay_buffer_class is FlatReplayBuffer:
                self.replay_buffer = FlatReplayBuffer(buffer_size, env.observation_space, env.action_space, self.device)
            elif replay_buffer_class in (HerReplayBuffer, FlatHerReplayBuffer):
                self.replay_buffer = FlatHerReplayBuffer(buffer_size, env.observation_space, env.action_space, self.device, n_envs=self.n_envs,
                                                         **(replay_buffer_kwargs or {}))
            else:
                raise NotImplementedError("the time-sliced collector stores transitions in a FlatReplayBuffer or, for hindsight relabelling, a FlatHerReplayBuffer")
        else:
            rb = replay_buffer_class or ReplayBuffer
            self.replay_buffer = rb(buffer_size, env.observation_space, env.action_space, self.device, n_envs=self.n_envs, **(replay_buffer_kwargs or {}))
        self.num_timesteps, self._n_updates = 0, 0
        self._last_obs = None
        self.logger = {}
        self._tensor_env = hasattr(env, "device")

    # ------------------------------------------------------------------ helpers
    def _obs_t(self, obs):
        return {k: _to_t(v, self.device) for k, v in obs.items() if k in ("observation", "achieved_goal", "desired_goal")}

    def _scale(self, a):            # policy space [-1, 1] -> env action space
        low = th.as_tensor(self.env.action_space.low, device=self.device); high = th.as_tensor(self.env.action_space.high, device=self.device)
        return low + 0.5 * (a + 1.0) * (high - low)

    def _true_next(self, new_obs, infos):
        """next_obs as the replay buffer should see it. An auto-resetting engine returns the RESET goals for a finished env; the goals
        the reward of that step was computed with (robot_env.py:175-178) are the final object position and its projection on the
        target direction, which the engine reports in info['object_position']: use those for every env (identical for unfinished ones)."""
        if not (self._tensor_env and isinstance(infos, dict) and "object_position" in infos and "achieved_goal" in new_obs):
            return new_obs
        benv = getattr(self.env, "env", self.env)
        d = th.as_tensor(np.asarray(benv.target_direction, dtype=np.float32), device=self.device)
        ag = infos["object_position"][:, :2].to(self.device, th.float32)
        p = (ag * d).sum(-1, keepdim=True) / (d * d).sum(-1, keepdim=True)
        return {"observation": new_obs["observation"], "achieved_goal": ag, "desired_goal": p * d}

    def _allreduce(self, params):
        grads = [p.grad for p in params if p.grad is not None]
        if not grads:
            return
        flat = th.cat([g.reshape(-1) for g in grads])
        dist.all_reduce(flat, op=dist.ReduceOp.SUM); flat /= dist.get_world_size()
        off = 0
        for g in grads:
            n = g.numel(); g.copy_(flat[off:off + n].view_as(g)); off += n

    # ------------------------------------------------------------------ one gradient step (SB3 SAC.train)
    def train(self, gradient_steps=1, batch_size=None):
        bs = batch_size or self.batch_size
        pol = self.policy
        for _ in range(gradient_steps):
            b = self.replay_buffer.sample(bs)
            obs, nobs = b["obs"], b["next_obs"]
            a_pi, logp = pol.action_log_prob(obs)
            ent_coef = self.log_ent_coef.exp().detach()
            if self.auto_ent:
                ent_loss = -(self.log_ent_coef * (logp + self.target_entropy).detach()).mean()
                self.ent_opt.zero_grad(); ent_loss.backward(); self.ent_opt.step()
            with th.no_grad():
                na, nlogp = pol.action_log_prob(nobs)
                q_next = th.min(th.stack(pol.q_values(nobs, na, target=True)), dim=0).values - ent_coef * nlogp
                target_q = b["rewards"] + (1.0 - b["dones"]) * self.gamma * q_next
            qs = pol.q_values(obs, b["actions"])
            critic_loss = 0.5 * sum(nn.functional.mse_loss(q, target_q) for q in qs)
            self.critic_opt.zero_grad(); critic_loss.backward()
            if self.distributed:
                self._allreduce(pol.critic_parameters())
            self.critic_opt.step()
            # actor: the critics (and, shared, their extractor) are not updated by this loss
            with th.no_grad():
                f = pol.critic_features(pol._prep(obs))
            q_pi = th.min(th.stack([q(th.cat([f, a_pi], dim=1)).squeeze(-1) for q in pol.q_nets]), dim=0).values
            actor_loss = (ent_coef * logp - q_pi).mean()
            self.actor_opt.zero_grad(); actor_loss.backward()
            if self.distributed:
                self._allreduce(pol.actor_parameters())
            self.actor_opt.step()
            pol.polyak(self.tau)
            self._n_updates += 1
        self.logger = {"actor_loss": actor_loss.detach(), "critic_loss": critic_loss.detach(), "ent_coef": ent_coef, "n_updates": self._n_updates}
        return self.logger

    # ------------------------------------------------------------------ SB3 surface
    def _learn_async(self, total_timesteps, callback, log_interval):
        """Collection over the time-sliced engine (see __init__). Per-env previous decision lives in HBM (obs 20 KB x N)."""
        from .async_rollout import BatchEngineAdapter
        eng = self.env if hasattr(self.env, "advance") else BatchEngineAdapter(self.env, self.async_budget_us)
        N, A, dev = eng.num_envs, eng.action_dim, self.device
        Ccap = min(int(self.async_capacity) if self.async_capacity else max(1, N // 4), N)
        lst = th.full((Ccap,), -1, dtype=th.int32, device=dev); cnt = th.zeros(1, dtype=th.int32, device=dev)
        slot_act = th.zeros(Ccap, A, device=dev); stage = th.zeros((Ccap,) + tuple(eng.obs_shape), dtype=th.uint8, device=dev)
        prev_obs = th.zeros((N + 1,) + tuple(eng.obs_shape), dtype=th.uint8, device=dev); prev_act = th.zeros(N + 1, A, device=dev)
        has_prev = th.zeros(N + 1, dtype=th.bool, device=dev); ar = th.arange(Ccap, device=dev)
        her = isinstance(self.replay_buffer, FlatHerReplayBuffer)
        if her:                                             # goals of every env's previous decision point, and the target directions
            prev_ag = th.zeros(N + 1, 2, device=dev); prev_dg = th.zeros(N + 1, 2, device=dev)
            benv = getattr(self.env, "env", self.env)
            tdir = th.as_tensor(np.asarray(getattr(eng, "target_direction", getattr(benv, "target_direction", (1.0, 0.0))), dtype=np.float32), device=dev)
            tdir = tdir.expand(N, 2) if tdir.ndim == 1 else tdir
        eng.reset()
        t0, tick, last_sync = time.time(), 0, 0
        self._async_ticks = 0
        # episode statistics on the device (read by probes / callbacks without a per-tick sync): running return and length per env, totals of finished episodes
        ep_ret = th.zeros(N + 1, device=dev); ep_len = th.zeros(N + 1, device=dev)
        self.ep_stats = {"count": th.zeros(1, device=dev), "ret_sum": th.zeros(1, device=dev), "len_sum": th.zeros(1, device=dev)}
        while self.num_timesteps < total_timesteps:
            out = eng.advance(slot_act, self.async_slice, lst, cnt)
            eng.observe_list(lst, cnt, stage)
            valid = (ar < cnt) & (lst >= 0)
            env = th.where(valid, lst, th.full_like(lst, N)).long()                  # row N: dump
            envc = env.clamp(max=N - 1)
            if her:
                # the goals the reward of the finished macro step was computed with (robot_env.py:175-178): the object's final position and its
                # projection on the target direction -- from info, not from the auto-reset goal outputs; the goals of the NEW decision point
                # are the engine's goal outputs (reset goals after a done)
                d = tdir[envc]
                nag = out["object_position"][envc, :2].float()
                ndg = ((nag * d).sum(-1, keepdim=True) / (d * d).sum(-1, keepdim=True)) * d
                self.replay_buffer.add_rows(valid & has_prev[env], prev_obs[env], stage, prev_act[env], out["reward"][envc].float(), out["done"][envc].float(),
                                            env=env, goals=(prev_ag[env], prev_dg[env], nag, ndg))
                prev_ag.index_copy_(0, env, out["achieved_goal"][envc].float()); prev_dg.index_copy_(0, env, out["desired_goal"][envc].float())
            else:
                self.replay_buffer.add_rows(valid & has_prev[env], prev_obs[env], stage, prev_act[env], out["reward"][envc].float(), out["done"][envc].float())
            # episode bookkeeping of the listed envs
            stored = valid & has_prev[env]
            r_now = th.where(stored, out["reward"][envc].float(), th.zeros(Ccap, device=dev)); d_now = stored & (out["done"][envc] > 0)
            ep_ret.index_add_(0, env, r_now); ep_len.index_add_(0, env, stored.float())
            self.ep_stats["count"] += d_now.sum(); self.ep_stats["ret_sum"] += (ep_ret[env] * d_now).sum(); self.ep_stats["len_sum"] += (ep_len[env] * d_now).sum()
            keep = (~d_now).float(); ep_ret.index_copy_(0, env, ep_ret[env] * keep); ep_len.index_copy_(0, env, ep_len[env] * keep); ep_ret[N] = 0; ep_len[N] = 0
            if self.num_timesteps < self.learning_starts:
                a = th.rand(Ccap, A, device=dev) * 2 - 1                              # warm-up: uniform actions
            else:
                with th.no_grad():
                    a, _ = self.policy.action_log_prob({"observation": stage})
            prev_obs.index_copy_(0, env, stage); prev_act.index_copy_(0, env, a); has_prev.index_fill_(0, env, True); has_prev[N] = False
            slot_act.copy_(self._scale(a))
            tick += 1; self._async_ticks = tick
            if tick % 8 == 0 or tick == 1:                                            # the only host sync: every 8 ticks
                self.replay_buffer.sync_size()
                self.num_timesteps = self.replay_buffer._host_size                   # transitions stored so far
            if callback is not None and not callback.on_step():
                break
            if self.num_timesteps >= self.learning_starts and tick % self.train_freq == 0 and self.replay_buffer.size() >= self.batch_size:
                self.train(self.gradient_steps)
            if self.verbose and tick % (log_interval * 100) == 0:
                print(f"[sac async] ticks {tick} timesteps {self.num_timesteps} fps {self.num_timesteps / max(1e-9, time.time() - t0):.0f} updates {self._n_updates}")
        return self

    def learn(self, total_timesteps, callback=None, reset_num_timesteps=True, log_interval=4):
        if isinstance(callback, (list, tuple)):
            callback = CallbackList(list(callback))
        if callback is not None:
            callback.init_callback(self); callback.on_training_start(locals(), globals())
        if reset_num_timesteps:
            self.num_timesteps = 0
        if self.async_slice > 0:
            self._learn_async(total_timesteps, callback, log_interval)
            if callback is not None:
                callback.on_training_end()
            return self
        if self._last_obs is None:
            self._last_obs = self._obs_t(self.env.reset())
        t0, step = time.time(), 0
        while self.num_timesteps < total_timesteps:
            if self.num_timesteps < self.learning_starts:
                a = th.rand(self.n_envs, self.policy.action_dim, device=self.device) * 2 - 1      # warm-up: uniform actions
            else:
                with th.no_grad():
                    a, _ = self.policy.action_log_prob(self._last_obs)
            env_a = self._scale(a)
            self.replay_buffer.stage_obs(self._last_obs)            # obs_t is stored before the env may rewrite its tensors
            new_obs, rew, done, infos = self.env.step(env_a if self._tensor_env else env_a.cpu().numpy())
            new_obs = self._obs_t(new_obs)
            self.replay_buffer.add(None, self._true_next(new_obs, infos), a, _to_t(rew, self.device, th.float32), _to_t(done, self.device, th.float32), infos)
            self._last_obs = new_obs
            self.num_timesteps += self.n_envs; step += 1
            if callback is not None and not callback.on_step():
                break
            if self.num_timesteps >= self.learning_starts and step % self.train_freq == 0 and self.replay_buffer.size() >= self.batch_size:
                self.train(self.gradient_steps)
            if self.verbose and step % (log_interval * 100) == 0:
                print(f"[sac] timesteps {self.num_timesteps} fps {self.num_timesteps / max(1e-9, time.time() - t0):.0f} updates {self._n_updates}")
        if callback is not None:
            callback.on_training_end()
        return self

    def predict(self, observation, state=None, episode_start=None, deterministic=False):
        obs = observation["observation"]
        single = (obs.ndim == 3)
        o = _to_t(obs, self.device)
        if single:
            o = o.unsqueeze(0)
        with th.no_grad():
            a, _ = self.policy.action_log_prob({"observation": o}, deterministic=deterministic)
        a = self._scale(a).cpu().numpy()
        return (a[0] if single else a), state

    def save(self, path):
        if not path.endswith(".zip"):
            path = path + ".zip"
        os.makedirs(os.path.dirname(path) or ".", exist_ok=True)
        with zipfile.ZipFile(path, "w") as z:
            items = (("policy.pth", self.policy.state_dict()), ("actor.optimizer.pth", self.actor_opt.state_dict()),
                     ("critic.optimizer.pth", self.critic_opt.state_dict()),
                     ("data.pth", dict(num_timesteps=self.num_timesteps, log_ent_coef=self.log_ent_coef.detach().cpu(), gamma=self.gamma, tau=self.tau,
                                       batch_size=self.batch_size, n_updates=self._n_updates)))
            for name, obj in items:
                b = io.BytesIO(); th.save(obj, b); z.writestr(name, b.getvalue())

    @classmethod
    def load(cls, path, env=None, custom_objects=None, device=None, **kwargs):
        if not path.endswith(".zip"):
            path = path + ".zip"
        kw = dict(kwargs)
        if custom_objects and "policy_kwargs" in custom_objects:
            kw["policy_kwargs"] = custom_objects["policy_kwargs"]
        kw.setdefault("buffer_size", 1)
        with zipfile.ZipFile(path) as z:
            meta = th.load(io.BytesIO(z.read("data.pth")), weights_only=False)
            model = cls("MultiInputPolicy", env, device=device, gamma=meta["gamma"], tau=meta["tau"], batch_size=meta["batch_size"], **kw)
            model.policy.load_state_dict(th.load(io.BytesIO(z.read("policy.pth")), map_location=model.device))
            model.actor_opt.load_state_dict(th.load(io.BytesIO(z.read("actor.optimizer.pth")), map_location=model.device))
            model.critic_opt.load_state_dict(th.load(io.BytesIO(z.read("critic.optimizer.pth")), map_location=model.device))
            with th.no_grad():
                model.log_ent_coef.copy_(meta["log_ent_coef"].to(model.device))
            model.num_timesteps, model._n_updates = meta["num_timesteps"], meta["n_updates"]
        return model
