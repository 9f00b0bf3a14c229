"""PPO with Stable-Baselines3's surface (``PPO(policy, env, policy_kwargs=...)``, ``learn``,
``predict``, ``save`` / ``load``), built for one-process-per-GPU data parallelism.

The reference trains SAC (train_agent.py:82); PPO is what BASELINE.json's north star asks for and has
no reference hyper-parameters: SB3's PPO defaults are used (gamma 0.99, GAE lambda 0.95,
clip 0.2, vf_coef 0.5, max_grad_norm 0.5, Adam 3e-4 eps 1e-5, advantage normalisation per minibatch),
with n_steps / batch_size / n_epochs sized for thousands of envs per GPU.

MI355X layout: the rollout buffer lives in HBM (uint8 observations: n_steps x N x 20 480 B), GAE is
a reversed scan on device, minibatches are index-gathers of the buffer, and with world_size > 1 the
gradients of one minibatch are flattened into a single bucket and summed with ONE all-reduce over
RCCL/xGMI (about 4 MB: latency-bound, so one collective per optimiser step). No physics
state ever crosses GPUs.
"""
import io
import os
import time
import zipfile

import numpy as np
import torch as th
import torch.distributed as dist

from .policies import ActorCriticPolicy
from .callbacks import CallbackList


def _to_t(x, device):
    if isinstance(x, th.Tensor):
        return x.to(device)
    return th.as_tensor(np.asarray(x), device=device)


class RolloutBuffer:
    def __init__(self, n_steps, n_envs, obs_shape, action_dim, device):
        self.n_steps, self.n_envs, self.device = n_steps, n_envs, device
        self.obs = th.zeros((n_steps, n_envs) + tuple(obs_shape), dtype=th.uint8, device=device)
        self.actions = th.zeros((n_steps, n_envs, action_dim), device=device)
        z = lambda: th.zeros((n_steps, n_envs), device=device)
        self.rewards, self.values, self.log_probs, self.dones, self.advantages, self.returns = z(), z(), z(), z(), z(), z()
        self.pos = 0

    def stage_obs(self, obs):
        """Row `pos` of the observation storage, written BEFORE the env steps: an env may hand out its own scratch buffer and
        rewrite it in place (the reference's RobotEnv mutates self.obs, robot_env.py:35,223; SB3 copies for the same reason)."""
        self.obs[self.pos].copy_(obs)

    def add(self, obs, actions, rewards, dones_before, values, log_probs):
        t = self.pos
        if obs is not None:
            self.obs[t].copy_(obs)
        self.actions[t].copy_(actions); self.rewards[t].copy_(rewards)
        self.dones[t].copy_(dones_before); self.values[t].copy_(values); self.log_probs[t].copy_(log_probs)
        self.pos += 1

    def compute_returns(self, last_values, last_dones, gamma, lam):
        adv = th.zeros(self.n_envs, device=self.device)
        for t in reversed(range(self.n_steps)):
            if t == self.n_steps - 1:
                nonterm, nextv = 1.0 - last_dones, last_values
            else:
                nonterm, nextv = 1.0 - self.dones[t + 1], self.values[t + 1]
            delta = self.rewards[t] + gamma * nextv * nonterm - self.values[t]
            adv = delta + gamma * lam * nonterm * adv
            self.advantages[t] = adv
        th.add(self.advantages, self.values, out=self.returns)      # in place: the update graph holds this storage
        self.pos = 0


from .gemm_choices import recorded_gemm_choices      # the library GEMMs of a captured graph through the kernels a recorded TunableOp search picked (scoped to the capture)


class _FusedPPOLoss(th.autograd.Function):
    """The minibatch loss below and its gradients as one kernel launch (engine.ppo_loss -> grip_ppo_loss, csrc/grip_policy.hip) instead of
    ~60 elementwise / reduction launches of 4096 elements: the same arithmetic in fp32 (tests/test_gpu_env_api.py). The gradients are
    computed with the loss and handed out, scaled, by backward()."""

    @staticmethod
    def forward(ctx, mean, log_std, values, actions, old_logp, adv, ret, clip_range, ent_coef, vf_coef):
        from ..engine import ppo_loss
        out, gm, gv, gl = ppo_loss(mean.detach(), log_std.detach(), values.detach(), actions, old_logp, adv, ret, clip_range, ent_coef, vf_coef)
        ctx.save_for_backward(gm, gv, gl)
        ctx.in_dtypes = (mean.dtype, log_std.dtype, values.dtype)
        loss, pl, vl = out[0], out[1], out[2]
        ctx.mark_non_differentiable(pl, vl)
        return loss, pl, vl

    @staticmethod
    def backward(ctx, g, _gpl, _gvl):
        gm, gv, gl = ctx.saved_tensors
        dm, dl, dv = ctx.in_dtypes
        return (gm * g).to(dm), (gl * g).to(dl), (gv * g).to(dv), None, None, None, None, None, None, None


class PPO:
    def __init__(self, policy, env, learning_rate=3e-4, n_steps=16, batch_size=4096, n_epochs=4, gamma=0.99, gae_lambda=0.95,
                 clip_range=0.2, ent_coef=0.0, vf_coef=0.5, max_grad_norm=0.5, policy_kwargs=None, verbose=0, tensorboard_log=None,
                 device=None, seed=None, autocast_dtype=None, async_slice=0, async_capacity=None, async_budget_us=0, overlap_update=False,
                 async_auto_slice=False, miopen_find=True):
        """async_slice > 0 switches rollout collection to the time-sliced engine (sb3/async_rollout.py): every tick gives
        each env at most `async_slice` calls of physics.step(), at most `async_capacity` finished envs (default N/4) are
        rendered and decided per tick, and a rollout is n_steps * N completed transitions whichever envs they come from.
        async_budget_us > 0 also caps a wavefront's slice by wall-clock time (include/grip_sim.h); async_auto_slice lets slice
        and budget follow the measured length of the macro steps (AsyncRollout.set_slice_ladder).
        overlap_update (async, GPU only): the update of rollout i runs on a second stream WHILE rollout i + 1 is collected,
        with a frozen copy of the policy (the parameters after update i - 1: one update of policy lag, as in asynchronous
        PPO variants); the records of the two rollouts live in AsyncRollout's two windows."""
        self.env = env
        self.n_envs = getattr(env, "num_envs", 1)
        self.device = th.device(device) if device is not None else getattr(env, "device", th.device("cuda" if th.cuda.is_available() else "cpu"))
        self.n_steps, self.batch_size, self.n_epochs = n_steps, batch_size, n_epochs
        self.gamma, self.gae_lambda, self.clip_range = gamma, gae_lambda, clip_range
        self.ent_coef, self.vf_coef, self.max_grad_norm = ent_coef, vf_coef, max_grad_norm
        self.verbose, self.tensorboard_log = verbose, tensorboard_log
        self.policy_kwargs = dict(policy_kwargs or {})
        self.autocast_dtype = autocast_dtype
        if seed is not None:
            th.manual_seed(seed)
        policy_class = ActorCriticPolicy if isinstance(policy, str) else policy
        self.policy = policy_class(env.observation_space, env.action_space, **self.policy_kwargs).to(self.device)
        if self.device.type == "cuda":
            self.policy = self.policy.to(memory_format=th.channels_last)
        self.distributed = dist.is_available() and dist.is_initialized() and dist.get_world_size() > 1
        if self.distributed:                      # identical initial parameters on every rank
            for p in self.policy.parameters():
                dist.broadcast(p.data, src=0)
        # on a GPU the minibatch update is captured into hipGraphs (forward+backward | gradient all-reduce | clip+Adam), which
        # needs the optimiser's step counter on the device
        self.graph_update = self.device.type == "cuda"
        self.fused_loss = self.device.type == "cuda"        # the minibatch loss and its gradients as one launch (_FusedPPOLoss)
        # MIOpen's find mode: the first (eager) minibatches time the library's convolution kernels per shape and keep the fastest
        # (the heuristic choice for the 4 x 4 stride-2 data gradient runs at a third of the rate): update 37.9 -> 35.1 ms per 16 minibatches.
        # The process-global switch is on only around those minibatches and the capture (_find_mode), not for the life of the process.
        self.miopen_find = bool(self.device.type == "cuda" and miopen_find)
        # on a GPU: the fused multi-tensor Adam (one kernel per step; the foreach form spends ~45 small launches per step on
        # dividing every parameter-shaped tensor by 0-dim bias corrections), capturable so that it can sit in the update graph
        self.optimizer = th.optim.Adam(self.policy.parameters(), lr=learning_rate, eps=1e-5, capturable=self.graph_update,
                                       fused=True if self.graph_update else None)
        self._upd = None
        # The explicit update sequence (sb3/fused_update.py) keeps the parameters in one flat buffer. They are laid out HERE, before anything can hold their
        # addresses -- the rollout copy below, and above all the captured rollout tick, which reads biases straight from the parameters: laid out at the first
        # update instead, the tick graph kept reading the freed old storage (NaN losses from the second rollout on; tools/train_probe.py found it).
        self._fused = None
        if self.explicit_update and self.device.type == "cuda":
            from .fused_update import FusedUpdate
            if FusedUpdate.applies(self):
                self._fused = FusedUpdate(self)
        obs_shape = env.observation_space["observation"].shape
        self.rollout_buffer = None if (async_slice and async_slice > 0) else RolloutBuffer(n_steps, self.n_envs, obs_shape, self.policy.action_dim, self.device)
        self._async = None
        self.overlap_update = bool(overlap_update) and bool(async_slice and async_slice > 0) and self.device.type == "cuda"
        self.policy_rollout = self.policy
        if self.overlap_update:
            import copy
            self.policy_rollout = copy.deepcopy(self.policy).requires_grad_(False)
            self._upd_stream = th.cuda.Stream(self.device)
            self._ev_copy = None
        if async_slice and async_slice > 0:
            from .async_rollout import AsyncRollout, BatchEngineAdapter
            eng = env if hasattr(env, "advance") else BatchEngineAdapter(env, async_budget_us)
            cap = int(async_capacity) if async_capacity else max(1, self.n_envs // 4)

            def policy_fn(obs_rows):
                with th.no_grad(), self._ac():
                    return self.policy_rollout({"observation": obs_rows})
            def policy_parts_fn(obs_rows):
                with th.no_grad(), self._ac():
                    return self.policy_rollout.forward_parts({"observation": obs_rows})
            # (under autocast the extractor's in-place fast path steps aside -- rollout_features() returns None -- and the fallback wants real
            # tensors: the tick then stages the observation rows as before)
            policy_parts_fn.accepts_record_rows = lambda: self.autocast_dtype is None and bool(getattr(self.policy_rollout, "accepts_record_rows", lambda: False)())
            parts = policy_parts_fn if hasattr(self.policy, "forward_parts") else None
            self._async = AsyncRollout(eng, policy_fn, policy_parts_fn=parts, target=n_steps * self.n_envs, capacity=min(cap, self.n_envs), slice_len=async_slice,
                                       gamma=gamma, gae_lambda=gae_lambda, action_low=env.action_space.low, action_high=env.action_space.high)
            self.rollout_buffer = None
            if self.device.type == "cuda" and hasattr(self.policy_rollout, "enable_rollout_cache"):
                self.policy_rollout.enable_rollout_cache()       # merged policy / value heads for the no-grad rollout forward
            if async_auto_slice:
                self._async.set_slice_ladder()
        self.num_timesteps = 0
        self._last_obs = None
        self._last_dones = None
        if self._fused is None:
            self._flat_grad = None
        self.logger = {}

    # ------------------------------------------------------------------ rollouts
    def _obs_t(self, obs):
        return {"observation": _to_t(obs["observation"], self.device)}

    def _ac(self):
        if self.autocast_dtype is not None and self.device.type == "cuda":
            # cache_enabled=False: the rollout tick and the minibatch update are captured into hipGraphs, and autocast's weight-cast
            # cache must not outlive a capture (the cached bf16 copies live in the graph's private pool: replays then read stale or
            # recycled memory -- NaN losses within a few updates)
            return th.autocast("cuda", dtype=self.autocast_dtype, cache_enabled=False)
        return th.autocast("cpu", enabled=False)

    def collect_rollouts(self, callback=None):
        if self._async is not None:
            if self.overlap_update and self._ev_copy is not None:
                # rollout weights of this rollout are in place; in stream order this also means the update before the last
                # one is done, i.e. the record window this rollout is about to overwrite is free
                th.cuda.current_stream(self.device).wait_event(self._ev_copy)
            if getattr(self.policy_rollout, "_rollout_cache", None) is not None:
                self.policy_rollout.refresh_rollout_cache()          # merged-head weights of the rollout forward follow the update
            state = {"n": 0, "ok": True}

            def on_poll(done_n):
                self.num_timesteps += done_n - state["n"]; state["n"] = done_n
                if callback is not None and not callback.on_step():
                    state["ok"] = False
                    return False
                return True
            done_n = self._async.collect(on_poll)
            self.num_timesteps += done_n - state["n"]
            return state["ok"]
        buf = self.rollout_buffer
        if self._last_obs is None:
            self._last_obs = self._obs_t(self.env.reset())
            self._last_dones = th.zeros(self.n_envs, device=self.device)
        low = th.as_tensor(self.env.action_space.low, device=self.device); high = th.as_tensor(self.env.action_space.high, device=self.device)
        for _ in range(self.n_steps):
            with th.no_grad(), self._ac():
                actions, values, log_probs = self.policy(self._last_obs)
            clipped = th.max(th.min(actions, high), low)
            env_actions = clipped if isinstance(self._last_obs["observation"], th.Tensor) and hasattr(self.env, "device") else clipped.cpu().numpy()
            buf.stage_obs(self._last_obs["observation"])            # obs_t goes into the buffer before the env may rewrite its tensors
            new_obs, rewards, dones, infos = self.env.step(env_actions)
            buf.add(None, actions, _to_t(rewards, self.device).float(), self._last_dones, values, log_probs)
            self._last_obs = self._obs_t(new_obs)
            self._last_dones = _to_t(dones, self.device).float()
            self.num_timesteps += self.n_envs
            if callback is not None and not callback.on_step():
                return False
        with th.no_grad(), self._ac():
            last_values = self.policy.predict_values(self._last_obs)
        buf.compute_returns(last_values, self._last_dones, self.gamma, self.gae_lambda)
        return True

    # ------------------------------------------------------------------ update
    def _bind_flat_grads(self):
        """One flat fp32 bucket IS the gradient storage: every parameter's .grad is a view into it (layout = parameter order), so the
        all-reduce of a minibatch is one collective on memory the backward pass wrote in place -- no concatenation before it, no
        copy-back after it, and the addresses the captured update graphs hold never change. (With the explicit update path the bucket is
        FusedUpdate's gradient buffer, in ITS layout.)"""
        if getattr(self, "_fused", None) is not None:
            self._fused.bind()
            return
        params = list(self.policy.parameters())
        n = sum(p.numel() for p in params)
        if self._flat_grad is None or self._flat_grad.numel() != n:
            self._flat_grad = th.zeros(n, dtype=th.float32, device=self.device)
        off = 0
        for p in params:
            k = p.numel()
            # same strides as the parameter (conv weights are channels_last on the GPU: the fused optimiser wants grad and param alike)
            view = th.as_strided(self._flat_grad, p.size(), p.stride(), storage_offset=off)
            if p.grad is None or p.grad.data_ptr() != view.data_ptr() or p.grad.stride() != view.stride():
                p.grad = view
            off += k

    def _zero_grads(self):
        if self.distributed:
            self._bind_flat_grads(); self._flat_grad.zero_()
        else:
            self.optimizer.zero_grad(set_to_none=False)

    def allreduce_ms(self):
        """mean device time of the gradient all-reduce over the last (up to 64) optimiser steps, and how many were timed; (None, 0) when none"""
        ev = getattr(self, "_ar_events", None)
        if not ev or ev["n"] == 0:
            return None, 0
        k = min(ev["n"], 64)
        th.cuda.synchronize(self.device)
        return sum(a.elapsed_time(b) for a, b in ev["ring"][:k]) / k, ev["n"]

    def _allreduce_grads(self):
        """Sum the ranks' minibatch gradients (about 4 MB of fp32: one collective per optimiser step, latency-bound on xGMI) and
        average. The clip + Adam step needs the reduced gradient and the next minibatch's forward needs the stepped parameters, so
        there is nothing of this replica's own to overlap the collective with short of applying stale gradients."""
        if self._flat_grad is None or any(p.grad is None or p.grad.untyped_storage().data_ptr() != self._flat_grad.untyped_storage().data_ptr()
                                          for p in self.policy.parameters()):
            # gradients produced outside the bucket (a caller ran backward before the first update): move them in once
            old = [None if p.grad is None else p.grad.detach().clone() for p in self.policy.parameters()]
            for p in self.policy.parameters():
                p.grad = None
            self._bind_flat_grads(); self._flat_grad.zero_()
            for p, g in zip(self.policy.parameters(), old):
                if g is not None:
                    p.grad.copy_(g)
        timed = self._flat_grad.is_cuda and not th.cuda.is_current_stream_capturing()
        if timed:                                   # device time of the collective, read back in allreduce_ms() (no sync here)
            ev = getattr(self, "_ar_events", None)
            if ev is None:
                ev = self._ar_events = {"ring": [(th.cuda.Event(enable_timing=True), th.cuda.Event(enable_timing=True)) for _ in range(64)], "n": 0}
            e0, e1 = ev["ring"][ev["n"] % 64]
            e0.record()
        dist.all_reduce(self._flat_grad, op=dist.ReduceOp.SUM)
        if timed:
            e1.record(); ev["n"] += 1
        self._flat_grad.mul_(1.0 / dist.get_world_size())

    def _loss_backward(self, src, idx):
        """Forward + PPO loss + backward of one minibatch `idx` (record / sample ids into the rollout storage `src`)."""
        obs, actions, old_logp, adv_all, ret_all = src
        fu = getattr(self, "_fused", None)
        if fu is not None:
            return fu.loss_backward(src, idx)
        if self.fused_loss and obs.is_cuda and hasattr(self.policy, "forward_parts"):
            if self.indexed_minibatches and obs.dtype == th.uint8 and obs.dim() == 4 and obs.is_contiguous() and idx.dtype == th.int64:
                from ..engine import IndexedRows              # the minibatch as row numbers: the first layer and its weight gradient read the rows where they lie
                ob = IndexedRows(obs, idx)
            else:
                ob = obs[idx]
            with self._ac():
                mean, log_std, values = self.policy.forward_parts({"observation": ob})
            loss, pl, vl = _FusedPPOLoss.apply(mean, log_std, values, actions[idx], old_logp[idx], adv_all[idx], ret_all[idx],
                                               self.clip_range, self.ent_coef, self.vf_coef)
            loss.backward()
            return pl, vl, loss.detach()
        adv = adv_all[idx]
        adv = (adv - adv.mean()) / (adv.std() + 1e-8)
        with self._ac():
            values, logp, entropy = self.policy.evaluate_actions({"observation": obs[idx]}, actions[idx])
        # the log-ratio is clamped before exp(): beyond +-20 the clipped surrogate is flat anyway, and an overflowing ratio times a zero
        # advantage is the NaN that ends a run (seen with the bf16 policy, whose rollout and update forwards round differently)
        ratio = th.exp(th.clamp(logp - old_logp[idx], -20.0, 20.0))
        pl = -th.min(adv * ratio, adv * th.clamp(ratio, 1 - self.clip_range, 1 + self.clip_range)).mean()
        vl = th.nn.functional.mse_loss(ret_all[idx], values)
        el = -entropy.mean()
        loss = pl + self.ent_coef * el + self.vf_coef * vl
        loss.backward()
        return pl.detach(), vl.detach(), loss.detach()

    indexed_minibatches = True       # minibatch observations handed to the policy as engine.IndexedRows (no gathered copy) on the fused-loss path
    explicit_update = True           # forward / loss / backward of a minibatch as an explicit launch sequence over flat parameter and gradient buffers
                                     # (sb3/fused_update.py) where the policy is the default one; False: autograd over the same kernels
    _fused = None

    def _select_update_path(self, src, idx):
        """Decide (once per storage) whether the explicit launch sequence applies; it re-lays the parameters when it is first taken."""
        fu = self._fused
        obs, actions, old_logp, adv_all, ret_all = src
        fits = (self.explicit_update and obs.is_cuda and obs.dtype == th.uint8 and obs.dim() == 4 and obs.is_contiguous() and idx.dtype == th.int64 and idx.dim() == 1
                and actions.dim() == 2 and all(t.is_cuda and t.dtype == th.float32 and t.is_contiguous() for t in (actions, old_logp, adv_all, ret_all))
                and all(t.dim() == 1 and t.numel() == actions.shape[0] for t in (old_logp, adv_all, ret_all)) and idx.numel() >= 2)
        if fu is not None and (not fits or not fu.intact()):
            self._fused = fu = None                      # somebody moved the parameters (they are laid out again below), or another kind of storage (autograd)
            self._flat_grad = None
            for p in self.policy.parameters():
                p.grad = None
            self._upd = None
        if fu is None and fits and not getattr(self, "_fused_declined", False):
            from .fused_update import FusedUpdate
            if FusedUpdate.applies(self):
                self._fused = FusedUpdate(self)
                self._upd = None
                self._parameters_moved()
            else:
                self._fused_declined = True

    def _parameters_moved(self):
        """the parameters have new addresses: whatever captured or cached the old ones must let go -- the rollout's tick graphs (they read biases straight
        from the parameters) and the merged rollout weights"""
        ar = self._async
        if ar is not None:
            ar._graph = None
            if getattr(ar, "_side_graph", None) is not None:
                ar._side_graph = [None, None]
        if self.policy_rollout is self.policy and getattr(self.policy_rollout, "_rollout_cache", None) is not None:
            self.policy_rollout.refresh_rollout_cache()

    def _fresh_grads(self, to_none):
        """what a backward pass needs of .grad beforehand"""
        if self._fused is not None:
            self._fused.bind()                           # every slot is overwritten: nothing to zero
        elif self.distributed:
            self._zero_grads()
        else:
            self.optimizer.zero_grad(set_to_none=to_none)
    fused_clip_adam = True           # clipping + Adam in two launches (engine.ClipAdam) where the optimiser is a plain torch Adam on float32 CUDA tensors

    def _apply(self):
        if self.fused_clip_adam and self.device.type == "cuda":
            ca = getattr(self, "_clip_adam", None)
            g0 = self.optimizer.param_groups[0] if self.optimizer.param_groups else {}
            hyper = (g0.get("lr"), tuple(g0.get("betas", ())), g0.get("eps"))
            if ca is None or ca.opt is not self.optimizer or ca.max_norm != float(self.max_grad_norm) or getattr(self, "_clip_adam_hyper", hyper) != hyper:
                from ..engine import ClipAdam
                # a captured update graph has the old object's scratch address and the optimiser's hyper-parameters baked into its launches: capture again
                if ca is not None and getattr(self, "_upd", None) is not None:
                    self._upd = None
                ca = self._clip_adam = ClipAdam(self.optimizer, self.max_grad_norm)
            self._clip_adam_hyper = hyper
            if all(p.grad is not None for p in self.policy.parameters()) and ca.step():
                return
        th.nn.utils.clip_grad_norm_(self.policy.parameters(), self.max_grad_norm)
        self.optimizer.step()

    def _minibatch_update(self, src, idx):
        """One optimiser step. Eager on CPU; on a GPU two captured graphs with the (eager) gradient all-reduce between them."""
        if self.graph_update and (self._upd is None or self._upd.get("fwd") is None):
            # the eager warm-up steps and the capture: MIOpen's find mode and the recorded GEMM choices are on for these only (both are process-global switches)
            prev = th.backends.cudnn.benchmark
            th.backends.cudnn.benchmark = prev or self.miopen_find
            try:
                with recorded_gemm_choices():
                    return self._minibatch_update_impl(src, idx)
            finally:
                th.backends.cudnn.benchmark = prev
        return self._minibatch_update_impl(src, idx)

    def _minibatch_update_impl(self, src, idx):
        u = self._upd
        key = tuple(t.data_ptr() for t in src) + (idx.numel(),)
        if not self.graph_update or u is None or u["key"] != key or u["fwd"] is None:
            self._select_update_path(src, idx)          # (not per replay: the checks are ~50 us of host time, and the update is not far from launch-bound)
        if not self.graph_update:
            if self._fused is not None:
                self._fused.bind()
            else:
                self._zero_grads()
            out = self._loss_backward(src, idx)
            if self.distributed:
                self._allreduce_grads()
            self._apply()
            return out
        u = self._upd                                    # (_select_update_path may have dropped it)
        if u is None or u["key"] != key:
            u = self._upd = {"key": key, "idx": th.zeros_like(idx), "warm": 0, "fwd": None, "apply": None, "out": None}
        u["idx"].copy_(idx)
        if u["warm"] < 2:                           # two eager steps on a side stream before capturing (PyTorch's capture recipe)
            cur = th.cuda.current_stream(self.device); side = th.cuda.Stream(self.device); side.wait_stream(cur)
            with th.cuda.stream(side):
                self._fresh_grads(to_none=True)
                out = self._loss_backward(src, u["idx"])
                if self.distributed:
                    self._allreduce_grads()
                self._apply()
            cur.wait_stream(side)
            u["warm"] += 1
            return out
        if u["fwd"] is None:
            th.cuda.synchronize(self.device)
            try:
                # distributed: .grad = views of the flat bucket, the captured backward accumulates in place; single process: the captured backward
                # allocates .grad from the graph's pool; explicit path: the gradient buffer's slots are overwritten
                self._fresh_grads(to_none=True)
                g1 = th.cuda.CUDAGraph()
                with th.cuda.graph(g1, capture_error_mode="thread_local"):
                    u["out"] = self._loss_backward(src, u["idx"])
                g2 = th.cuda.CUDAGraph()
                with th.cuda.graph(g2, capture_error_mode="thread_local"):
                    self._apply()
                    if self._fused is not None:
                        pass
                    elif self.distributed:
                        self._flat_grad.zero_()
                    else:
                        self.optimizer.zero_grad(set_to_none=False)
                u["fwd"], u["apply"] = g1, g2
            except Exception as ex:                               # noqa: BLE001 -- fall back to the eager update
                import warnings
                warnings.warn(f"hipGraph capture of the PPO update failed ({ex}); continuing eagerly")
                th.cuda.synchronize(self.device)
                self.graph_update = False; self._upd = None
                return self._minibatch_update_impl(src, idx)
        u["fwd"].replay()
        if self.distributed:
            self._allreduce_grads()
        u["apply"].replay()
        return u["out"]

    def train(self):
        total = self.n_steps * self.n_envs
        if self._async is not None:                      # records of the async rollout, addressed through `sel`
            buf = self._async
            src = (buf.obs, buf.actions, buf.log_probs, buf.advantages, buf.returns)
            sel = buf.training_indices()
        else:
            buf = self.rollout_buffer
            T = self.n_steps                      # may be shorter than the buffer (a caller trimming its last rollout)
            src = (buf.obs[:T].view((total,) + buf.obs.shape[2:]), buf.actions[:T].view(total, -1), buf.log_probs[:T].view(-1),
                   buf.advantages[:T].view(-1), buf.returns[:T].view(-1))
            sel = None
        if not self.overlap_update:
            return self._train_on(src, sel, total)
        # overlapped: on the update stream, after the rollout's data is complete -- first hand the current parameters to the
        # rollout copy (nobody is ticking: the next collect_rollouts waits for that copy), then update while it collects
        main = th.cuda.current_stream(self.device)
        ready = th.cuda.Event(); ready.record(main)
        if sel is not None:
            sel.record_stream(self._upd_stream)          # allocated on this stream, read by the other: keep the allocator off it
            self._sel_in_flight = sel
        with th.cuda.stream(self._upd_stream):
            self._upd_stream.wait_event(ready)
            with th.no_grad():
                th._foreach_copy_(list(self.policy_rollout.parameters()), list(self.policy.parameters()))
            ev = th.cuda.Event(); ev.record(self._upd_stream); self._ev_copy = ev
            stats = self._train_on(src, sel, total)
        return stats

    def finish_updates(self):
        """Wait (on the current stream) for an overlapped update still in flight."""
        if self.overlap_update:
            done = th.cuda.Event(); done.record(self._upd_stream)
            th.cuda.current_stream(self.device).wait_event(done)

    def _train_on(self, src, sel, total):
        bs = min(self.batch_size, total)
        stats = {}
        # once per train() (~50 us of host time), not per replay: are the parameters still where the captured update reads and writes them? A caller who moved or
        # re-assigned one (policy.to(), p.data = ...) would otherwise train the old flat buffer while the live module stands still.
        if self._fused is not None and self._upd is not None and not self._fused.intact():
            self._fused = None; self._flat_grad = None; self._upd = None
            for p in self.policy.parameters():
                p.grad = None
            self._parameters_moved()
        for _ in range(self.n_epochs):
            perm = th.randperm(total, device=self.device)
            if sel is not None:
                perm = sel[perm]
            for s in range(0, total - bs + 1, bs):
                pl, vl, loss = self._minibatch_update(src, perm[s:s + bs])
                stats = {"policy_loss": pl, "value_loss": vl, "loss": loss}
        # the captured update writes the parameters by graph replay, through raw pointers: autograd's version counters did not see it. Bump them,
        # so that everything cached from the parameters (the policy's merged rollout weights) knows it is stale -- collect_rollouts() refreshes
        # explicitly, an evaluation straight after train() relies on this.
        if self.device.type == "cuda":
            from ..engine import ClipAdam
            ClipAdam.bump_versions(list(self.policy.parameters()))
        self.logger = stats
        return stats

    # ------------------------------------------------------------------ SB3 surface
    def learn(self, total_timesteps, callback=None, reset_num_timesteps=True, log_interval=1):
        if isinstance(callback, (list, tuple)):
            callback = CallbackList(list(callback))
        if callback is not None:
            callback.init_callback(self)
            callback.on_training_start(locals(), globals())
        if reset_num_timesteps:
            self.num_timesteps = 0
        it, t0 = 0, time.time()
        while self.num_timesteps < total_timesteps:
            ok = self.collect_rollouts(callback)
            # every rank must leave this loop after the same iteration: train() holds one all-reduce per minibatch, and a rank that
            # went on alone would wait in it for ever. The time-sliced rollout counts polled completions, which overshoot the target
            # by a rank-dependent amount, and a callback may stop one rank only: agree on both flags (MAX) once per iteration.
            stop_now, last = (not ok), self.num_timesteps >= total_timesteps
            if self.distributed:                            # (single process: plain Python flags, no device tensor and no host sync)
                stop = th.tensor([1.0 if stop_now else 0.0, 1.0 if last else 0.0], device=self.device)
                dist.all_reduce(stop, op=dist.ReduceOp.MAX)
                stop_now, last = bool(stop[0] > 0), bool(stop[1] > 0)
            if stop_now:
                break
            self.train()
            it += 1
            if last:                                        # another rank reached the target: leave with it; num_timesteps keeps what THIS rank collected
                break
            if self.verbose and it % log_interval == 0:
                fps = self.num_timesteps / max(1e-9, time.time() - t0)
                print(f"[ppo] iter {it} timesteps {self.num_timesteps} fps {fps:.0f} loss {float(self.logger.get('loss', 0)):.4f}")
        self.finish_updates()
        if callback is not None:
            callback.on_training_end()
        return self

    def predict(self, observation, state=None, episode_start=None, deterministic=False):
        obs = observation["observation"]
        single = (obs.ndim == 3)
        o = _to_t(obs, self.device)
        if single:
            o = o.unsqueeze(0)
        with th.no_grad(), self._ac():
            actions, _, _ = self.policy({"observation": o}, deterministic=deterministic)
        low = th.as_tensor(self.env.action_space.low, device=self.device); high = th.as_tensor(self.env.action_space.high, device=self.device)
        a = th.max(th.min(actions, high), low).cpu().numpy()
        return (a[0] if single else a), state

    def save(self, path):
        self.finish_updates()
        if not path.endswith(".zip"):
            path = path + ".zip"
        os.makedirs(os.path.dirname(path) or ".", exist_ok=True)
        with zipfile.ZipFile(path, "w") as z:
            for name, obj in (("policy.pth", self.policy.state_dict()), ("policy.optimizer.pth", self.optimizer.state_dict())):
                b = io.BytesIO(); th.save(obj, b); z.writestr(name, b.getvalue())
            meta = dict(num_timesteps=self.num_timesteps, n_steps=self.n_steps, batch_size=self.batch_size, n_epochs=self.n_epochs,
                        gamma=self.gamma, gae_lambda=self.gae_lambda, clip_range=self.clip_range)
            b = io.BytesIO(); th.save(meta, b); z.writestr("data.pth", b.getvalue())

    @classmethod
    def load(cls, path, env=None, custom_objects=None, device=None, **kwargs):
        if not path.endswith(".zip"):
            path = path + ".zip"
        kw = dict(kwargs)
        if custom_objects and "policy_kwargs" in custom_objects:
            kw["policy_kwargs"] = custom_objects["policy_kwargs"]
        with zipfile.ZipFile(path) as z:
            meta = th.load(io.BytesIO(z.read("data.pth")), weights_only=False)
            for k in ("n_steps", "batch_size", "n_epochs", "gamma", "gae_lambda", "clip_range"):
                kw.setdefault(k, meta[k])
            model = cls("MultiInputPolicy", env, device=device, **kw)
            model.policy.load_state_dict(th.load(io.BytesIO(z.read("policy.pth")), map_location=model.device))
            model.optimizer.load_state_dict(th.load(io.BytesIO(z.read("policy.optimizer.pth")), map_location=model.device))
            model.num_timesteps = meta["num_timesteps"]
        return model
