"""Bookkeeping of the asynchronous rollout (sb3/async_rollout.py) on CPU against a scripted engine that follows the
grip_batch_advance protocol: record chains, rewards landing on the right decision, GAE through next_rec, carry-over."""
import numpy as np
import torch as th

from mujoco_rl_manipulate_unknown_objects_amd.sb3.async_rollout import AsyncRollout
from mujoco_rl_manipulate_unknown_objects_amd import spaces


def duration(env, k):           # ticks a macro step takes
    return 1 + (env * 7 + k * 3) % 5


def reward_of(env, k):
    return 0.01 * env + 0.1 * k


def value_of(env, k):
    return 0.5 + 0.03 * env - 0.02 * k


HORIZON = 6


class ScriptedEngine:
    """N envs; env e's k-th macro step lasts duration(e, k) ticks, pays reward_of(e, k), ends an episode every HORIZON steps.
    Observations encode (env, k) so that the fake policy can answer value_of(env, k)."""

    def __init__(self, n, capacity, slow=1, segments=1):
        """segments > 1 mimics engine.MixedBatch: the envs are split into `segments` contiguous groups, group g lists its
        waiting envs in rows [g * capacity / segments, ...) only, so the list has holes (-1) and the count is the capacity."""
        self.slow = slow; self.segments = segments
        self.num_envs, self.action_dim, self.obs_shape, self.device = n, 6, (5, 64, 64), th.device("cpu")
        self.cap = capacity
        self.k = np.zeros(n, int)                 # macro steps finished so far
        self.left = np.zeros(n, int)              # remaining ticks of the step in flight (0 = waiting)
        self.slot = -np.ones(n, int)
        self.reward = th.zeros(n); self.done = th.zeros(n, dtype=th.uint8)
        self.started_with = {}                    # (env, k) -> action row it was started with
        self.rot = 0

    def reset(self):
        pass

    def advance(self, slot_actions, slice_len, lst, cnt):
        n = self.num_envs
        for e in range(n):
            if self.left[e] == 0 and self.slot[e] >= 0:
                self.started_with[(e, self.k[e])] = slot_actions[self.slot[e]].clone()
                self.left[e] = self.slow * duration(e, self.k[e]); self.slot[e] = -1
            if self.left[e] > 0:
                self.left[e] -= 1
                if self.left[e] == 0:
                    self.reward[e] = reward_of(e, self.k[e]); self.done[e] = 1 if (self.k[e] % HORIZON) == HORIZON - 1 else 0
                    self.k[e] += 1
        waiting = [(e - self.rot) % n for e in range(n)]
        waiting = sorted(v for v in waiting if self.left[(v + self.rot) % n] == 0)
        ids = [(v + self.rot) % n for v in waiting]
        lst.fill_(-1)
        if self.segments == 1:
            for r, e in enumerate(ids[:self.cap]):
                lst[r] = e; self.slot[e] = r
            cnt[0] = min(len(ids), self.cap)
        else:
            cg, per = self.cap // self.segments, -(-n // self.segments)
            used = [0] * self.segments
            for e in ids:
                g = e // per
                if used[g] < cg:
                    r = g * cg + used[g]; used[g] += 1
                    lst[r] = e; self.slot[e] = r
            cnt[0] = self.cap
        self.rot = (self.rot + self.cap) % n
        return {"reward": self.reward, "done": self.done}

    def observe_list(self, lst, cnt, rows):
        for r in range(int(cnt[0])):
            e = int(lst[r])
            if e < 0:
                continue
            rows[r].zero_(); rows[r, 0, 0, 0] = e; rows[r, 0, 0, 1] = self.k[e] % 256; rows[r, 0, 0, 2] = self.k[e] // 256


def fake_policy(rows):
    env = rows[:, 0, 0, 0].float(); k = rows[:, 0, 0, 1].float() + 256 * rows[:, 0, 0, 2].float()
    actions = th.stack([env, k, th.zeros_like(env), th.zeros_like(env), th.zeros_like(env), th.zeros_like(env)], 1) / 1000.0
    return actions, 0.5 + 0.03 * env - 0.02 * k, -(env + k)


def reference_gae(n_first, n_done, env, gamma, lam):
    """Plain per-env GAE over macro steps n_first .. n_done-1, bootstrapped with the value of decision n_done."""
    adv = {}
    a = 0.0
    for k in reversed(range(n_first, n_done)):
        d = 1.0 if (k % HORIZON) == HORIZON - 1 else 0.0
        delta = reward_of(env, k) + gamma * value_of(env, k + 1) * (1 - d) - value_of(env, k)
        a = delta + gamma * lam * (1 - d) * a
        adv[k] = a
    return adv


import pytest


@pytest.mark.parametrize("segments", [1, 2])
def test_records_chain_rewards_and_gae_over_two_rollouts(segments):
    n, cap, gamma, lam = 13, 4, 0.97, 0.9
    eng = ScriptedEngine(n, cap, segments=segments)
    ro = AsyncRollout(eng, fake_policy, target=40, capacity=cap, slice_len=8, gamma=gamma, gae_lambda=lam, poll_every=1)
    first = np.zeros(n, int)
    for rollout in range(3):
        got = ro.collect()
        assert got >= 40
        rows = ro.window_rows()
        comp = rows[(ro.completed[rows] & ro.is_rec[rows])].tolist()
        assert len(comp) == got
        per_env = {}
        for r in comp:
            e = int(ro.rec_env[r]); k = int(round(float(ro.actions[r, 1]) * 1000))
            assert int(round(float(ro.actions[r, 0]) * 1000)) == e           # the record holds that env's decision
            assert abs(float(ro.rewards[r]) - reward_of(e, k)) < 1e-6         # reward of step k landed on decision k
            assert float(ro.dones[r]) == (1.0 if (k % HORIZON) == HORIZON - 1 else 0.0)
            assert abs(float(ro.values[r]) - value_of(e, k)) < 1e-6
            nr = int(ro.next_rec[r]); assert nr > r and int(ro.rec_env[nr]) == e
            per_env.setdefault(e, []).append((k, r))
            # the env was started with exactly the action of this record
            assert th.allclose(eng.started_with[(e, k)], ro.actions[r])
        for e, lst in per_env.items():
            ks = [k for k, _ in sorted(lst)]
            assert ks == list(range(first[e], first[e] + len(ks)))            # no gap, no duplicate, continues the last rollout
            ref = reference_gae(ks[0], ks[-1] + 1, e, gamma, lam)
            for k, r in lst:
                assert abs(float(ro.advantages[r]) - ref[k]) < 1e-5, (e, k)
                assert abs(float(ro.returns[r]) - (ref[k] + value_of(e, k))) < 1e-5
            first[e] = ks[-1] + 1
        idx = ro.training_indices()
        assert idx.numel() == 40 and bool(ro.completed[idx].all())
    st = ro.stats()
    assert st["episodes"] > 0 and abs(st["ep_len_mean"] - HORIZON) < 1e-6


def test_ppo_learns_on_scripted_async_engine_cpu():
    """PPO.collect_rollouts + train over the async path with the real AugmentedNatureCNN policy (CPU, tiny)."""
    from mujoco_rl_manipulate_unknown_objects_amd.sb3 import PPO
    from mujoco_rl_manipulate_unknown_objects_amd.models.feature_extractor import AugmentedNatureCNN

    class Env(ScriptedEngine):
        def __init__(self):
            super().__init__(6, 3)
            self.observation_space = spaces.Dict({"observation": spaces.Box(0, 255, (5, 64, 64), np.uint8)})
            self.action_space = spaces.Box(-1.0, 1.0, (6,), np.float32)

    env = Env()
    model = PPO("MultiInputPolicy", env, n_steps=4, batch_size=12, n_epochs=1, device="cpu", async_slice=8, async_capacity=3,
                policy_kwargs=dict(features_extractor_class=AugmentedNatureCNN, share_features_extractor=True, net_arch=[32]))
    before = [p.detach().clone() for p in model.policy.parameters()]
    for _ in range(2):
        assert model.collect_rollouts()
        stats = model.train()
    assert np.isfinite(float(stats["loss"]))
    assert model.num_timesteps >= 2 * 24
    assert any(not th.equal(a, b) for a, b in zip(before, model.policy.parameters()))


# ---------------------------------------------------------------------------------------------------------------------
# N > 1: ranks whose envs finish at different speeds still run the same number of optimiser steps (exactly n_steps * N
# training records per rollout on every rank), so the per-minibatch gradient all-reduce never deadlocks and the replicas
# stay bit-identical.
def _async_worker(rank, world, port, q):
    import os
    import torch.distributed as dist
    from mujoco_rl_manipulate_unknown_objects_amd.sb3 import PPO
    from mujoco_rl_manipulate_unknown_objects_amd.models.feature_extractor import AugmentedNatureCNN
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    th.manual_seed(10 + rank)

    class Env(ScriptedEngine):
        def __init__(self):
            super().__init__(6, 3, slow=1 + 2 * rank)           # rank 1's macro steps take three times as many ticks
            self.observation_space = spaces.Dict({"observation": spaces.Box(0, 255, (5, 64, 64), np.uint8)})
            self.action_space = spaces.Box(-1.0, 1.0, (6,), np.float32)

    model = PPO("MultiInputPolicy", Env(), n_steps=3, batch_size=6, n_epochs=1, device="cpu", async_slice=8, async_capacity=3,
                policy_kwargs=dict(features_extractor_class=AugmentedNatureCNN, share_features_extractor=True, net_arch=[16]))
    ticks = []
    for _ in range(2):
        model.collect_rollouts(); ticks.append(model._async.tick); model.train()
    p = th.cat([x.detach().reshape(-1) for x in model.policy.parameters()])
    q.put((rank, p.numpy(), ticks, model.num_timesteps))
    dist.destroy_process_group()


def test_gloo_world2_async_rollouts_keep_replicas_identical():
    import socket
    import torch.multiprocessing as mp
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_async_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = {}
    for _ in range(2):
        r, p, ticks, nts = q.get(timeout=300)
        res[r] = (p, ticks, nts)
    for p in procs:
        p.join(60)
        assert p.exitcode == 0
    assert np.array_equal(res[0][0], res[1][0])                 # same parameters after two distributed updates
    assert res[0][1] != res[1][1]                               # although the ranks' rollouts took different numbers of ticks
    assert res[0][2] >= 2 * 18 and res[1][2] >= 2 * 18


def test_slice_ladder_follows_the_macro_step_length():
    """AsyncRollout.set_slice_ladder: the rung is chosen from the measured mean number of physics.step() calls per macro step,
    with 10 % hysteresis around a threshold, and a change drops the captured tick graph."""
    eng = ScriptedEngine(8, 4)
    eng.budget_us = 2000
    ro = AsyncRollout(eng, fake_policy, target=8, capacity=4, slice_len=96, gamma=0.99, gae_lambda=0.95)
    ro.set_slice_ladder()
    assert ro.ladder == ((0, 96, 2000), (215, 120, 2500), (270, 192, 4000))

    def feed(mean, n=100):
        ro._graph = "captured"
        ro.substeps_total += int(mean * n)
        ro._retune(n)
        return ro.S, eng.budget_us, ro._graph

    assert feed(180) == (96, 2000, "captured")                 # stays on the first rung, graph kept
    assert feed(225) == (96, 2000, "captured")                 # within 10 % of the 215 threshold: no move
    assert feed(250) == (120, 2500, None)                      # clearly above: second rung, graph dropped
    assert feed(262) == (120, 2500, "captured")                # within 10 % of 270
    assert feed(320) == (192, 4000, None)
    assert feed(255) == (192, 4000, "captured")                # hysteresis on the way down too
    assert feed(150) == (96, 2000, None)
    assert abs(ro.mean_substeps - 150) < 1e-9
