"""SAC + replay / HER buffers (sb3/sac.py) on CPU: the reference's own call shapes (train_agent.py:58-92)."""
import numpy as np
import torch

from mujoco_rl_manipulate_unknown_objects_amd import spaces
from mujoco_rl_manipulate_unknown_objects_amd.sb3 import SAC, HerReplayBuffer, ReplayBuffer
from mujoco_rl_manipulate_unknown_objects_amd.models.feature_extractor import AugmentedNatureCNN

KW = dict(features_extractor_class=AugmentedNatureCNN, share_features_extractor=True, net_arch=[32, 32])


class GoalVec:
    """4 envs, episodes of 5 steps; achieved goal walks along x; reward = progress + e^-|dg - ag| (her_buffer form)."""
    observation_space = spaces.Dict({"observation": spaces.Box(0, 255, shape=(5, 64, 64), dtype=np.uint8),
                                     "achieved_goal": spaces.Box(-10, 10, shape=(2,), dtype=np.float32),
                                     "desired_goal": spaces.Box(-10, 10, shape=(2,), dtype=np.float32)})
    action_space = spaces.Box(-1., 1., shape=(6,), dtype=np.float32)

    def __init__(self, n=4, seed=0):
        self.num_envs = n; self.rng = np.random.default_rng(seed); self.t = np.zeros(n, int); self.x = np.zeros(n)

    def _obs(self):
        ag = np.stack([self.x, np.zeros(self.num_envs)], 1).astype(np.float32)
        return {"observation": self.rng.integers(0, 255, (self.num_envs, 5, 64, 64), dtype=np.uint8), "achieved_goal": ag,
                "desired_goal": np.tile(np.array([[1.0, 0.0]], np.float32), (self.num_envs, 1))}

    def reset(self):
        self.t[:] = 0; self.x[:] = 0
        return self._obs()

    def step(self, a):
        self.x += 0.1 * (1 + np.arange(self.num_envs)); self.t += 1
        ag = np.stack([self.x, np.zeros(self.num_envs)], 1)
        rew = 0.25 + np.exp(-np.linalg.norm(np.array([1.0, 0.0]) - ag, axis=1))
        done = self.t == 5
        obs = self._obs()
        for e in np.nonzero(done)[0]:
            self.t[e] = 0; self.x[e] = 0
        o2 = self._obs(); obs["achieved_goal"][done] = o2["achieved_goal"][done]      # auto-reset obs for finished envs
        return obs, rew.astype(np.float32), done, [{}] * self.num_envs


def test_sac_learns_saves_loads(tmp_path):
    env = GoalVec()
    m = SAC("MultiInputPolicy", env, policy_kwargs=KW, buffer_size=400, batch_size=16, learning_starts=8, device="cpu", seed=0, verbose=0,
            tensorboard_log=str(tmp_path))
    tgt0 = [p.detach().clone() for p in m.policy.q_targets.parameters()]
    fe0 = [p.detach().clone() for p in m.policy.actor_features.parameters()]
    m.learn(80)
    assert m.num_timesteps == 80 and m._n_updates > 0
    for k in ("actor_loss", "critic_loss"):
        assert np.isfinite(float(m.logger[k]))
    assert any(not torch.equal(a, b) for a, b in zip(tgt0, m.policy.q_targets.parameters()))          # polyak moved the targets
    assert any(not torch.equal(a, b) for a, b in zip(fe0, m.policy.actor_features.parameters()))       # critic loss trains the shared extractor
    assert float(m.log_ent_coef.detach().exp()) != 1.0                                                          # entropy coefficient is tuned
    m.save(str(tmp_path / "best_model"))
    m2 = SAC.load(str(tmp_path / "best_model"), env=env, device="cpu", custom_objects={"policy_kwargs": KW})
    o = env.reset()
    a1, _ = m.predict(o, deterministic=True); a2, _ = m2.predict(o, deterministic=True)
    assert np.array_equal(a1, a2) and a1.shape == (4, 6) and np.abs(a1).max() <= 1.0


def test_her_future_relabelling():
    env = GoalVec()
    buf = HerReplayBuffer(200, env.observation_space, env.action_space, torch.device("cpu"), n_envs=4, n_sampled_goal=4,
                          goal_selection_strategy="future", online_sampling=True, max_episode_length=5)
    obs = {k: torch.as_tensor(v) for k, v in env.reset().items()}
    for t in range(23):                                   # four full episodes + three open steps
        no, r, d, _ = env.step(None)
        no = {k: torch.as_tensor(v) for k, v in no.items()}
        # the stored next achieved goal of a finished step is the terminal one, not the reset observation
        term = no["achieved_goal"].clone()
        term[torch.as_tensor(d)] = torch.tensor([[0.1 * (1 + e) * 5, 0.0] for e in range(4)])[torch.as_tensor(d)]
        buf.add(obs, {**no, "achieved_goal": term}, torch.zeros(4, 6), torch.as_tensor(r), torch.as_tensor(d).float())
        obs = no
    assert buf.size() == 23 * 4
    g = torch.Generator().manual_seed(1)
    b = buf.sample(4000, generator=g)
    rel = b["relabelled"]
    assert abs(float(rel.float().mean()) - 0.8 * 20 / 23) < 0.03            # her_ratio 4/5 of the rows whose episode is closed
    ag_next = b["next_obs"]["achieved_goal"]; dg = b["obs"]["desired_goal"]
    # relabelled goals are achieved goals of the same env at this or a later step of the episode: x' >= x, multiple of the env's stride
    assert bool((dg[rel][:, 0] >= ag_next[rel][:, 0] - 1e-6).all()) and bool((dg[rel][:, 1] == 0).all())
    assert bool((dg[~rel] == torch.tensor([1.0, 0.0])).all())
    want = 0.25 + torch.exp(-torch.linalg.norm(dg - ag_next, dim=1))
    assert torch.allclose(b["rewards"], want, atol=1e-5)
    # plain buffer: same storage, no goals touched
    pb = ReplayBuffer(200, env.observation_space, env.action_space, torch.device("cpu"), n_envs=4)
    pb.add(obs, obs, torch.zeros(4, 6), torch.ones(4), torch.zeros(4))
    assert pb.sample(8)["rewards"].shape == (8,)


def test_flat_replay_ring_and_sac_over_a_scripted_time_sliced_engine():
    """n1 on the time-sliced schedule: the flat ring scatters varying batches behind a device-side pointer and wraps; SAC.learn over a
    scripted asynchronous engine (envs finish on their own clocks) stores, for every env, the chain obs_t -> next_obs == obs_{t+1}."""
    import torch as th
    from mujoco_rl_manipulate_unknown_objects_amd import spaces
    from mujoco_rl_manipulate_unknown_objects_amd.sb3 import SAC, FlatReplayBuffer
    osp = spaces.Dict({"observation": spaces.Box(0, 255, shape=(5, 64, 64), dtype=np.uint8)}); asp = spaces.Box(-1., 1., shape=(6,), dtype=np.float32)
    rb = FlatReplayBuffer(10, osp, asp, th.device("cpu"))
    for k in range(4):
        mask = th.tensor([True, False, True, True]); val = th.arange(4).float() + 10 * k
        rb.add_rows(mask, th.zeros(4, 5, 64, 64, dtype=th.uint8), th.ones(4, 5, 64, 64, dtype=th.uint8), th.zeros(4, 6), val, th.zeros(4))
    assert rb.sync_size() == 10 and int(rb.ptr) == 12
    # rows 10, 11 wrapped onto slots 0, 1: rewards of batch 3 are 30 (row 9), 32 (row 10 -> slot 0), 33 (row 11 -> slot 1)
    assert rb.rewards[:10].tolist() == [32.0, 33.0, 3.0, 10.0, 12.0, 13.0, 20.0, 22.0, 23.0, 30.0]

    class Eng:                                             # env e finishes a macro step every (1 + e % 3) ticks; obs of env e at its k-th decision = 40 e + k
        num_envs, action_dim, device, obs_shape = 6, 6, th.device("cpu"), (5, 64, 64)
        observation_space, action_space = osp, asp

        def __init__(self):
            self.k = th.zeros(6, dtype=th.int64); self.clock = th.zeros(6, dtype=th.int64); self.waiting = th.ones(6, dtype=th.bool)
            self.out = {"reward": th.zeros(6), "done": th.zeros(6, dtype=th.uint8), "n_substeps": th.zeros(6, dtype=th.int32)}
            self.listed = th.zeros(0, dtype=th.int64)

        def reset(self):
            pass

        def advance(self, slot_actions, slice_len, ready_list, ready_count, lag=1):
            got = ready_list[:int(ready_count.item())]; got = got[got >= 0].long()
            self.waiting[got] = False
            self.clock[~self.waiting] += 1
            fin = (~self.waiting) & (self.clock % (1 + th.arange(6) % 3) == 0)
            self.k[fin] += 1; self.out["reward"][fin] = self.k[fin].float(); self.waiting |= fin
            ids = self.waiting.nonzero().flatten()[:ready_list.numel()]
            ready_list.fill_(-1); ready_list[:ids.numel()] = ids.int(); ready_count.fill_(ids.numel()); self.listed = ids
            return self.out

        def observe_list(self, ready_list, ready_count, obs_rows, records=None, record_row=None):
            obs_rows.zero_()
            for r, e in enumerate(self.listed.tolist()):
                obs_rows[r].fill_(int(40 * e + self.k[e]))
    eng = Eng()
    s = SAC("MultiInputPolicy", eng, buffer_size=64, learning_starts=10_000, batch_size=8, device="cpu", seed=0, async_slice=4, async_capacity=6,
            policy_kwargs=dict(share_features_extractor=True, net_arch=[256, 256]))
    s.learn(40)
    n = s.replay_buffer.sync_size()
    assert n >= 40
    o = s.replay_buffer.obs[:n, 0, 0, 0].long(); no = s.replay_buffer.next_obs[:n, 0, 0, 0].long(); r = s.replay_buffer.rewards[:n]
    assert th.equal(no, o + 1)                              # next observation of a transition = that env's next decision point
    assert th.equal(r.long(), no % 40)                      # and the reward is the one reported when the env was listed again
    assert set((o // 40).tolist()) == set(range(6))         # every env contributes, on its own clock


def test_flat_her_buffer_relabels_with_a_future_goal_of_the_same_episode():
    """HER over the time-sliced collector (n1; train_agent.py:57-79: 'future', n_sampled_goal = 4): transitions of different envs arrive
    interleaved and at arbitrary ring rows; a relabelled sample must take the goal achieved at a LATER step of the SAME episode of the SAME
    env, only episodes that are complete are relabelled, the reward changes by exactly the swap of the e^-|dg - ag| term, and a recycled
    trajectory slot (or a row of an open episode) is left alone."""
    import torch as th
    from mujoco_rl_manipulate_unknown_objects_amd import spaces
    from mujoco_rl_manipulate_unknown_objects_amd.sb3 import FlatHerReplayBuffer
    osp = spaces.Dict({"observation": spaces.Box(0, 255, shape=(1, 2, 2), dtype=np.uint8)}); asp = spaces.Box(-1., 1., shape=(2,), dtype=np.float32)
    N, C = 5, 4
    rb = FlatHerReplayBuffer(4000, osp, asp, th.device("cpu"), n_envs=N, n_sampled_goal=4, max_episode_length=50, n_slots=1024)
    g = th.Generator().manual_seed(0)
    ep = [0] * N; k = [0] * N; length = [3 + 2 * e for e in range(N)]              # env e: episodes of 3 + 2 e steps
    code = lambda e, j, s: float(e * 10000 + j * 100 + s)                          # goal achieved after step s of episode j of env e
    rows_total = 0
    for tick in range(400):
        envs = th.randperm(N, generator=g)[:C]                                     # which envs finished a macro step this tick (any order)
        mask = th.rand(C, generator=g) < 0.8
        ag, dg, nag, ndg, done = th.zeros(C, 2), th.zeros(C, 2), th.zeros(C, 2), th.zeros(C, 2), th.zeros(C)
        for r, e in enumerate(envs.tolist()):
            if not mask[r]:
                continue
            nag[r, 0] = code(e, ep[e], k[e]); ndg[r, 0] = -1.0; ag[r, 0] = code(e, ep[e], k[e]) - 1; dg[r, 0] = -1.0
            k[e] += 1
            if k[e] == length[e]:
                done[r] = 1.0; ep[e] += 1; k[e] = 0
        obs = th.zeros(C, 1, 2, 2, dtype=th.uint8)
        rb.add_rows(mask, obs, obs, th.zeros(C, 2), th.full((C,), 7.0), done, env=th.where(mask, envs, th.full_like(envs, N)), goals=(ag, dg, nag, ndg))
        rows_total += int(mask.sum())
    assert rb.sync_size() == rows_total < 4000
    b = rb.sample(4096, generator=g)
    rel = b["relabelled"]
    assert 0.6 < rel.float().mean() < 0.85                                         # her_ratio 0.8 of the rows whose episode is complete
    own = b["next_obs"]["achieved_goal"][:, 0]; new = b["obs"]["desired_goal"][:, 0]
    e0, j0, s0 = (own // 10000).long(), ((own % 10000) // 100).long(), (own % 100).long()
    e1, j1, s1 = (new // 10000).long(), ((new % 10000) // 100).long(), (new % 100).long()
    assert th.equal(e0[rel], e1[rel]) and th.equal(j0[rel], j1[rel]) and (s1[rel] >= s0[rel]).all()
    assert (s1[rel] < (3 + 2 * e0[rel])).all() and (s1[rel] > s0[rel]).any() and (s1[rel] == s0[rel]).any()
    assert (new[~rel] == -1.0).all()                                               # untouched rows keep the stored desired goal
    # reward: 7 - e^-|old dg - ag'| + e^-|new goal - ag'|
    want = 7.0 - th.exp(-(own[rel] + 1.0).abs()) + th.exp(-(new[rel] - own[rel]).abs())
    assert th.allclose(b["rewards"][rel], want, atol=1e-6) and (b["rewards"][~rel] == 7.0).all()
    # rows of the episodes still open are never relabelled
    open_rows = th.tensor([j0[i] == ep[int(e0[i])] for i in range(len(own))])
    assert not rel[open_rows].any() and open_rows.any()
    # recycling: with only 8 slots the early episodes' slots have been handed out again -- their rows must not be relabelled with foreign goals
    rb2 = FlatHerReplayBuffer(4000, osp, asp, th.device("cpu"), n_envs=N, n_sampled_goal=4, max_episode_length=50, n_slots=8)
    ep = [0] * N; k = [0] * N
    for tick in range(400):
        envs = th.arange(C); mask = th.ones(C, dtype=th.bool)
        ag, dg, nag, ndg, done = th.zeros(C, 2), th.zeros(C, 2), th.zeros(C, 2), th.zeros(C, 2), th.zeros(C)
        for r, e in enumerate(envs.tolist()):
            nag[r, 0] = code(e, ep[e], k[e]); ndg[r, 0] = -1.0; k[e] += 1
            if k[e] == length[e]:
                done[r] = 1.0; ep[e] += 1; k[e] = 0
        obs = th.zeros(C, 1, 2, 2, dtype=th.uint8)
        rb2.add_rows(mask, obs, obs, th.zeros(C, 2), th.zeros(C), done, env=envs, goals=(ag, dg, nag, ndg))
    rb2.sync_size()
    b2 = rb2.sample(4096, generator=g); rel2 = b2["relabelled"]
    own2 = b2["next_obs"]["achieved_goal"][:, 0]; new2 = b2["obs"]["desired_goal"][:, 0]
    assert rel2.any() and th.equal((own2[rel2] // 100).long(), (new2[rel2] // 100).long())       # same env, same episode, always
    assert (rel2.float().mean() < 0.2)                                             # most rows' slots are gone: left alone


def test_flat_her_buffer_keeps_a_long_episode_apart_from_the_slot_s_next_owner():
    """Slots are handed out round-robin. One env's long episode outlasts n_slots - n_envs completions of another env's one-step episodes, so its slot
    is given to a newcomer while it is still open (advisor, round 3): the displaced episode must not publish a trajectory that mixes both envs' goals --
    every relabelled sample still takes a goal of ITS OWN env and episode."""
    import torch as th
    from mujoco_rl_manipulate_unknown_objects_amd import spaces
    from mujoco_rl_manipulate_unknown_objects_amd.sb3 import FlatHerReplayBuffer
    osp = spaces.Dict({"observation": spaces.Box(0, 255, shape=(1, 2, 2), dtype=np.uint8)}); asp = spaces.Box(-1., 1., shape=(2,), dtype=np.float32)
    rb = FlatHerReplayBuffer(2000, osp, asp, th.device("cpu"), n_envs=2, n_sampled_goal=4, max_episode_length=50, n_slots=4)
    code = lambda e, j, s: float(e * 10000 + j * 100 + s)
    ep, k = [0, 0], [0, 0]
    length = [30, 1]                                                               # env 0: 30-step episodes, env 1: one-step episodes
    for tick in range(240):
        envs = th.arange(2); mask = th.ones(2, dtype=th.bool)
        ag, dg, nag, ndg, done = th.zeros(2, 2), th.zeros(2, 2), th.zeros(2, 2), th.zeros(2, 2), th.zeros(2)
        for r, e in enumerate(envs.tolist()):
            nag[r, 0] = code(e, ep[e] % 100, k[e]); ndg[r, 0] = -1.0; k[e] += 1
            if k[e] == length[e]:
                done[r] = 1.0; ep[e] += 1; k[e] = 0
        obs = th.zeros(2, 1, 2, 2, dtype=th.uint8)
        rb.add_rows(mask, obs, obs, th.zeros(2, 2), th.zeros(2), done, env=envs, goals=(ag, dg, nag, ndg))
    n = rb.sync_size()
    # every row that sample() would relabel (its slot's generation is the row's, the slot's episode is complete) must find ONLY goals of its own env and
    # episode in the slot, from its own step to the episode's end -- checked over all rows and all candidate steps, not by sampling
    checked = 0
    for i in range(n):
        slot, step = int(rb.row_slot[i]), int(rb.row_step[i]); L = int(rb.traj_len[slot])
        if int(rb.slot_gen[slot]) != int(rb.row_gen[i]) or L <= step:
            continue
        own = float(rb.next_achieved[i, 0])
        goals = rb.traj_goal[slot, step:L, 0]
        assert ((goals // 100).long() == int(own // 100)).all(), (i, own, goals.tolist())
        assert float(goals[0]) == own
        checked += 1
    assert checked >= 1
    g = th.Generator().manual_seed(1)
    b = rb.sample(8192, generator=g); rel = b["relabelled"]
    own = b["next_obs"]["achieved_goal"][:, 0]; new = b["obs"]["desired_goal"][:, 0]
    assert rel.any()
    assert th.equal((own[rel] // 100).long(), (new[rel] // 100).long())            # same env AND same episode: never the slot's other tenant
