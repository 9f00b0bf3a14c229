"""PPO host logic on CPU, and the N > 1 path on gloo (world_size 2): one flattened all-reduce gives every
rank the gradient of the concatenated batch."""
import os
import socket

import numpy as np
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from mujoco_rl_manipulate_unknown_objects_amd import spaces
from mujoco_rl_manipulate_unknown_objects_amd.sb3 import PPO
from mujoco_rl_manipulate_unknown_objects_amd.sb3.ppo import RolloutBuffer

KW = dict(share_features_extractor=True, net_arch=[256, 256])


class FakeVec:
    observation_space = spaces.Dict({"observation": spaces.Box(0, 255, shape=(5, 64, 64), dtype=np.uint8)})
    action_space = spaces.Box(-1., 1., shape=(6,), dtype=np.float32)

    def __init__(self, n, seed):
        self.num_envs = n
        self.rng = np.random.default_rng(seed)

    def reset(self):
        return {"observation": self.rng.integers(0, 255, (self.num_envs, 5, 64, 64), dtype=np.uint8)}

    def step(self, a):
        return self.reset(), self.rng.random(self.num_envs).astype(np.float32), self.rng.random(self.num_envs) < 0.2, [{}] * self.num_envs


def test_gae_matches_reference_recursion():
    buf = RolloutBuffer(5, 3, (5, 64, 64), 6, torch.device("cpu"))
    g = torch.Generator().manual_seed(0)
    buf.rewards = torch.rand(5, 3, generator=g); buf.values = torch.rand(5, 3, generator=g)
    buf.dones = (torch.rand(5, 3, generator=g) < 0.3).float()
    last_v = torch.rand(3, generator=g); last_d = torch.tensor([0., 1., 0.])
    buf.compute_returns(last_v, last_d, 0.99, 0.95)
    adv = np.zeros((5, 3)); last = np.zeros(3)
    for t in reversed(range(5)):
        nt = 1 - (last_d.numpy() if t == 4 else buf.dones[t + 1].numpy()); nv = last_v.numpy() if t == 4 else buf.values[t + 1].numpy()
        delta = buf.rewards[t].numpy() + 0.99 * nv * nt - buf.values[t].numpy()
        last = delta + 0.99 * 0.95 * nt * last; adv[t] = last
    assert np.allclose(buf.advantages.numpy(), adv, atol=1e-6) and np.allclose(buf.returns.numpy(), adv + buf.values.numpy(), atol=1e-6)


def test_ppo_learn_save_load(tmp_path):
    env = FakeVec(8, 0)
    m = PPO("MultiInputPolicy", env, n_steps=4, batch_size=16, n_epochs=2, device="cpu", policy_kwargs=KW, seed=0)
    assert sum(p.numel() for p in m.policy.parameters()) == 999853       # 602784 + 199180 + 197889
    m.learn(64)
    assert m.num_timesteps == 64 and np.isfinite(float(m.logger["loss"]))
    m.save(str(tmp_path / "best_model"))
    m2 = PPO.load(str(tmp_path / "best_model"), env=env, device="cpu", custom_objects={"policy_kwargs": KW})
    o = env.reset()
    a1, _ = m.predict(o, deterministic=True); a2, _ = m2.predict(o, deterministic=True)
    assert np.array_equal(a1, a2) and a1.shape == (8, 6)


def _worker(rank, world, port, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    torch.manual_seed(100 + rank)          # different init per rank: the constructor must broadcast rank 0's
    m = PPO("MultiInputPolicy", FakeVec(4, 50 + rank), n_steps=2, batch_size=8, n_epochs=1, device="cpu", policy_kwargs=KW)
    p0 = torch.cat([p.detach().reshape(-1) for p in m.policy.parameters()])
    # per-rank loss on rank-specific data, then the single flattened all-reduce
    x = {"observation": torch.from_numpy(FakeVec(4, 7 + rank).reset()["observation"])}
    acts = torch.zeros(4, 6)
    v, lp, _ = m.policy.evaluate_actions(x, acts)
    (v.sum() + lp.sum()).backward()
    local = torch.cat([p.grad.reshape(-1) for p in m.policy.parameters()]).clone()
    m._allreduce_grads()
    avg = torch.cat([p.grad.reshape(-1) for p in m.policy.parameters()])
    q.put((rank, p0.numpy(), local.numpy(), avg.numpy()))
    m.learn(16)                               # a full distributed iteration runs
    p1 = torch.cat([p.detach().reshape(-1) for p in m.policy.parameters()])
    q.put((rank + 10, p1.numpy(), None, None))
    dist.destroy_process_group()


def test_gloo_world2_gradient_allreduce():
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    got = {}
    for _ in range(4):
        r, a, b, c = q.get(timeout=300)
        got[r] = (a, b, c)
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert np.array_equal(got[0][0], got[1][0])                                   # parameters broadcast from rank 0
    mean = (got[0][1] + got[1][1]) / 2
    assert np.allclose(got[0][2], mean, atol=1e-6) and np.allclose(got[1][2], mean, atol=1e-6)
    assert np.allclose(got[10][0], got[11][0], atol=1e-6)                         # ranks stay in lock-step after an update


def test_monitor_results_and_save_on_best_callback(tmp_path):
    """Monitor CSV -> load_results / ts2xy -> SaveOnBestTrainingRewardCallback (models/callbacks.py:41-82 of the reference)."""
    from mujoco_rl_manipulate_unknown_objects_amd.sb3.vec_env import Monitor
    from mujoco_rl_manipulate_unknown_objects_amd.sb3.results_plotter import load_results, ts2xy
    from mujoco_rl_manipulate_unknown_objects_amd.models.callbacks import SaveOnBestTrainingRewardCallback

    class OneEnv:
        observation_space = FakeVec.observation_space; action_space = FakeVec.action_space

        def __init__(self):
            self.t = 0

        def reset(self):
            self.t = 0
            return {"observation": np.zeros((5, 64, 64), np.uint8)}

        def step(self, a):
            self.t += 1
            return {"observation": np.zeros((5, 64, 64), np.uint8)}, 0.5, self.t == 3, {}

        def close(self):
            pass
    log_dir = str(tmp_path)
    env = Monitor(OneEnv(), os.path.join(log_dir, "log_file"))
    for _ in range(4):
        env.reset()
        d = False
        while not d:
            _, _, d, info = env.step(np.zeros(6))
        assert info["episode"]["l"] == 3 and abs(info["episode"]["r"] - 1.5) < 1e-9
    df = load_results(log_dir)
    x, y = ts2xy(df, "timesteps")
    assert list(x) == [3, 6, 9, 12] and np.allclose(y, 1.5)
    model = PPO("MultiInputPolicy", FakeVec(2, 0), n_steps=2, batch_size=4, n_epochs=1, device="cpu", policy_kwargs=KW)
    cb = SaveOnBestTrainingRewardCallback(check_freq=1, log_dir=log_dir, verbose=0)
    cb.init_callback(model)
    assert cb.on_step() and cb.best_mean_reward == 1.5
    assert os.path.exists(os.path.join(log_dir, "best_model_training.zip"))


class InPlaceVec:
    """A vector env that hands out ONE observation tensor and rewrites it in place on every step, as the reference's RobotEnv does
    with self.obs (robot_env.py:35,223). Observation t is filled with the value t."""
    observation_space = spaces.Dict({"observation": spaces.Box(0, 255, shape=(5, 64, 64), dtype=np.uint8),
                                     "achieved_goal": spaces.Box(-10, 10, shape=(2,), dtype=np.float32),
                                     "desired_goal": spaces.Box(-10, 10, shape=(2,), dtype=np.float32)})
    action_space = spaces.Box(-1., 1., shape=(6,), dtype=np.float32)
    device = torch.device("cpu")                     # tensor env: PPO / SAC keep what it returns without converting

    def __init__(self, n):
        self.num_envs = n; self.t = 0
        self.buf = torch.zeros(n, 5, 64, 64, dtype=torch.uint8); self.ag = torch.zeros(n, 2); self.dg = torch.zeros(n, 2)

    def _obs(self):
        self.buf.fill_(self.t); self.ag.fill_(float(self.t)); self.dg.fill_(float(self.t) + 0.5)
        return {"observation": self.buf, "achieved_goal": self.ag, "desired_goal": self.dg}

    def reset(self):
        self.t = 0
        return self._obs()

    def step(self, a):
        self.t += 1
        return self._obs(), torch.full((self.num_envs,), float(self.t)), torch.zeros(self.num_envs, dtype=torch.bool), {}


def test_rollout_and_replay_buffers_survive_an_env_that_reuses_its_observation_buffer():
    """The observation stored with action_t must be obs_t although the env rewrites its tensor during step() (round-1 advisor
    finding: the buffers held obs_{t+1}; SAC saw obs == next_obs)."""
    from mujoco_rl_manipulate_unknown_objects_amd.sb3 import SAC
    env = InPlaceVec(3)
    m = PPO("MultiInputPolicy", env, n_steps=4, batch_size=12, n_epochs=1, device="cpu", policy_kwargs=KW, seed=0)
    m.collect_rollouts()
    assert m.rollout_buffer.obs[:, 0, 0, 0, 0].tolist() == [0, 1, 2, 3]
    assert m.rollout_buffer.rewards[:, 0].tolist() == [1.0, 2.0, 3.0, 4.0]
    env2 = InPlaceVec(2)
    s = SAC("MultiInputPolicy", env2, buffer_size=16, learning_starts=1000, batch_size=4, device="cpu", policy_kwargs=KW, seed=0)
    s.learn(8)
    rb = s.replay_buffer
    assert rb.obs[:4, 0, 0, 0, 0].tolist() == [0, 1, 2, 3] and rb.next_obs[:4, 0, 0, 0, 0].tolist() == [1, 2, 3, 4]
    assert rb.achieved[:4, 0, 0].tolist() == [0.0, 1.0, 2.0, 3.0] and rb.next_achieved[:4, 0, 0].tolist() == [1.0, 2.0, 3.0, 4.0]
    assert rb.next_desired[:4, 0, 0].tolist() == [1.5, 2.5, 3.5, 4.5]


class ScriptedAsyncEngine:
    """Duck-typed time-sliced engine on CPU: env e finishes a macro step every (1 + (e + rank) % 3) ticks."""

    def __init__(self, n, rank):
        self.num_envs, self.action_dim, self.device, self.obs_shape, self.rank = n, 6, torch.device("cpu"), (5, 64, 64), rank
        self.clock = torch.zeros(n, dtype=torch.int64); self.waiting = torch.ones(n, dtype=torch.bool); self.t = 0
        self.out = {"reward": torch.zeros(n), "done": torch.zeros(n, dtype=torch.uint8), "n_substeps": torch.zeros(n, dtype=torch.int32)}

    def reset(self):
        pass

    def advance(self, slot_actions, slice_len, ready_list, ready_count, lag=1):
        self.t += 1
        period = 1 + (torch.arange(self.num_envs) + self.rank) % 3
        listed = ready_list[:int(ready_count.item())]
        listed = listed[listed >= 0].long()
        self.waiting[listed] = False                       # they got an action: run again
        self.clock[~self.waiting] += 1
        fin = (~self.waiting) & (self.clock % period == 0)
        self.out["reward"][fin] = 1.0; self.out["n_substeps"][fin] = 10
        self.waiting |= fin
        ids = self.waiting.nonzero().flatten()[:ready_list.numel()]
        ready_list.fill_(-1); ready_list[:ids.numel()] = ids.int(); ready_count.fill_(ids.numel())
        return self.out

    def observe_list(self, ready_list, ready_count, obs_rows, records=None, record_row=None):
        obs_rows.fill_(self.t % 250)


def _async_learn_worker(rank, world, port, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    eng = ScriptedAsyncEngine(8, rank)
    eng.observation_space = FakeVec.observation_space; eng.action_space = FakeVec.action_space
    m = PPO("MultiInputPolicy", eng, n_steps=2, batch_size=16, n_epochs=1, device="cpu", policy_kwargs=KW, async_slice=4, async_capacity=8)
    m.learn(40)                                            # polled completions overshoot the target by a rank-dependent amount
    p1 = torch.cat([p.detach().reshape(-1) for p in m.policy.parameters()])
    q.put((rank, p1.numpy(), m.num_timesteps))
    dist.destroy_process_group()


def test_gloo_world2_async_learn_leaves_the_loop_on_the_same_iteration():
    """PPO.learn over the time-sliced rollout with two ranks whose envs finish at different rates: num_timesteps advances by polled
    completions, so the ranks would pass `total_timesteps` on different iterations and the slower one would wait for ever in the
    gradient all-reduce (round-1 advisor finding). The loop agrees on stopping; replicas stay identical."""
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_async_learn_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    got = {}
    for _ in range(2):
        r, a, n = q.get(timeout=300)
        got[r] = (a, n)
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert np.allclose(got[0][0], got[1][0], atol=1e-6)
    assert got[0][1] >= 40 and got[1][1] >= 40
