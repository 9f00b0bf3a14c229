"""-m gpu: the Gym / SB3-shaped surfaces (RobotEnv, GpuVecEnv, PPO) on the real engine."""
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def torch():
    import torch as t
    if not t.cuda.is_available():
        pytest.skip("no GPU")
    return t


def test_robot_env_surface(torch):
    from mujoco_rl_manipulate_unknown_objects_amd.simulation.environment.robot_env import RobotEnv, default_config, Status
    env = RobotEnv(default_config(sim_env="/xmls/sand_ball_env.xml"))
    assert env.action_space.shape == (6,) and env.observation_space["observation"].shape == (5, 64, 64)
    obs = env.reset()
    assert obs["observation"].shape == (5, 64, 64) and obs["observation"].dtype == np.uint8
    assert obs["achieved_goal"].dtype == np.float32 and np.allclose(obs["desired_goal"], [1, 0])
    obs, reward, done, info = env.step(env.action_space.sample())
    keys = {"old_obs", "new_obs", "init_obj_pos", "final_obj_pos", "target_dir", "gripper_open", "controls", "object_grasped",
            "episode_step", "episode_rewards", "status", "gripper_position", "object_position", "position_reached",
            "total_distance", "line_distance"}
    assert keys <= set(info)                                     # robot_env.py:226-241
    assert isinstance(reward, float) and isinstance(done, bool) and info["status"] == Status.RUNNING and info["episode_step"] == 1
    r2 = env.compute_reward(obs["achieved_goal"], obs["desired_goal"], info)
    assert float(r2) == pytest.approx(reward, abs=1e-4)
    assert env.render(mode="rgb_array").shape == (480, 1920, 3)    # robot_env.py:302-340: cameras 0..2 side by side at 64 * (10, 7.5)
    env.close()


def test_render_cameras_video_and_gif(torch, orc, tmp_path):
    """n4: RobotEnv.render (robot_env.py:302-340) -- the three cameras at the zoomed size, rgb / depth / human modes; the gripper view
    at zoom 1 is the observation's own RGB; the general-camera kernel agrees with the oracle's ray caster at another size; the video
    recorder of train_agent.py:25-29 and the GIF / 3-D plot of eval_agent.py:11-25,72-76 write their files."""
    from mujoco_rl_manipulate_unknown_objects_amd.simulation.environment.robot_env import RobotEnv, default_config
    from mujoco_rl_manipulate_unknown_objects_amd.sb3 import DummyVecEnv, VecVideoRecorder
    from mujoco_rl_manipulate_unknown_objects_amd.visuals import make_gif, plot_3D
    env = RobotEnv(default_config(sim_env="/xmls/sugar_cube_env.xml"))
    obs = env.reset()
    rng = np.random.default_rng(3)
    for _ in range(3):
        a = rng.uniform(-1, 1, 6).astype(np.float32); a[0] = abs(a[0])
        obs, _, _, info = env.step(a)
    allcams = env.render(mode="rgb_array")
    assert allcams.shape == (480, 1920, 3) and allcams.dtype == np.uint8
    for k in range(3):                                              # every panel shows a scene, not a constant image
        assert allcams[:, 640 * k:640 * (k + 1)].reshape(-1, 3).std(0).max() > 5
    d = env.render(mode="depth_array")
    assert d.shape == (480, 1920) and np.isfinite(d).all() and 0 <= d.min() and d.max() <= 255
    assert env.render(mode="human") is None and env.last_frame.shape == (480, 1920, 3)
    # gripper camera, zoom 1: the same pixels as the observation (two kernels, one scene)
    env.config.camera_id, env.config.rendering_zoom_width, env.config.rendering_zoom_height = 2, 1, 1
    g = env.render(mode="rgb_array")
    assert g.shape == (64, 64, 3)
    o = obs["observation"][:3].transpose(1, 2, 0).astype(int)
    assert (np.abs(g.astype(int) - o) <= 1).mean() > 0.99
    # the general-camera kernel against the oracle's ray caster at 128 x 96
    import ctypes as C
    m = orc.Model("sugar_cube"); e = orc.EnvOracle(m); e.reset()
    q = env.batch.get_state()[0][0]
    e.d.qpos[:] = [float(x) for x in q]; orc.lib().orc_fwd_position(m.ptr, C.byref(e.e.d))
    rgb_o = np.zeros((96, 128, 3), np.uint8); dep_o = np.zeros((96, 128), np.float32)
    orc.lib().orc_render(m.ptr, C.byref(e.e.d), 128, 96, rgb_o.ctypes.data_as(C.POINTER(C.c_ubyte)), dep_o.ctypes.data_as(C.POINTER(C.c_float)))
    rgb_g = env.batch.render_camera(0, width=128, height=96).cpu().numpy(); dep_g = env.batch.render_camera(0, width=128, height=96, depth=True).cpu().numpy()
    assert (np.abs(rgb_g.astype(int) - rgb_o.astype(int)) <= 1).mean() > 0.98
    near = dep_o < 5
    assert (np.abs(dep_g - dep_o)[near] < 1e-3).mean() > 0.98
    # upper camera looks straight down on the object: the object's colour is at the image centre
    env.config.camera_id, env.config.rendering_zoom_width, env.config.rendering_zoom_height = 1, 2, 2
    up = env.render(mode="rgb_array")
    assert up.shape == (128, 128, 3)
    # video recorder + GIF + 3-D plot
    env.config.camera_id = 2
    venv = VecVideoRecorder(DummyVecEnv([lambda: env]), video_folder=str(tmp_path / "videos"), record_video_trigger=lambda x: x % 4 == 0, video_length=3, name_prefix="probe")
    venv.reset()
    frames, gp, op = [], [], []
    for _ in range(6):
        ob, r, dn, infos = venv.step([rng.uniform(-1, 1, 6).astype(np.float32)])
        frames.append(env.render(mode="rgb_array")); gp.append(infos[0]["gripper_position"]); op.append(infos[0]["object_position"])
    venv.close_video_recorder()
    assert len(venv.saved) >= 1 and all(os.path.getsize(p) > 500 for p in venv.saved)
    gif = make_gif(frames, str(tmp_path / "gifs" / "traj.gif")); png = plot_3D(gp, op, str(tmp_path / "plot.png"))
    assert os.path.getsize(gif) > 500 and os.path.getsize(png) > 2000
    env.close()


def test_ppo_learns_on_gpu_vec_env(torch):
    from mujoco_rl_manipulate_unknown_objects_amd.simulation.environment.robot_env import BatchedRobotEnv, default_config
    from mujoco_rl_manipulate_unknown_objects_amd.sb3 import PPO, GpuVecEnv
    from mujoco_rl_manipulate_unknown_objects_amd.models.feature_extractor import AugmentedNatureCNN
    env = GpuVecEnv(BatchedRobotEnv(default_config(sim_env="/xmls/sand_ball_env.xml"), n_envs=128, auto_reset=True))
    model = PPO("MultiInputPolicy", env, n_steps=4, batch_size=128, n_epochs=2,
                policy_kwargs=dict(features_extractor_class=AugmentedNatureCNN, share_features_extractor=True, net_arch=[256, 256]))
    before = [p.detach().clone() for p in model.policy.parameters()]
    model.learn(total_timesteps=128 * 4 * 2)
    assert model.num_timesteps == 1024 and np.isfinite(float(model.logger["loss"]))
    assert any(not torch.equal(a, b) for a, b in zip(before, model.policy.parameters()))
    a, _ = model.predict({"observation": env.env._obs[0].cpu().numpy()}, deterministic=True)
    assert a.shape == (6,) and np.abs(a).max() <= 1
    env.close()


def test_sac_with_her_on_gpu_vec_env(torch):
    """The reference's training call (train_agent.py:58-81: SAC + HerReplayBuffer, her_buffer reward) over the batched GPU env."""
    from mujoco_rl_manipulate_unknown_objects_amd.simulation.environment.robot_env import BatchedRobotEnv, default_config
    from mujoco_rl_manipulate_unknown_objects_amd.sb3 import SAC, HerReplayBuffer, GpuVecEnv
    from mujoco_rl_manipulate_unknown_objects_amd.models.feature_extractor import AugmentedNatureCNN
    cfg = default_config(sim_env="/xmls/sand_ball_env.xml", her_buffer=True, time_horizon=6)
    env = GpuVecEnv(BatchedRobotEnv(cfg, n_envs=32, device_index=0, auto_reset=True))
    model = SAC("MultiInputPolicy", env, replay_buffer_class=HerReplayBuffer,
                replay_buffer_kwargs=dict(n_sampled_goal=4, goal_selection_strategy="future", online_sampling=True, max_episode_length=cfg.time_horizon),
                learning_starts=64, buffer_size=32 * 64, batch_size=64, seed=0,
                policy_kwargs=dict(features_extractor_class=AugmentedNatureCNN, share_features_extractor=True, net_arch=[256, 256]))
    model.learn(32 * 10)
    torch.cuda.synchronize()
    assert model._n_updates >= 7 and all(np.isfinite(float(model.logger[k])) for k in ("actor_loss", "critic_loss"))
    b = model.replay_buffer.sample(256)
    assert bool(b["relabelled"].any()) and bool(torch.isfinite(b["rewards"]).all())
    a, _ = model.predict({"observation": env.reset()["observation"]}, deterministic=True)
    assert a.shape == (32, 6) and np.abs(a).max() <= 1.0
    env.close()


def test_fused_observation_preprocessing_matches_tensor_ops(torch):
    """grip_obs_preprocess (cast + / 255 + NHWC in one kernel) against the three tensor passes it replaces, and the feature
    extractor's two input paths against each other."""
    from mujoco_rl_manipulate_unknown_objects_amd import engine, spaces
    from mujoco_rl_manipulate_unknown_objects_amd.models.feature_extractor import AugmentedNatureCNN
    g = torch.Generator(device="cuda"); g.manual_seed(0)
    for ch in (5, 4):
        obs = torch.randint(0, 256, (37, ch, 64, 64), dtype=torch.uint8, device="cuda", generator=g)
        img, other = engine.obs_preprocess(obs)
        ref = obs.float() / 255.0
        assert img.is_contiguous(memory_format=torch.channels_last) and torch.equal(img, ref[:, :-1]) and torch.equal(other, ref[:, -1, 0, :2])
        space = spaces.Dict({"observation": spaces.Box(0, 255, (ch, 64, 64), np.uint8)})
        fe = AugmentedNatureCNN(space).cuda().to(memory_format=torch.channels_last)
        with torch.no_grad():
            a = fe({"observation": obs}); b = fe({"observation": ref})
        assert torch.allclose(a, b, rtol=1e-5, atol=1e-6)


def test_fused_first_layer_matches_the_tensor_library(torch):
    """grip_conv1_u8 (uint8 rows -> /255 -> Conv2d(4, 32, 8, 4) -> bias -> ReLU on f32 MFMA, plus the two pad scalars) against
    torch.nn.functional on the same weights, for contiguous and channels-last weight layouts and ragged batch sizes; and the
    extractor's no-grad path (which uses it) against its autograd path (which does not). fp32 both ways: 2e-5 absolute on
    activations of order 1 (summation order differs)."""
    import torch.nn.functional as F
    from mujoco_rl_manipulate_unknown_objects_amd.engine import conv1_u8
    from mujoco_rl_manipulate_unknown_objects_amd.models.feature_extractor import AugmentedNatureCNN
    from mujoco_rl_manipulate_unknown_objects_amd.simulation.controller.sensor import RGBDSensor
    from mujoco_rl_manipulate_unknown_objects_amd.simulation.environment.robot_env import default_config
    g = torch.Generator(device="cuda"); g.manual_seed(0)
    for n in (1, 7, 130):
        obs = torch.randint(0, 256, (n, 5, 64, 64), dtype=torch.uint8, device="cuda", generator=g)
        w = torch.randn(32, 4, 8, 8, device="cuda", generator=g) * 0.1; b = torch.randn(32, device="cuda", generator=g)
        ref = F.relu(F.conv2d(obs[:, :4].float() / 255.0, w, b, stride=4))
        for wl in (w, w.contiguous(memory_format=torch.channels_last)):
            y, other = conv1_u8(obs, wl, b)
            assert y.shape == ref.shape and y.is_contiguous(memory_format=torch.channels_last)
            assert (y - ref).abs().max().item() < 2e-5
            assert torch.equal(other, obs[:, 4, 0, :2].float() / 255.0)
    # exact-integer check of the operand / accumulator lane maps: weights = one-hot taps, so every output is one pixel / 255
    obs = torch.randint(0, 256, (3, 5, 64, 64), dtype=torch.uint8, device="cuda", generator=g)
    w = torch.zeros(32, 4, 8, 8, device="cuda")
    taps = [(n_, n_ % 4, (3 * n_) % 8, (5 * n_ + 1) % 8) for n_ in range(32)]
    for n_, ci, ky, kx in taps:
        w[n_, ci, ky, kx] = 255.0
    y, _ = conv1_u8(obs, w, torch.zeros(32, device="cuda"))
    for n_, ci, ky, kx in taps:
        want = obs[:, ci, ky:ky + 57:4, kx:kx + 57:4].float()
        assert (y[:, n_] - want).abs().max().item() < 1e-3, n_
    fe = AugmentedNatureCNN(RGBDSensor(config=default_config()).setup_observation_space()).cuda().to(memory_format=torch.channels_last)
    obs = torch.randint(0, 256, (33, 5, 64, 64), dtype=torch.uint8, device="cuda", generator=g)
    with torch.no_grad():
        fast = fe({"observation": obs})
    slow = fe({"observation": obs})                       # autograd on: tensor-library path
    assert slow.requires_grad and not fast.requires_grad
    assert (fast - slow.detach()).abs().max().item() < 2e-5


def test_overlapped_update_keeps_one_update_of_lag(torch):
    """overlap_update: rollout i + 1 is collected with the parameters after update i - 1 while update i runs on a second
    stream; the rollout copy only ever holds complete parameter sets."""
    from mujoco_rl_manipulate_unknown_objects_amd.simulation.environment.robot_env import BatchedRobotEnv, default_config
    from mujoco_rl_manipulate_unknown_objects_amd.sb3 import PPO, GpuVecEnv
    from mujoco_rl_manipulate_unknown_objects_amd.models.feature_extractor import AugmentedNatureCNN
    env = GpuVecEnv(BatchedRobotEnv(default_config(sim_env="/xmls/sand_ball_env.xml"), n_envs=64, device_index=0, auto_reset=True))
    m = PPO("MultiInputPolicy", env, n_steps=2, batch_size=64, n_epochs=1, seed=0, async_slice=32, async_capacity=32, overlap_update=True,
            policy_kwargs=dict(features_extractor_class=AugmentedNatureCNN, share_features_extractor=True, net_arch=[64, 64]))
    flat = lambda pol: torch.cat([p.detach().reshape(-1) for p in pol.parameters()]).clone()
    history = [flat(m.policy)]                                   # theta_0
    for i in range(4):
        assert m.collect_rollouts()
        m.train(); m.finish_updates(); torch.cuda.synchronize()
        history.append(flat(m.policy))                           # theta_{i+1}
        assert torch.equal(flat(m.policy_rollout), history[i])   # the copy made before update i ran: parameters of update i - 1
        assert not torch.equal(history[i + 1], history[i]) and bool(torch.isfinite(history[i + 1]).all())
    assert m.num_timesteps >= 4 * 128
    env.close()


def test_eval_agent_loop_shape(torch, tmp_path):
    """eval_agent.py:28-70 of the reference against this package: SAC.load(best_model, env=env, custom_objects=...), then the
    deterministic predict / step loop reading the info keys it accumulates, and render(mode='rgb_array')."""
    from mujoco_rl_manipulate_unknown_objects_amd.simulation.environment.robot_env import RobotEnv, default_config
    from mujoco_rl_manipulate_unknown_objects_amd.sb3 import SAC, DummyVecEnv
    from mujoco_rl_manipulate_unknown_objects_amd.models.feature_extractor import AugmentedNatureCNN
    kw = dict(features_extractor_class=AugmentedNatureCNN, share_features_extractor=True, net_arch=[256, 256])
    env = RobotEnv(default_config(sim_env="/xmls/sugar_cube_env.xml", time_horizon=4))
    SAC("MultiInputPolicy", DummyVecEnv([lambda: env]), policy_kwargs=kw, buffer_size=8, device="cuda").save(str(tmp_path / "best_model"))
    model = SAC.load(str(tmp_path / "best_model"), env=env, custom_objects={"policy_kwargs": kw})
    total_step, line_step, robot_step, obj_step, frames = [], [], [], [], []
    obs = env.reset()
    for i in range(10):
        action, _states = model.predict(obs, deterministic=True)
        obs, rewards, dones, info = env.step(action)
        total_step.append(info["total_distance"]); line_step.append(info["line_distance"])
        robot_step.append(info["gripper_position"].copy()); obj_step.append(info["object_position"].copy())
        frames.append(env.render(mode='rgb_array'))
        if dones:
            break
    assert i == 3 and dones and info["status"].name == "TIME_LIMIT"           # time_horizon = 4
    assert frames[0].shape == (480, 1920, 3) and frames[0].dtype == np.uint8      # robot_env.py:302-340: three cameras at the zoomed size
    assert np.isfinite(sum(total_step)) and np.isfinite(sum(line_step)) and robot_step[0].shape == (3,) and obj_step[0].shape == (3,)
    env.close()


@pytest.mark.parametrize("extra", [[], ["--lockstep", "--no-cpu-baseline"], ["--overlap-update", "--no-cpu-baseline"]])
def test_bench_json_contract(torch, extra):
    """bench.py prints exactly one JSON line with the driver's keys, the roofline object and (N = 1) the cpu baseline."""
    import json, os, subprocess, sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    cmd = [sys.executable, os.path.join(root, "bench.py"), "--envs", "256", "--steps", "4", "--warmup", "2", "--rollout", "2", "--minibatch", "256",
           "--capacity", "64", "--object", "sand_ball"] + extra
    r = subprocess.run(cmd, capture_output=True, text=True, timeout=280)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1
    d = json.loads(lines[0])
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline", "dtype", "data",
              "config", "roofline"):
        assert k in d, k
    assert d["n_gpus"] == 1 and d["steps"] == 4 and d["warmup"] == 2 and d["scaling"] == "weak" and d["vs_baseline"] is None and d["value"] > 0
    assert "workload" in d["config"] and "model" not in d["config"]
    for k in ("bound", "achieved", "peak", "unit", "frac", "traffic"):
        assert k in d["roofline"], k
    assert abs(d["roofline"]["frac"] - d["roofline"]["achieved"] / d["roofline"]["peak"]) < 1e-12
    if "--no-cpu-baseline" not in extra:
        for k in ("value", "unit", "cores", "kind", "sample"):
            assert k in d["cpu_baseline"], k


def test_integration_md_ctypes_stub_runs(torch):
    """The ctypes binding INTEGRATION.md section 2 shows a maintainer of the reference is executed as written (library and
    model paths resolved to this checkout) and its results compared with the package's own binding."""
    import os
    import re
    from mujoco_rl_manipulate_unknown_objects_amd import engine
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    md = open(os.path.join(root, "INTEGRATION.md")).read()
    blocks = re.findall(r"```python\n(.*?)```", md, re.S)
    stub = next(b for b in blocks if 'C.CDLL("libgrip_sim.so")' in b)
    stub = stub.replace('"libgrip_sim.so"', repr(engine.LIB_PATH)).replace('b"assets/acorn_env.grpm"', repr(engine.asset_path("acorn").encode()))
    torch.manual_seed(5)
    ns = {}
    exec(compile(stub, "INTEGRATION.md#2", "exec"), ns)
    torch.cuda.synchronize()
    N = ns["N"]
    assert N == 4096 and ns["obs"].shape == (N, 5, 64, 64) and ns["obs"].float().mean().item() > 1.0
    assert torch.isfinite(ns["reward"]).all() and ns["done"].max().item() <= 1
    b = engine.Batch("acorn", N)
    out = b.step(ns["actions"]); ref_obs = b.observe(); torch.cuda.synchronize()
    assert torch.equal(out["reward"], ns["reward"]) and torch.equal(out["done"], ns["done"]) and torch.equal(ref_obs, ns["obs"])
    ns["L"].grip_batch_destroy(ns["batch"]); ns["L"].grip_model_free(ns["model"])
    b.close()


def test_sac_over_the_time_sliced_engine(torch):
    """n1 on the asynchronous schedule (train_agent.py:82-92 is SAC): collection through grip_batch_advance, transitions of the envs
    that finished a macro step go into the flat replay ring, updates run between ticks. The stored chains are consistent (a
    transition's next observation is a real rendered frame that differs from its observation), losses are finite, parameters move."""
    from mujoco_rl_manipulate_unknown_objects_amd.simulation.environment.robot_env import BatchedRobotEnv, default_config
    from mujoco_rl_manipulate_unknown_objects_amd.sb3 import SAC, GpuVecEnv
    from mujoco_rl_manipulate_unknown_objects_amd.models.feature_extractor import AugmentedNatureCNN
    env = GpuVecEnv(BatchedRobotEnv(default_config(sim_env="/xmls/sand_ball_env.xml", time_horizon=4), n_envs=256, auto_reset=True))
    model = SAC("MultiInputPolicy", env, buffer_size=4096, learning_starts=600, batch_size=128, seed=0, async_slice=48, async_capacity=128, async_budget_us=0,
                policy_kwargs=dict(features_extractor_class=AugmentedNatureCNN, share_features_extractor=True, net_arch=[256, 256]))
    before = [p.detach().clone() for p in model.policy.parameters()]
    model.learn(total_timesteps=1500)
    rb = model.replay_buffer
    n = rb.sync_size()
    assert n >= 1500 and model._n_updates > 0
    assert np.isfinite(float(model.logger["critic_loss"])) and np.isfinite(float(model.logger["actor_loss"]))
    assert any(not torch.equal(a, b) for a, b in zip(before, model.policy.parameters()))
    o, no = rb.obs[:n], rb.next_obs[:n]
    assert int((o != no).flatten(1).any(1).sum()) > 0.9 * n                # the env moved between the two frames
    assert float(rb.dones[:n].mean()) > 0 and float(rb.rewards[:n].abs().max()) < 10
    assert (rb.actions[:n].abs() <= 1).all()
    env.close()


def test_sac_with_hindsight_replay_over_the_time_sliced_engine(torch):
    """n1, the reference's HER branch (train_agent.py:57-79: HerReplayBuffer, n_sampled_goal 4, 'future', online sampling) on the schedule
    the engine is fast on: the time-sliced collector fills a FlatHerReplayBuffer -- per-episode goal trajectories beside the flat ring --
    and SAC trains on relabelled samples. Stored goals are the ones the step's reward was computed with (final object position and its
    projection on the target direction); a relabelled sample carries a goal achieved later in the same episode and its reward moved by
    exactly the e^-|dg - ag| swap (robot_env.py:268-271)."""
    from mujoco_rl_manipulate_unknown_objects_amd.simulation.environment.robot_env import BatchedRobotEnv, default_config
    from mujoco_rl_manipulate_unknown_objects_amd.sb3 import SAC, GpuVecEnv, HerReplayBuffer, FlatHerReplayBuffer
    from mujoco_rl_manipulate_unknown_objects_amd.models.feature_extractor import AugmentedNatureCNN
    env = GpuVecEnv(BatchedRobotEnv(default_config(sim_env="/xmls/sand_ball_env.xml", time_horizon=4, her_buffer=True), n_envs=256, auto_reset=True))
    model = SAC("MultiInputPolicy", env, buffer_size=8192, learning_starts=600, batch_size=128, seed=0, async_slice=48, async_capacity=128, async_budget_us=0,
                replay_buffer_class=HerReplayBuffer, replay_buffer_kwargs=dict(n_sampled_goal=4, goal_selection_strategy="future", online_sampling=True, max_episode_length=4),
                policy_kwargs=dict(features_extractor_class=AugmentedNatureCNN, share_features_extractor=True, net_arch=[256, 256]))
    rb = model.replay_buffer
    assert isinstance(rb, FlatHerReplayBuffer)
    model.learn(total_timesteps=2500)
    n = rb.sync_size()
    assert n >= 2500 and model._n_updates > 0 and np.isfinite(float(model.logger["critic_loss"]))
    # stored goals: the desired goal of a step is the projection of its achieved goal on the target direction (1, 0): (x, 0)
    nag, ndg = rb.next_achieved[:n], rb.next_desired[:n]
    assert torch.allclose(ndg[:, 0], nag[:, 0], atol=1e-6) and float(ndg[:, 1].abs().max()) == 0.0
    assert float(rb.dones[:n].mean()) > 0.15                                # 4-step episodes
    g = torch.Generator(device="cuda"); g.manual_seed(1)
    b = rb.sample(4096, generator=g); rel = b["relabelled"]
    assert 0.3 < float(rel.float().mean()) < 0.85                           # 0.8 of the rows whose episode is complete and still in the trajectory store
    newg = b["obs"]["desired_goal"]; own = b["next_obs"]["achieved_goal"]
    assert torch.equal(newg, b["next_obs"]["desired_goal"])
    # the relabelled goal is an achieved goal of the same episode: of this very step, or of one of its (at most 3) successors
    diff = (newg[rel] - own[rel]).abs().amax(1)
    assert float((diff == 0).float().mean()) > 0.2 and float((diff > 0).float().mean()) > 0.2 and float(diff.max()) < 0.2
    # reward moved by exactly the swap of the goal-dependent term
    i_rewards = b["rewards"]
    assert torch.isfinite(i_rewards).all() and float(i_rewards.abs().max()) < 12
    env.close()


def test_fused_first_layer_gradients_match_the_library_path(torch):
    """a17 in training: AugmentedNatureCNN's first layer through grip_conv1_u8 (f32 MFMA, custom autograd: _Conv1U8) against the tensor
    library's convolution + ReLU on the same uint8 observations. The layer in isolation under a linear loss (a ReLU sitting within 5e-8 of
    zero may flip between the two forwards; under a linear loss that moves a gradient by one term of ~7e5): output to 2e-5, weight and
    bias gradients to 1e-4 relative. The whole extractor: features to 2e-5."""
    from mujoco_rl_manipulate_unknown_objects_amd import spaces
    from mujoco_rl_manipulate_unknown_objects_amd.models.feature_extractor import AugmentedNatureCNN, _Conv1U8
    osp = spaces.Dict({"observation": spaces.Box(0, 255, shape=(5, 64, 64), dtype=np.uint8)})
    torch.manual_seed(0)
    net = AugmentedNatureCNN(osp).cuda().to(memory_format=torch.channels_last)
    obs = torch.randint(0, 256, (96, 5, 64, 64), dtype=torch.uint8, device="cuda")
    c0 = net.cnn[0]
    G = torch.randn(96, 32, 15, 15, device="cuda").contiguous(memory_format=torch.channels_last)
    c0.zero_grad(set_to_none=True)
    ya, other = _Conv1U8.apply(obs, c0.weight, c0.bias)
    (ya * G).sum().backward()
    gwa, gba = c0.weight.grad.clone(), c0.bias.grad.clone()
    c0.zero_grad(set_to_none=True)
    x = (obs[:, :4].float() / 255.0).contiguous(memory_format=torch.channels_last)
    yb = torch.relu(torch.nn.functional.conv2d(x, c0.weight, c0.bias, stride=4))
    (yb * G).sum().backward()
    gwb, gbb = c0.weight.grad.clone(), c0.bias.grad.clone()
    assert (ya - yb).abs().max() < 2e-5 * (1 + yb.abs().max())
    assert torch.equal(other, obs[:, 4, 0, :2].float() / 255.0)
    assert (gwa - gwb).abs().max() < 1e-4 * gwb.abs().max() and (gba - gbb).abs().max() < 1e-4 * gbb.abs().max()
    feats = {}
    net.fused_trunk_training = False                   # (the all-layers path has its own test: tests/test_gpu_train_kernels.py)
    for fused in (True, False):
        net.fused_first_layer_training = fused
        feats[fused] = net({"observation": obs}).detach()
        assert feats[fused].requires_grad is False
    assert (feats[True] - feats[False]).abs().max() < 2e-5 * (1 + feats[False].abs().max())


def test_merged_heads_rollout_forward_matches_the_module_forward(torch):
    """ActorCriticPolicy.forward_parts with the rollout cache (policy | value MLPs as merged GEMMs, NHWC flatten as a view) against the
    module-by-module forward on the same weights: fp32 both ways, sums re-associated -- 2e-5 on means and values of order 0.1..1. The
    cache follows the parameters: through refresh_rollout_cache() and, outside a stream capture, by itself (version stamps) -- checked by changing them."""
    from mujoco_rl_manipulate_unknown_objects_amd.sb3.policies import ActorCriticPolicy
    from mujoco_rl_manipulate_unknown_objects_amd.models.feature_extractor import AugmentedNatureCNN
    from mujoco_rl_manipulate_unknown_objects_amd.simulation.controller.sensor import RGBDSensor
    from mujoco_rl_manipulate_unknown_objects_amd.simulation.controller.actuator import Actuator
    from mujoco_rl_manipulate_unknown_objects_amd.simulation.environment.robot_env import default_config
    cfg = default_config()
    torch.manual_seed(3)
    pol = ActorCriticPolicy(RGBDSensor(config=cfg).setup_observation_space(), Actuator(config=cfg).setup_action_space(),
                            features_extractor_class=AugmentedNatureCNN, net_arch=[256, 256]).cuda().to(memory_format=torch.channels_last)
    with torch.no_grad():                                   # biases and the action head away from their zero / 0.01 initialisation
        for p in pol.parameters():
            if p.ndim == 1:
                p.add_(0.1 * torch.randn_like(p))
        pol.action_net.weight.mul_(30.0)
    g = torch.Generator(device="cuda"); g.manual_seed(1)
    for n in (1, 130, 1024):
        obs = {"observation": torch.randint(0, 256, (n, 5, 64, 64), dtype=torch.uint8, device="cuda", generator=g)}
        with torch.no_grad():
            ref = pol.forward_parts(obs)
        assert pol._rollout_cache is None and pol.enable_rollout_cache()
        with torch.no_grad():
            got = pol.forward_parts(obs)
        assert got[0].shape == ref[0].shape == (n, 6) and got[2].shape == ref[2].shape == (n,)
        assert (got[0] - ref[0]).abs().max() < 2e-5 and (got[2] - ref[2]).abs().max() < 2e-5 and torch.equal(got[1], ref[1])
        assert ref[0].abs().max() > 0.05 and ref[2].abs().max() > 0.05
        # with autograd on the module path runs (the update never sees the cache)
        assert pol.forward_parts(obs)[0].requires_grad
        pol._rollout_cache = None
    pol.enable_rollout_cache()
    with torch.no_grad():
        before = pol.forward_parts(obs)[0].clone()
        pol.action_net.bias.add_(1.0)
        auto = pol.forward_parts(obs)[0].clone()            # an in-place parameter change is noticed (version stamps): no stale weights
        pol.refresh_rollout_cache()
        fresh = pol.forward_parts(obs)[0]
        sd = {k: v.clone() for k, v in pol.state_dict().items()}; sd["action_net.bias"] = sd["action_net.bias"] - 1.0
        pol.load_state_dict(sd)
        loaded = pol.forward_parts(obs)[0]
    assert (auto - before - 1.0).abs().max() < 1e-5 and (fresh - before - 1.0).abs().max() < 1e-5 and (loaded - before).abs().max() < 1e-5
    # the hand-written optimiser step writes the parameters through raw pointers: it must bump their version counters, or the merged
    # weights of the step before would be used (advisor, round 3)
    from mujoco_rl_manipulate_unknown_objects_amd.engine import ClipAdam
    opt = torch.optim.Adam(pol.parameters(), lr=1e-2, capturable=True)
    for q in pol.parameters():
        q.grad = torch.ones_like(q)
    with torch.no_grad():
        pre = pol.forward_parts(obs)[0].clone()
    assert ClipAdam(opt, 0.5).step()
    with torch.no_grad():
        post = pol.forward_parts(obs)[0].clone()             # merged path, refreshed by itself
        cache, pol._rollout_cache = pol._rollout_cache, None
        ref_post = pol.forward_parts(obs)[0]                 # module path on the stepped parameters
        pol._rollout_cache = cache
    assert (post - pre).abs().max() > 1e-3 and (post - ref_post).abs().max() < 2e-5
    # value-only extractor or unequal MLP shapes: no merged path
    pol2 = ActorCriticPolicy(RGBDSensor(config=cfg).setup_observation_space(), Actuator(config=cfg).setup_action_space(),
                             features_extractor_class=AugmentedNatureCNN, net_arch=dict(pi=[64], vf=[64, 64])).cuda()
    assert not pol2.enable_rollout_cache() and pol2._rollout_cache is None


def test_fused_ppo_loss_matches_the_tensor_formula(torch):
    """grip_ppo_loss (loss + gradients of one minibatch in one launch) against the tensor-library formula PPO._loss_backward uses on
    CPU, fp32 both ways: loss, policy loss, value loss and the gradients w.r.t. mean, values and log_std. Ratios inside and outside the
    clip range, both advantage signs, a saturated log-ratio (+-20 clamp) and ragged sizes."""
    from mujoco_rl_manipulate_unknown_objects_amd.sb3.ppo import _FusedPPOLoss
    from mujoco_rl_manipulate_unknown_objects_amd.sb3.policies import ActorCriticPolicy
    g = torch.Generator(device="cuda"); g.manual_seed(5)
    for n, A, clip, ent, vf in ((4096, 6, 0.2, 0.0, 0.5), (130, 6, 0.1, 0.01, 1.0), (2, 3, 0.2, 0.0, 0.5), (1500, 8, 0.3, 0.02, 0.25)):
        rnd = lambda *s: torch.randn(*s, device="cuda", generator=g)
        mean = (0.3 * rnd(n, A)).requires_grad_(True); log_std = (0.2 * rnd(A)).requires_grad_(True); values = rnd(n).requires_grad_(True)
        actions = mean.detach() + torch.exp(log_std.detach()) * rnd(n, A)
        old = ActorCriticPolicy._log_prob(mean.detach(), log_std.detach(), actions) + 0.3 * rnd(n)       # ratios spread around 1, many clipped
        if n > 100:
            old[:3] -= 30.0; old[3:6] += 30.0                                                           # saturated log-ratios
        adv = rnd(n); ret = rnd(n)
        # reference: the tensor formula
        a = (adv - adv.mean()) / (adv.std() + 1e-8)
        logp = ActorCriticPolicy._log_prob(mean, log_std, actions)
        ratio = torch.exp(torch.clamp(logp - old, -20.0, 20.0))
        pl = -torch.min(a * ratio, a * torch.clamp(ratio, 1 - clip, 1 + clip)).mean()
        vl = torch.nn.functional.mse_loss(ret, values)
        el = -(0.5 + 0.5 * np.log(2 * np.pi) + log_std).sum(-1).expand(n).mean()
        loss = pl + ent * el + vf * vl
        gm, gl, gv = torch.autograd.grad(loss, (mean, log_std, values))
        frac_clipped = ((ratio < 1 - clip) | (ratio > 1 + clip)).float().mean().item()
        assert n < 100 or 0.1 < frac_clipped < 0.9
        loss2, pl2, vl2 = _FusedPPOLoss.apply(mean, log_std, values, actions, old, adv, ret, clip, ent, vf)
        gm2, gl2, gv2 = torch.autograd.grad(loss2, (mean, log_std, values))
        for x, y, name in ((loss, loss2, "loss"), (pl, pl2, "pl"), (vl, vl2, "vl")):
            x, y = float(x.detach()), float(y.detach())
            assert abs(x - y) < 2e-6 * max(1.0, abs(x)), (n, name, x, y)
        for x, y, name in ((gm, gm2, "d mean"), (gl, gl2, "d log_std"), (gv, gv2, "d values")):
            assert (x - y).abs().max() < 2e-6 * max(1.0, float(x.abs().max())) + 1e-9, (n, name, float((x - y).abs().max()), float(x.abs().max()))


def test_fused_second_and_third_convolution_match_the_tensor_library(torch):
    """grip_conv23 (conv2 + ReLU + conv3 + ReLU of AugmentedNatureCNN, y2 kept in LDS; since round 5 k_conv23_b3: bf16 matrix pipe, both operands as three bf16
    terms -- fp32-equivalent) against torch.nn.functional on the same weights: contiguous and channels-last weight layouts, batch sizes that are odd, not multiples of
    the image pair a trip takes, fewer and more pairs than the persistent workgroups (one per CU), and an exact-integer case that pins the operand / accumulator lane
    maps (small integers split exactly, so the result is bit-exact). 2e-5 relative to the activations' scale."""
    import torch.nn.functional as F
    from mujoco_rl_manipulate_unknown_objects_amd.engine import conv23, conv23_prep
    g = torch.Generator(device="cuda"); g.manual_seed(2)
    rnd = lambda *s: torch.randn(*s, device="cuda", generator=g)
    for n in (1, 3, 4, 130, 1024, 1537):
        y1 = torch.relu(rnd(n, 32, 15, 15)).contiguous(memory_format=torch.channels_last)
        w2 = rnd(64, 32, 4, 4) / 22.0; b2 = 0.1 * rnd(64); w3 = rnd(64, 64, 3, 3) / 24.0; b3 = 0.1 * rnd(64)
        ref = F.relu(F.conv2d(F.relu(F.conv2d(y1, w2, b2, stride=2)), w3, b3))
        for cl in (False, True):
            w2x = w2.contiguous(memory_format=torch.channels_last) if cl else w2
            w3x = w3.contiguous(memory_format=torch.channels_last) if cl else w3
            mats = conv23_prep(w2x, w3x)
            out = conv23(y1, mats[0], b2, mats[1], b3)
            assert out.shape == (n, 64, 4, 4) and out.is_contiguous(memory_format=torch.channels_last)
            assert (out - ref).abs().max() < 2e-5 * max(1.0, float(ref.abs().max())), (n, cl, float((out - ref).abs().max()), float(ref.abs().max()))
        assert float(ref.abs().max()) > 0.5 and float((ref > 0).float().mean()) > 0.2
    # exact integers: small integer activations and one-hot-ish integer weights, every product and sum exact in fp32
    y1 = torch.randint(0, 4, (6, 32, 15, 15), device="cuda", generator=g).float().contiguous(memory_format=torch.channels_last)
    w2 = torch.randint(-1, 2, (64, 32, 4, 4), device="cuda", generator=g).float(); w3 = torch.randint(-1, 2, (64, 64, 3, 3), device="cuda", generator=g).float()
    b2 = torch.randint(-3, 4, (64,), device="cuda", generator=g).float(); b3 = torch.randint(-3, 4, (64,), device="cuda", generator=g).float()
    ref = F.relu(F.conv2d(F.relu(F.conv2d(y1, w2, b2, stride=2)), w3, b3))
    mats = conv23_prep(w2, w3)
    assert torch.equal(conv23(y1, mats[0], b2, mats[1], b3), ref)
    # rewriting the operand matrices in place keeps their addresses
    p0 = mats[0].data_ptr(); mats2 = conv23_prep(w2 * 2, w3, *mats)
    assert mats2[0].data_ptr() == p0 and torch.equal(conv23(y1, mats2[0], b2, mats2[1], b3), F.relu(F.conv2d(F.relu(F.conv2d(y1, 2 * w2, b2, stride=2)), w3, b3)))
