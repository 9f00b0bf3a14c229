"""-m gpu: mixed-object batches (BASELINE.json configs[3]: several object models and both target directions in one
process, envs sorted by group). Every group must behave exactly like a stand-alone batch of its object, and like the
CPU oracle of that object and direction."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu

GROUPS = [("sand_ball", 24, (1.0, 0.0)), ("sugar_cube", 17, (1.0, 1.0)), ("bread_crumb", 16, (1.0, 0.0)), ("acorn", 7, (1.0, 1.0))]


@pytest.fixture(scope="module")
def torch():
    import torch as t
    if not t.cuda.is_available():
        pytest.skip("no GPU")
    return t


@pytest.fixture(scope="module")
def engine(torch):
    from mujoco_rl_manipulate_unknown_objects_amd import engine as e
    e.lib()
    return e


def test_mixed_batch_equals_its_groups_run_alone(engine, torch):
    """Ragged groups (not multiples of a wavefront's 4 envs), three macro steps and the observation: bit-identical to one
    engine.Batch per group fed the same action rows; masked reset touches only the masked envs of the right group."""
    mb = engine.MixedBatch(GROUPS, auto_reset=1)
    solo = [engine.Batch(o, n, target_dir=d, auto_reset=1) for o, n, d in GROUPS]
    n = mb.n
    assert n == sum(g[1] for g in GROUPS) and mb.offsets[-1] == n
    mb.reset()
    for b in solo:
        b.reset()
    rng = np.random.default_rng(5)
    for t in range(3):
        acts = torch.from_numpy(rng.uniform(-1, 1, (n, 6)).astype(np.float32)).cuda()
        out = mb.step(acts)
        obs = mb.observe()
        torch.cuda.synchronize()
        for g, b in enumerate(solo):
            lo, hi = mb.offsets[g], mb.offsets[g + 1]
            o = b.step(acts[lo:hi]); ob = b.observe(); torch.cuda.synchronize()
            for k in o:
                assert torch.equal(o[k], out[k][lo:hi]), (t, g, k)
            assert torch.equal(ob, obs[lo:hi]), (t, g)
    assert (out["fault"] == 0).all()
    # the per-env direction reaches the goals: desired_goal of a 45-degree group moves along (1, 1)
    dg = out["desired_goal"].cpu().numpy(); td = mb.target_dirs.cpu().numpy()
    assert np.allclose(td[:24], [1, 0]) and np.allclose(td[24:41], [1, 1])
    # masked reset: one env in the second group
    mask = torch.zeros(n, dtype=torch.uint8); mask[30] = 1
    q_before = mb.get_state()[0]
    mb.reset(mask.cuda()); torch.cuda.synchronize()
    q_after = mb.get_state()[0]
    changed = np.where(np.abs(q_after - q_before).max(axis=1) > 0)[0]
    assert list(changed) == [30]
    es = mb.get_flags()[0]
    assert es[30] == 0 and (np.delete(es, 30) == 3).all()
    mb.close()
    for b in solo:
        b.close()
    assert dg.shape == (n, 2)


def test_mixed_batch_against_the_oracle(engine, orc, torch):
    """First two macro steps from reset of every env of a mixed batch vs the oracle of that env's object and direction
    (the checks of tests/test_gpu_parity.py::test_macro_step_parity; tolerances stated below)."""
    groups = [(o, 6, d) for o, _, d in GROUPS]
    mb = engine.MixedBatch(groups)
    mb.reset()
    envs = []
    for o, n, d in groups:
        m = orc.Model(o)
        for _ in range(n):
            e = orc.EnvOracle(m, target_dir=d); e.reset(); envs.append(e)
    rng = np.random.default_rng(11)
    for t in range(2):
        acts = rng.uniform(-1, 1, (mb.n, 6)).astype(np.float32)
        out = mb.step(torch.from_numpy(acts).cuda()); torch.cuda.synchronize()
        o_np = {k: v.cpu().numpy() for k, v in out.items()}
        assert (o_np["fault"] == 0).all()
        for i, e in enumerate(envs):
            o = e.step(acts[i])
            assert o.n_substeps == o_np["n_substeps"][i], (t, i)
            assert (o.done, o.status, o.episode_step, o.gripper_open, o.object_grasped) == \
                (o_np["done"][i], o_np["status"][i], o_np["episode_step"][i], o_np["gripper_open"][i], o_np["object_grasped"][i])
            tolp = 1e-5 if t == 0 else 1e-4
            assert np.abs(np.array(o.gripper_pos) - o_np["gripper_position"][i]).max() < tolp
            # an object the gripper has knocked is still rocking at the end of the second step: fp32 and fp64 contact
            # sequences have separated by then, so only the first step is held to the tight bound (reward = 30 x progress)
            tolo = 5e-5 if t == 0 else 2e-3
            assert np.abs(np.array(o.final_obj_pos) - o_np["object_position"][i]).max() < tolo
            assert np.abs(np.array(o.desired_goal) - o_np["desired_goal"][i]).max() < tolo
            assert abs(o.reward - o_np["reward"][i]) < (1e-3 if t == 0 else 30 * tolo)
    mb.close()


def test_batch_set_rules(engine, torch):
    """A set needs one action / observation layout; the capacity of a merged ready list is a multiple of the group count; a set
    of one batch is that batch."""
    import ctypes as C
    a = engine.Batch("sand_ball", 8); b = engine.Batch("sugar_cube", 8, include_roll=0)
    ptr = C.c_void_p(); arr = (C.c_void_p * 2)(a.ptr, b.ptr)
    assert engine.lib().grip_batchset_create(arr, 2, None, C.byref(ptr)) != 0
    assert b"include_roll" in engine.lib().grip_last_error()
    a.close(); b.close()
    mb = engine.MixedBatch([("sand_ball", 8, (1, 0)), ("sugar_cube", 8, (1, 1)), ("acorn", 8, (1, 0))])
    lst = torch.full((8,), -1, dtype=torch.int32, device="cuda"); cnt = torch.zeros(1, dtype=torch.int32, device="cuda")
    with pytest.raises(engine.GripError):
        mb.advance(torch.zeros(8, 6, device="cuda"), 16, lst, cnt)          # 8 rows for 3 groups
    mb.close()
    one = engine.MixedBatch([("sand_ball", 9, (1, 0))]); solo = engine.Batch("sand_ball", 9)
    acts = torch.rand(9, 6, device="cuda") * 2 - 1
    o1 = one.step(acts); o2 = solo.step(acts); torch.cuda.synchronize()
    for k in o1:
        assert torch.equal(o1[k], o2[k]), k
    assert torch.equal(one.observe(), solo.observe())
    one.close(); solo.close()


FIELDS = ["reward", "done", "achieved_goal", "desired_goal", "status", "episode_step", "gripper_open", "object_grasped",
          "position_reached", "total_distance", "line_distance", "gripper_position", "object_position", "init_obj_pos",
          "n_substeps", "fault"]


def test_mixed_time_sliced_equals_lockstep(engine, torch):
    """The time-sliced schedule over a mixed batch: every group lists its finished envs in its own segment of the ready list
    (holes are -1, the count is the capacity). Each env plays its own action sequence; every macro step's outputs and the
    final state are bit-identical to the lock-step mixed batch."""
    groups = [("sand_ball", 20, (1.0, 0.0)), ("sugar_cube", 17, (1.0, 1.0)), ("bread_crumb", 12, (1.0, 0.0))]
    steps, cap, slice_len = 3, 24, 37
    ref_b = engine.MixedBatch(groups)
    n = ref_b.n
    rng = np.random.default_rng(13)
    actions = rng.uniform(-1, 1, (steps, n, 6)).astype(np.float32); actions[:, :, 0] = np.abs(actions[:, :, 0])
    ref = []
    for t in range(steps):
        o = ref_b.step(torch.from_numpy(actions[t]).cuda()); torch.cuda.synchronize()
        ref.append({k: o[k].cpu().numpy().copy() for k in FIELDS})
    ref_state = ref_b.get_state(); ref_b.close()

    mb = engine.MixedBatch(groups)
    lst = torch.full((cap,), -1, dtype=torch.int32, device="cuda"); cnt = torch.zeros(1, dtype=torch.int32, device="cuda")
    slot_act = torch.zeros(cap, 6, device="cuda")
    obs = torch.zeros(cap, 5, 64, 64, dtype=torch.uint8, device="cuda")
    given = np.zeros(n, int); got = [[None] * steps for _ in range(n)]; final_state = [None] * n
    seg = cap // len(groups)
    for tick in range(20000):
        out = mb.advance(slot_act, slice_len, lst, cnt)
        mb.observe_list(lst, cnt, obs)
        torch.cuda.synchronize()
        assert int(cnt.item()) == cap
        ids = lst.cpu().numpy()
        listed = ids[ids >= 0]
        assert len(set(listed.tolist())) == len(listed)
        for g in range(len(groups)):                                   # a segment lists only envs of its group, packed to the front
            s_ = ids[g * seg:(g + 1) * seg]; k = int((s_ >= 0).sum())
            assert (s_[:k] >= mb.offsets[g]).all() and (s_[:k] < mb.offsets[g + 1]).all() and (s_[k:] == -1).all()
        o_host = {k: out[k].cpu().numpy() for k in FIELDS}
        st = mb.get_state() if len(listed) else None
        full = mb.observe() if len(listed) else None
        new_act = np.zeros((cap, 6), np.float32)
        for r in range(cap):
            e = int(ids[r])
            if e < 0:
                continue
            assert torch.equal(obs[r], full[e])                         # the listed env's observation sits in its row
            if given[e] > 0 and got[e][given[e] - 1] is None:
                got[e][given[e] - 1] = {k: o_host[k][e].copy() for k in FIELDS}
                if given[e] == steps:
                    final_state[e] = [a[e].copy() for a in st]
            if given[e] < steps:
                new_act[r] = actions[given[e], e]; given[e] += 1
        slot_act.copy_(torch.from_numpy(new_act))
        if all(g_[steps - 1] is not None for g_ in got):
            break
    else:
        raise AssertionError("envs did not finish")
    for e in range(n):
        for t in range(steps):
            for k in FIELDS:
                assert np.array_equal(got[e][t][k], ref[t][k][e]), (e, t, k)
        for a, b_ in zip(final_state[e], ref_state):
            assert np.array_equal(a, b_[e]), e
    mb.close()


def test_mixed_env_trains_time_sliced(torch):
    """PPO over the time-sliced schedule of a mixed batch (the tick, one launch per phase over the eight groups, captured as a hipGraph)."""
    from mujoco_rl_manipulate_unknown_objects_amd.simulation.environment.robot_env import MixedBatchedRobotEnv, default_config
    from mujoco_rl_manipulate_unknown_objects_amd.sb3 import PPO, GpuVecEnv
    from mujoco_rl_manipulate_unknown_objects_amd.models.feature_extractor import AugmentedNatureCNN
    env = MixedBatchedRobotEnv(default_config(), envs_per_group=16, auto_reset=True)
    model = PPO("MultiInputPolicy", GpuVecEnv(env), n_steps=2, batch_size=128, n_epochs=1, seed=3, async_slice=64, async_capacity=32,
                async_budget_us=2000, policy_kwargs=dict(features_extractor_class=AugmentedNatureCNN, share_features_extractor=True, net_arch=[256, 256]))
    ar = model._async
    assert ar is not None and ar.C == 32
    before = [p.detach().clone() for p in model.policy.parameters()]
    for _ in range(3):
        assert model.collect_rollouts()
        stats = model.train()
    torch.cuda.synchronize()
    assert ar._graph is not None                                        # the tick ran as a captured graph
    assert np.isfinite(float(stats["loss"])) and model.num_timesteps >= 3 * 2 * 128
    rows = ar.window_rows()
    rec = rows[ar.is_rec[rows]]
    envs = ar.rec_env[rec]
    assert envs.min().item() >= 0 and envs.max().item() < 128 and len(set((envs // 16).tolist())) == 8     # every group decides
    assert any(not torch.equal(a, b) for a, b in zip(before, model.policy.parameters()))
    env.close()


def test_mixed_env_trains_with_ppo(torch):
    """MixedBatchedRobotEnv behind GpuVecEnv: one lock-step PPO rollout + update over 4 objects x 2 directions."""
    from mujoco_rl_manipulate_unknown_objects_amd.simulation.environment.robot_env import MixedBatchedRobotEnv, default_config
    from mujoco_rl_manipulate_unknown_objects_amd.sb3 import PPO, GpuVecEnv
    from mujoco_rl_manipulate_unknown_objects_amd.models.feature_extractor import AugmentedNatureCNN
    env = MixedBatchedRobotEnv(default_config(), envs_per_group=8, auto_reset=True)
    assert env.n_envs == 64 and env.target_direction.shape == (64, 2)
    model = PPO("MultiInputPolicy", GpuVecEnv(env), n_steps=2, batch_size=64, n_epochs=1, seed=3,
                policy_kwargs=dict(features_extractor_class=AugmentedNatureCNN, share_features_extractor=True, net_arch=[256, 256]))
    assert model._async is None                      # a mixed batch steps in lock-step
    before = [p.detach().clone() for p in model.policy.parameters()]
    model.learn(total_timesteps=2 * 64)
    torch.cuda.synchronize()
    assert any(not torch.equal(a, b) for a, b in zip(before, model.policy.parameters()))
    assert all(torch.isfinite(p).all() for p in model.policy.parameters())
    env.close()


def test_random_episodes_resemble_the_reference_training_logs(engine, torch):
    """The one piece of physics-level data the reference holds: the Monitor CSVs of its published runs. The episodes before
    the first evaluation (training step 2000) are played by a freshly initialised SAC actor (near-random actions) in the real
    MuJoCo simulation (tests/golden/reference_monitor_early_episodes.json, tools/make_monitor_fixture.py). First episodes of
    this engine under uniformly random actions, same four objects and two directions, must show the same statistics: how
    often an episode survives to the 400-step time limit instead of ending in Status.FAIL, how long episodes last, how much
    progress reward random pushing collects. A statistical band (about three standard errors of the 51 reference episodes
    plus the policy mismatch), not a bit-level pin: reference 43 % at the limit / mean length 301 / mean return 1.24; this
    engine measured 37 % / 271 / 1.16 over 1024 episodes (profiles/r01_episode_stats_uniform.json)."""
    import json
    import os
    ref = json.load(open(os.path.join(os.path.dirname(__file__), "golden", "reference_monitor_early_episodes.json")))["runs"]
    ref_len = np.array([l for r in ref.values() for l in r["lengths"]]); ref_ret = np.array([x for r in ref.values() for x in r["returns"]])
    assert len(ref_len) == 51 and ref_len.max() == 400          # time_horizon 400 (base_config.py) is what ends the longest ones

    per = 48
    groups = [(k.rsplit("_dir", 1)[0], per, (1.0, 0.0) if k.endswith("dir0") else (1.0, 1.0)) for k in ref]
    mb = engine.MixedBatch(groups, auto_reset=1)
    mb.reset()
    n = mb.n
    g = torch.Generator(device="cuda"); g.manual_seed(7)
    ret = torch.zeros(n, device="cuda"); first_ret = torch.zeros(n, device="cuda")
    first_len = torch.zeros(n, dtype=torch.int32, device="cuda"); open_ = torch.ones(n, dtype=torch.bool, device="cuda")
    for t in range(1, 401):
        out = mb.step(torch.rand(n, 6, device="cuda", generator=g) * 2 - 1)
        ret += out["reward"]
        d = out["done"].bool() & open_
        first_ret[d] = ret[d]; first_len[d] = t
        open_ &= ~d
    torch.cuda.synchronize()
    assert not open_.any()                                       # every env ended its first episode by the time limit
    ln = first_len.cpu().numpy(); rt = first_ret.cpu().numpy()
    assert ln.max() == 400 and ln.min() >= 1
    frac_ref, frac = (ref_len == 400).mean(), (ln == 400).mean()
    assert abs(frac - frac_ref) < 0.2, (frac, frac_ref)
    assert abs(ln.mean() - ref_len.mean()) < 60, (ln.mean(), ref_len.mean())
    assert abs(np.median(ln) - np.median(ref_len)) < 90, (np.median(ln), np.median(ref_len))
    assert 0.5 * ref_ret.mean() < rt.mean() < 2.0 * ref_ret.mean(), (rt.mean(), ref_ret.mean())
    assert rt.min() >= 0.0 and rt.max() < 60.0                   # progress reward is never negative (reward.py:18-41)
    mb.close()


def test_mixed_env_renders_every_group_with_its_own_model(torch):
    """RobotEnv.render on a mixed batch (robot_env.py:302-340): the global env index is mapped to its (object, direction) group's batch and
    the camera pose is built from THAT group's model (object inertial frame, static cameras). The gripper camera's view at zoom 1 equals the
    env's own observation RGB; the tracking cameras look at the env's own object; an env of another object renders a different image."""
    from mujoco_rl_manipulate_unknown_objects_amd.simulation.environment.robot_env import MixedBatchedRobotEnv, BatchedRobotEnv, default_config
    cfg = default_config(camera_id=2, rendering_zoom_width=1, rendering_zoom_height=1)
    env = MixedBatchedRobotEnv(cfg, envs_per_group=4)
    obs = env.reset()["observation"].cpu().numpy()
    frames = {}
    for e in (0, 9, 17, 30):                                   # acorn dir 0, sand_ball dir 0 (group 2), sugar_cube dir 0, bread_crumb dir 45
        img = env.render("rgb_array", env_index=e)
        assert img.shape == (64, 64, 3) and img.dtype == np.uint8
        same = (np.abs(img.astype(int) - obs[e, :3].transpose(1, 2, 0).astype(int)) <= 1).mean()
        assert same > 0.99, (e, same)
        frames[e] = img
    env.config.camera_id = 0                                   # workbench camera: tracks the env's own object
    wb = {e: env.render("rgb_array", env_index=e) for e in (0, 9, 17, 30)}
    assert all(w.shape == (64, 64, 3) for w in wb.values())
    assert any((wb[0] != wb[e]).any() for e in (9, 17, 30))    # different objects, different pictures
    # against the single-object env of the same group: identical pixels
    solo = BatchedRobotEnv(default_config(sim_env="/xmls/sugar_cube_env.xml", camera_id=0, rendering_zoom_width=1, rendering_zoom_height=1), n_envs=1)
    solo.reset()
    assert np.array_equal(solo.render("rgb_array", env_index=0), wb[17])
    depth = env.render("depth_array", env_index=30)
    assert depth.shape == (64, 64) and np.isfinite(depth.astype(np.float64)).all()
    solo.close(); env.close()
