"""-m gpu: time-sliced stepping (grip_batch_advance) against lock-step grip_batch_step. Same arithmetic per env, other
schedule: every per-env output has to be bit-identical."""
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

FIELDS = ["reward", "done", "achieved_goal", "desired_goal", "status", "episode_step", "gripper_open", "object_grasped",
          "position_reached", "total_distance", "line_distance", "gripper_position", "object_position", "init_obj_pos",
          "n_substeps", "fault"]


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


def lockstep_reference(engine, torch, obj, n, actions, steps):
    b = engine.Batch(obj, n)
    outs = []
    for t in range(steps):
        o = b.step(torch.from_numpy(actions[t]).cuda())
        torch.cuda.synchronize()
        outs.append({k: o[k].cpu().numpy().copy() for k in FIELDS})
    state = b.get_state()
    b.close()
    return outs, state


@pytest.mark.parametrize("capacity,slice_len", [(64, 37), (24, 16), (64, 2000)])
def test_sliced_macro_steps_equal_lockstep(engine, torch, capacity, slice_len):
    """Each env does `steps` macro steps with its own action sequence, served through a ready list of `capacity` rows
    (smaller than the batch: envs queue) in slices of `slice_len` physics steps. Outputs of every macro step and the
    final state are bit-identical to the lock-step batch."""
    n, steps, obj = 64, 3, "sand_ball"
    rng = np.random.default_rng(3)
    actions = rng.uniform(-1, 1, (steps, n, 6)).astype(np.float32)
    actions[:, :, 0] = np.abs(actions[:, :, 0])
    ref, ref_state = lockstep_reference(engine, torch, obj, n, actions, steps)

    b = engine.Batch(obj, n)
    lst = torch.full((capacity,), -1, dtype=torch.int32, device="cuda"); cnt = torch.zeros(1, dtype=torch.int32, device="cuda")
    slot_act = torch.zeros(capacity, 6, device="cuda")
    given = np.zeros(n, int)            # actions handed to env so far
    got = [[None] * steps for _ in range(n)]
    final_state = [None] * n
    parked = np.zeros(n, bool)
    for tick in range(20000):
        out = b.advance(slot_act, slice_len, lst, cnt)
        torch.cuda.synchronize()
        c = int(cnt.item()); ids = lst.cpu().numpy()
        assert (ids[c:] == -1).all() and len(set(ids[:c].tolist())) == c
        o_host = {k: out[k].cpu().numpy() for k in FIELDS}
        st = b.get_state() if c else None
        new_act = np.zeros((capacity, 6), np.float32)
        for r in range(c):
            e = int(ids[r])
            if given[e] > 0 and got[e][given[e] - 1] is None:           # just finished macro step number given[e]
                got[e][given[e] - 1] = {k: o_host[k][e].copy() for k in FIELDS}
                if given[e] == steps:
                    final_state[e] = [a[e].copy() for a in st]
            if given[e] < steps:
                new_act[r] = actions[given[e], e]; given[e] += 1
            else:
                parked[e] = True                                           # gets a zero action: result ignored
        slot_act.copy_(torch.from_numpy(new_act))
        if all(g[steps - 1] is not None for g in got):
            break
    else:
        raise AssertionError("envs did not finish")
    for e in range(n):
        for t in range(steps):
            for k in FIELDS:
                assert np.array_equal(got[e][t][k], ref[t][k][e]), (e, t, k, got[e][t][k], ref[t][k][e])
        for a, r in zip(final_state[e], ref_state):
            assert np.array_equal(a, r[e])
    b.close()


def test_observe_list_rows_match_full_observation(engine, torch):
    n, cap = 48, 16
    b = engine.Batch("sugar_cube", n)
    rng = np.random.default_rng(1)
    b.step(torch.from_numpy(rng.uniform(-1, 1, (n, 6)).astype(np.float32)).cuda())
    full = b.observe()
    ids = torch.tensor([5, 47, 0, 13, 22], dtype=torch.int32, device="cuda")
    lst = torch.full((cap,), -1, dtype=torch.int32, device="cuda"); lst[:5] = ids
    cnt = torch.tensor([5], dtype=torch.int32, device="cuda")
    rows = torch.full((cap, 5, 64, 64), 77, dtype=torch.uint8, device="cuda")
    b.observe_list(lst, cnt, rows)
    torch.cuda.synchronize()
    assert torch.equal(rows[:5], full[ids.long()])
    assert bool((rows[5:] == 77).all())          # rows beyond the count are not touched
    b.close()


@pytest.mark.parametrize("pipeline", [False, True])
def test_fused_recorder_and_graph_replay_match_tensor_op_bookkeeping(engine, torch, pipeline):
    """AsyncRollout three ways on identical batches with a deterministic policy: tensor-op bookkeeping (the CPU-tested
    reference), the fused HIP recorder, and the fused recorder replayed from a captured hipGraph. Same records, same GAE.
    pipeline=True: decisions run on a side stream while the next slice advances (lag 2) -- another schedule, checked the same way."""
    from mujoco_rl_manipulate_unknown_objects_amd.sb3.async_rollout import AsyncRollout, BatchEngineAdapter
    from mujoco_rl_manipulate_unknown_objects_amd.simulation.environment.robot_env import BatchedRobotEnv, default_config

    def policy(rows):
        x = rows.float().mean(dim=(1, 2, 3)); pad = rows[:, 4, 0, :2].float().sum(1)
        k = torch.arange(1, 7, device=rows.device).float()
        act = 1.3 * torch.cos(x[:, None] * k[None, :] * 0.37 + pad[:, None])
        act[:, 0] = act[:, 0].abs()
        return act, x / 255.0, -x / 100.0

    def run(fused, graph):
        env = BatchedRobotEnv(default_config(sim_env="/xmls/sand_ball_env.xml"), n_envs=96, device_index=0, auto_reset=True)
        ro = AsyncRollout(BatchEngineAdapter(env), policy, target=96 * 3, capacity=32, slice_len=24, gamma=0.99, gae_lambda=0.95,
                          action_low=[-1] * 6, action_high=[1] * 6, poll_every=2, use_graph=graph, fused=fused, pipeline=pipeline)
        res = []
        for _ in range(2):
            n = ro.collect()
            torch.cuda.synchronize()
            keep = ro.window_rows()
            res.append(dict(n=n, ticks=ro.tick, **{k: getattr(ro, k)[keep].clone() for k in
                            ("rewards", "dones", "next_rec", "completed", "is_rec", "rec_env", "actions", "log_probs", "values", "advantages", "returns", "obs")}))
        st = ro.stats(); sub = int(ro.substeps_total.item())
        env.close()
        return res, st, sub

    ref, st_ref, sub_ref = run(fused=False, graph=False)
    for fused, graph in ((True, False), (True, True)):
        got, st, sub = run(fused, graph)
        assert sub == sub_ref and st["episodes"] == st_ref["episodes"] and abs(st["ep_rew_mean"] - st_ref["ep_rew_mean"]) < 1e-4
        for a, b in zip(ref, got):
            assert a["n"] == b["n"] and a["ticks"] == b["ticks"]
            m = a["is_rec"]
            assert torch.equal(m, b["is_rec"]) and torch.equal(a["completed"], b["completed"])
            c = a["completed"] & m
            for k in ("rec_env", "actions", "log_probs", "values", "obs"):
                assert torch.equal(a[k][m], b[k][m]), k
            for k in ("rewards", "dones", "next_rec"):
                assert torch.equal(a[k][c], b[k][c]), k
            for k in ("advantages", "returns"):
                assert torch.allclose(a[k][c], b[k][c], rtol=1e-5, atol=1e-6), k


def test_intrinsic_reward_lockstep_and_async(engine, torch):
    """--im_reward (reward.py:44-77): the GPU adds intrinsic_reward(old_obs, new_obs) to the progress reward. Checked against
    the C oracle on the rendered observations, for the lock-step env and for the decision records of the async rollout."""
    from oracle import orc
    from mujoco_rl_manipulate_unknown_objects_amd.sb3.async_rollout import AsyncRollout, BatchEngineAdapter
    from mujoco_rl_manipulate_unknown_objects_amd.simulation.environment.robot_env import BatchedRobotEnv, default_config
    n = 24
    rng = np.random.default_rng(5)
    plain = BatchedRobotEnv(default_config(sim_env="/xmls/sugar_cube_env.xml"), n_envs=n, device_index=0)
    novel = BatchedRobotEnv(default_config(sim_env="/xmls/sugar_cube_env.xml", im_reward=True), n_envs=n, device_index=0)
    o0 = plain.reset()["observation"].clone(); novel.reset()
    for t in range(2):
        a = torch.from_numpy(rng.uniform(-1, 1, (n, 6)).astype(np.float32)).cuda()
        o1, r_plain, _, _ = plain.step(a); o1 = o1["observation"].clone(); r_plain = r_plain.clone()
        _, r_novel, _, _ = novel.step(a)
        torch.cuda.synchronize()
        for i in range(n):
            want = float(r_plain[i]) + orc.intrinsic_reward(o0[i].cpu().numpy(), o1[i].cpu().numpy(), True)
            assert abs(float(r_novel[i]) - want) < 1e-4 * max(1.0, abs(want)), (t, i, float(r_novel[i]), want)
        o0 = o1
    plain.close(); novel.close()

    def policy(rows):
        x = rows.float().mean(dim=(1, 2, 3))
        k = torch.arange(1, 7, device=rows.device).float()
        act = torch.cos(x[:, None] * k[None, :] * 0.37); act[:, 0] = act[:, 0].abs()
        return act, x / 255.0, -x / 100.0

    def run(im):
        env = BatchedRobotEnv(default_config(sim_env="/xmls/sugar_cube_env.xml", im_reward=im), n_envs=32, device_index=0, auto_reset=True)
        ro = AsyncRollout(BatchEngineAdapter(env), policy, target=64, capacity=16, slice_len=24, gamma=0.99, gae_lambda=0.95,
                          action_low=[-1] * 6, action_high=[1] * 6, poll_every=2)
        ro.collect(); torch.cuda.synchronize()
        keep = ro.window_rows()
        res = {k: getattr(ro, k)[keep].clone().cpu() for k in ("rewards", "completed", "is_rec", "next_rec", "obs")}
        res["next_rec"] = res["next_rec"] - ro.carry0          # window-relative ids
        env.close()
        return res
    base, nov = run(False), run(True)
    assert torch.equal(base["completed"], nov["completed"]) and torch.equal(base["obs"], nov["obs"])
    idx = (base["completed"] & base["is_rec"]).nonzero().flatten().tolist()
    assert len(idx) >= 64
    for r in idx[:40]:
        nr = int(base["next_rec"][r])
        want = float(base["rewards"][r]) + orc.intrinsic_reward(base["obs"][r].numpy(), base["obs"][nr].numpy(), True)
        assert abs(float(nov["rewards"][r]) - want) < 1e-4 * max(1.0, abs(want)), (r, float(nov["rewards"][r]), want)


def test_fused_gaussian_head_matches_tensor_ops(engine, torch):
    """Recorder kernel with the fused Gaussian head (action = mean + exp(log_std) * noise, its log-probability) against the
    same rollout with sampling and scoring done by tensor ops on the same noise stream."""
    import math
    from mujoco_rl_manipulate_unknown_objects_amd.sb3.async_rollout import AsyncRollout, BatchEngineAdapter
    from mujoco_rl_manipulate_unknown_objects_amd.simulation.environment.robot_env import BatchedRobotEnv, default_config
    log_std = torch.tensor([-0.5, -0.2, 0.0, 0.1, -1.0, 0.3], device="cuda")

    def parts(rows):
        x = rows.float().mean(dim=(1, 2, 3))
        k = torch.arange(1, 7, device=rows.device).float()
        return 0.5 * torch.cos(x[:, None] * k[None, :] * 0.37), log_std, x / 255.0

    def sampled(rows):
        mean, ls, v = parts(rows)
        z = torch.randn_like(mean)
        return mean + ls.exp() * z, v, (-0.5 * z * z - ls - 0.5 * math.log(2 * math.pi)).sum(-1)

    def run(fused_head):
        torch.manual_seed(123)
        env = BatchedRobotEnv(default_config(sim_env="/xmls/sand_ball_env.xml"), n_envs=48, device_index=0, auto_reset=True)
        ro = AsyncRollout(BatchEngineAdapter(env), sampled, policy_parts_fn=parts if fused_head else None, target=96, capacity=16, slice_len=24,
                          gamma=0.99, gae_lambda=0.95, action_low=[-1] * 6, action_high=[1] * 6, poll_every=2, use_graph=False, fused=True)
        ro.collect(); torch.cuda.synchronize()
        keep = ro.window_rows()
        res = {k: getattr(ro, k)[keep].clone() for k in ("actions", "log_probs", "values", "is_rec", "rewards", "completed")}
        env.close()
        return res
    a, b = run(False), run(True)
    # expf / fma rounding differs in the last bit, the envs then part ways chaotically: compare the decisions of the first ticks
    first = slice(48, 48 + 3 * 16)
    m = a["is_rec"][first]
    assert torch.equal(m, b["is_rec"][first]) and int(m.sum()) >= 32
    assert torch.allclose(a["actions"][first][m], b["actions"][first][m], rtol=1e-5, atol=1e-5)
    assert torch.allclose(a["log_probs"][first][m], b["log_probs"][first][m], rtol=1e-5, atol=1e-4)
    assert torch.allclose(a["values"][first][m], b["values"][first][m], rtol=1e-5, atol=1e-5)


@pytest.mark.gpu
def test_observations_rendered_once_into_the_record_rows():
    """Decision phase of the time-sliced trainer: grip_batch_observe_list with obs_dev = NULL renders the listed envs ONLY into the trainer's
    record rows (from the device-side row base), and grip_conv1_u8_rows reads them there: same bytes as the staged copy, same first-layer
    output bit for bit, and the PPO tick takes that path (no staging write) when its policy can."""
    import torch
    from mujoco_rl_manipulate_unknown_objects_amd import engine
    from mujoco_rl_manipulate_unknown_objects_amd.simulation.environment.robot_env import BatchedRobotEnv, default_config
    from mujoco_rl_manipulate_unknown_objects_amd.sb3 import PPO, GpuVecEnv
    from mujoco_rl_manipulate_unknown_objects_amd.models.feature_extractor import AugmentedNatureCNN
    n, cap = 96, 32
    b = engine.Batch("sugar_cube", n, auto_reset=1)
    lst = torch.full((cap,), -1, dtype=torch.int32, device="cuda"); cnt = torch.zeros(1, dtype=torch.int32, device="cuda")
    g = torch.Generator(device="cuda"); g.manual_seed(0)
    for _ in range(6):
        b.advance(torch.rand(cap, 6, device="cuda", generator=g) * 2 - 1, 64, lst, cnt)
    c = int(cnt.item()); assert c > 0
    stage = torch.zeros(cap, 5, 64, 64, dtype=torch.uint8, device="cuda")
    rec_a = torch.full((200, 5, 64, 64), 7, dtype=torch.uint8, device="cuda"); rec_b = rec_a.clone()
    row = torch.tensor([61], dtype=torch.int64, device="cuda")
    b.observe_list(lst, cnt, stage, rec_a, row)                  # both destinations
    b.observe_list(lst, cnt, None, rec_b, row)                   # record rows only
    torch.cuda.synchronize()
    assert torch.equal(rec_a, rec_b) and torch.equal(rec_b[61:61 + c], stage[:c]) and (rec_b[:61] == 7).all() and (rec_b[61 + cap:] == 7).all()
    assert int((stage[:c, :3] != 0).sum()) > 1000                # real pictures
    w = torch.randn(32, 4, 8, 8, device="cuda"); bias = torch.randn(32, device="cuda")
    y0, o0 = engine.conv1_u8(rec_b[61:61 + cap].contiguous(), w, bias)
    y1, o1 = engine.conv1_u8(engine.RecordRows(rec_b, row, cap), w, bias)
    assert torch.equal(y0, y1) and torch.equal(o0, o1)
    b.close()
    env = BatchedRobotEnv(default_config(sim_env="/xmls/sand_ball_env.xml"), n_envs=64, auto_reset=True)
    model = PPO("MultiInputPolicy", GpuVecEnv(env), n_steps=2, batch_size=64, n_epochs=1, async_slice=32, async_capacity=32, async_budget_us=1000,
                policy_kwargs=dict(features_extractor_class=AugmentedNatureCNN, share_features_extractor=True, net_arch=[256, 256]))
    assert model._async.policy_parts_fn.accepts_record_rows()
    model._async.obs_stage.fill_(9)
    model.collect_rollouts(); st = model.train(); torch.cuda.synchronize()
    assert np.isfinite(float(st["loss"]))
    assert (model._async.obs_stage == 9).all()                  # the staging rows were never written
    rows = model._async.window_rows()
    assert int((model._async.obs[rows][:, :3] != 0).sum()) > 1000   # the records hold the rendered observations
    env.close()


def test_async_tick_under_bf16_autocast_keeps_its_graph(engine, torch):
    """PPO(autocast_dtype=bf16) (bench.py --policy-dtype bf16): the extractor's in-place fast path steps aside under autocast, so the tick must
    not hand the policy engine.RecordRows (whose fallback used to sync with the host: illegal during the stream capture, the tick graph was
    dropped with a warning). The gate now covers autocast, the fallback is a sync-free gather, and the captured tick survives."""
    import warnings
    from mujoco_rl_manipulate_unknown_objects_amd.simulation.environment.robot_env import BatchedRobotEnv, default_config
    from mujoco_rl_manipulate_unknown_objects_amd.sb3 import PPO, GpuVecEnv
    from mujoco_rl_manipulate_unknown_objects_amd.models.feature_extractor import AugmentedNatureCNN
    env = BatchedRobotEnv(default_config(sim_env="/xmls/sand_ball_env.xml"), n_envs=64, auto_reset=True)
    model = PPO("MultiInputPolicy", GpuVecEnv(env), n_steps=2, batch_size=64, n_epochs=1, async_slice=32, async_capacity=32, async_budget_us=1000,
                autocast_dtype=torch.bfloat16,
                policy_kwargs=dict(features_extractor_class=AugmentedNatureCNN, share_features_extractor=True, net_arch=[256, 256]))
    assert not model._async.policy_parts_fn.accepts_record_rows()
    with warnings.catch_warnings():
        warnings.simplefilter("error")                      # "hipGraph capture of the rollout tick failed" would be a warning
        for _ in range(2):
            model.collect_rollouts(); st = model.train()
    torch.cuda.synchronize()
    assert np.isfinite(float(st["loss"]))
    assert model._async._graph is not None                  # the tick is replayed from its graph
    # the fallback itself: rows without a host sync, equal to the slice
    rec = torch.randint(0, 256, (50, 5, 64, 64), dtype=torch.uint8, device="cuda"); row = torch.tensor([7], dtype=torch.int64, device="cuda")
    assert torch.equal(engine.RecordRows(rec, row, 9).materialize(), rec[7:16])
    env.close()


def test_every_slice_launch_is_timed_by_the_device_clock(engine, torch):
    """grip_batch_device_time (round 4): the first workgroup of a grip_batch_advance launch stamps its start, every wave its end, the compaction that
    follows adds the difference to a device counter -- so launches replayed from a hipGraph are timed too. Eager and replayed launches are both counted,
    the mean duration is positive and no longer than the wall-clock budget plus one physics.step()."""
    b = engine.Batch("sand_ball", 256, auto_reset=1)
    cap = 64
    lst = torch.full((cap,), -1, dtype=torch.int32, device="cuda"); cnt = torch.zeros(1, dtype=torch.int32, device="cuda")
    act = torch.rand(cap, 6, device="cuda") * 2 - 1
    b.device_time(reset=True)
    for _ in range(5):
        b.advance(act, 24, lst, cnt, 500)
    ms, n = b.device_time(reset=False)
    assert n == 5 and 0.0 < ms < 0.5 + 0.2
    s = torch.cuda.Stream()
    with torch.cuda.stream(s):
        b.advance(act, 24, lst, cnt, 500)                   # warm up on the capture stream
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g, stream=s):
            b.advance(act, 24, lst, cnt, 500)
        for _ in range(7):
            g.replay()
    torch.cuda.synchronize()
    ms2, n2 = b.device_time(reset=True)
    assert n2 == 5 + 1 + 7 and 0.0 < ms2 < 0.7                # the captured launch itself does not run; its seven replays do
    assert b.device_time(reset=False)[1] == 0
    b.close()


def test_permlane32_swap_hands_each_half_of_a_wave_to_the_other(torch):
    """halves_u / halves_f of csrc/grip_physics.h (v_permlane32_swap with the same register in both operands: the lower half twice, the upper half twice),
    on which every clone-lane split of the physics kernel rests: tools/hiptests/t_swap.hip, compiled here and run on the device."""
    import subprocess, tempfile
    src = os.path.join(ROOT, "tools", "hiptests", "t_swap.hip")
    with tempfile.TemporaryDirectory() as d:
        exe = os.path.join(d, "t_swap")
        subprocess.run(["/opt/rocm/bin/hipcc", "-O3", "--offload-arch=gfx950", "-std=c++17", "-w", "-o", exe, src], check=True, capture_output=True, timeout=300)
        r = subprocess.run([exe], capture_output=True, text=True, timeout=120)
    assert r.returncode == 0 and "t_swap: ok" in r.stdout, (r.stdout, r.stderr)


def test_rollout_tick_graph_follows_the_parameters_through_updates():
    """The captured rollout tick reads the extractor's biases straight from the parameters. The explicit update sequence keeps the parameters in one flat
    buffer (sb3/fused_update.py); laid out at the first update, the tick graph of the first rollout went on reading the freed old storage -- garbage biases,
    NaN losses from the second rollout on (tools/train_probe.py found it; a single collect + train, as the other tests do, did not). Now they are laid out at
    construction. Here: three rollouts with two updates between them, ticks replayed from the graph; the values the LAST rollout recorded must be the
    current policy's values of the recorded observations (eager forward, 1e-4), and the losses finite."""
    import numpy as np, torch
    from mujoco_rl_manipulate_unknown_objects_amd.simulation.environment.robot_env import BatchedRobotEnv, default_config
    from mujoco_rl_manipulate_unknown_objects_amd.sb3 import PPO, GpuVecEnv
    from mujoco_rl_manipulate_unknown_objects_amd.models.feature_extractor import AugmentedNatureCNN
    env = BatchedRobotEnv(default_config(sim_env="/xmls/sand_ball_env.xml", time_horizon=50), n_envs=256, auto_reset=True)
    model = PPO("MultiInputPolicy", GpuVecEnv(env), n_steps=4, batch_size=256, n_epochs=1, seed=2, async_slice=48, async_capacity=128, async_budget_us=1000,
                policy_kwargs=dict(features_extractor_class=AugmentedNatureCNN, share_features_extractor=True, net_arch=[256, 256]))
    assert model._fused is not None
    ar = model._async
    for it in range(3):
        model.collect_rollouts()
        if it < 2:
            st = model.train()
            assert np.isfinite(float(st["loss"])) and np.isfinite(float(st["value_loss"])), (it, st)
    torch.cuda.synchronize()
    assert ar._graph is not None and ar.total_ticks > 3 * ar.graph_after            # the ticks were replays
    assert model._fused is not None and model._fused.intact() and all(torch.isfinite(p).all() for p in model.policy.parameters())
    rows = torch.arange(ar.tick0, ar.tick0 + ar.tick * ar.C, device=ar.dev)     # this rollout's own decisions (the carry rows before them were decided before the last update)
    rows = rows[ar.is_rec[rows]][:512]
    assert rows.numel() >= 256
    with torch.no_grad():
        v = model.policy.predict_values({"observation": ar.obs[rows]})
    err = (v - ar.values[rows]).abs().max().item()
    assert err < 1e-4 * max(1.0, ar.values[rows].abs().max().item()), err
    env.close()
