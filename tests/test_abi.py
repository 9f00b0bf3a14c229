"""The C-ABI library loads without a GPU and exports every symbol include/grip_sim.h declares;
the product path fails loudly (no CPU fallback) when there is no device."""
import ctypes as C
import os
import re

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def lib():
    from mujoco_rl_manipulate_unknown_objects_amd import engine
    engine.build_library()
    return engine.lib()


def test_every_declared_symbol_is_exported(lib):
    from mujoco_rl_manipulate_unknown_objects_amd import engine
    hdr = open(os.path.join(ROOT, "include", "grip_sim.h")).read()
    declared = sorted(set(re.findall(r"\b(grip_[a-z0-9_]+)\s*\(", hdr)))
    assert len(declared) >= 19
    raw = C.CDLL(engine.LIB_PATH)
    for name in declared:
        assert hasattr(raw, name), name
    assert sorted(engine.EXPORTS) == declared


def test_model_load_is_host_only_and_validates(lib):
    from mujoco_rl_manipulate_unknown_objects_amd import engine
    m = engine.Model("bread_crumb")
    assert lib.grip_model_nvert(m.ptr) == 408 + 70 + 120 + 70 + 120 + 573
    with pytest.raises(engine.GripError):
        engine.Model(os.path.join(ROOT, "README.md"))


def test_model_load_rejects_truncated_blobs(lib, tmp_path):
    """A cut-off or corrupted .grpm is refused with a message, never read past its end."""
    from mujoco_rl_manipulate_unknown_objects_amd import engine
    blob = open(os.path.join(ROOT, "mujoco_rl_manipulate_unknown_objects_amd", "assets", "sand_ball_env.grpm"), "rb").read()
    for cut in (0, 3, 11, 12, 40, 61, len(blob) // 3, len(blob) // 2, len(blob) - 9):
        f = tmp_path / f"cut{cut}.grpm"
        f.write_bytes(blob[:cut])
        with pytest.raises(engine.GripError):
            engine.Model(str(f))
    bad = bytearray(blob); bad[12 + 28:12 + 32] = (9).to_bytes(4, "little")       # first entry claims 9 dimensions
    f = tmp_path / "ndim.grpm"; f.write_bytes(bytes(bad))
    with pytest.raises(engine.GripError):
        engine.Model(str(f))
    bad = bytearray(blob); bad[12 + 32:12 + 36] = (0x7FFFFFFF).to_bytes(4, "little")   # ... or a size beyond the file
    f = tmp_path / "dims.grpm"; f.write_bytes(bytes(bad))
    with pytest.raises(engine.GripError):
        engine.Model(str(f))


def test_entry_points_validate_arguments_without_a_gpu(lib):
    """Bad arguments are refused with a message before anything touches a device."""
    vp = C.c_void_p
    out = vp()
    assert lib.grip_batchset_create(None, 0, None, C.byref(out)) != 0 and b"grip_batchset_create" in lib.grip_last_error()
    arr = (vp * 1)(None)
    assert lib.grip_batchset_create(arr, 1, None, C.byref(out)) != 0 and b"null batch" in lib.grip_last_error()
    assert lib.grip_batchset_step(None, None, None) != 0 and b"grip_batchset_step" in lib.grip_last_error()
    assert lib.grip_batchset_advance(None, None, 1, 0, 1, 8, None, None, None) != 0
    assert lib.grip_batchset_observe(None, None, None) != 0 and lib.grip_batchset_observe_list(None, None, 8, None, None, None, None) != 0
    assert lib.grip_batchset_num_envs(None) == -1 and lib.grip_batch_num_envs(None) == -1
    assert lib.grip_batch_set_state_storage(None, 1, None) != 0 and b"grip_batch_set_state_storage" in lib.grip_last_error()
    assert lib.grip_batch_step(None, None, None, None) != 0 and lib.grip_batch_advance(None, None, 1, 0, 1, 8, None, None, None, None) != 0
    assert lib.grip_rollout_tick(None, None) != 0 and b"grip_rollout_tick" in lib.grip_last_error()
    assert lib.grip_obs_preprocess(None, 0, 5, None, None, None) != 0 and b"grip_obs_preprocess" in lib.grip_last_error()
    lib.grip_batchset_destroy(None); lib.grip_batch_destroy(None)          # no-ops


def test_no_cpu_fallback():
    import torch
    from mujoco_rl_manipulate_unknown_objects_amd import engine
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    with pytest.raises(engine.GripError):
        engine.Batch("sand_ball", 4)
    ptr = C.c_void_p()
    m = engine.Model("sand_ball")
    assert engine.lib().grip_batch_create(m.ptr, 4, 0, C.byref(ptr)) != 0
    assert b"hip" in engine.lib().grip_last_error().lower() or b"device" in engine.lib().grip_last_error().lower()


def test_config_mirrors_reference_flags():
    from mujoco_rl_manipulate_unknown_objects_amd.config.train_config import TrainConfig
    from mujoco_rl_manipulate_unknown_objects_amd.config.eval_config import EvalConfig
    c = TrainConfig().parse([])
    assert (c.sim_env, c.max_steps, c.time_horizon, c.pos_tolerance, c.grasp_tolerance) == ("/xmls/acorn_env.xml", 400, 400, 0.002, 0.03)
    assert (c.total_timesteps, c.batch_size, c.eval_freq, c.eval_episodes) == (500000, 256, 2000, 3)
    assert TrainConfig().parse(["--full_observation", "False"]).full_observation is True     # type=bool quirk
    assert TrainConfig().parse(["--name", "abc", "--suffix", "{direction}_x"]).name == "abc_0_x"
    assert EvalConfig().parse([]).train_env == "bread_crumb"
