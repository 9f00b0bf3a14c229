"""Opt-in cross-check of the oracle's physics against a real MuJoCo.

Physics parity is UNPINNED here: MuJoCo is not installed in this image or on the GPU box and the reference holds no vector at the
physics.step() boundary, so the C oracle restates MuJoCo's published model and is checked by known-answer tests only. Wherever a
`mujoco` wheel and the reference's MJCF + meshes ARE available (a maintainer's machine: set GRIP_REFERENCE_XMLS to the directory
holding sand_ball_env.xml etc.), this test pins it: it loads the same model into MuJoCo, copies oracle states in, and compares one
mj_forward / ten mj_step calls. Skipped otherwise -- it never fails for lack of MuJoCo."""
import importlib.util
import os

import numpy as np
import pytest

HAVE_MUJOCO = importlib.util.find_spec("mujoco") is not None
XMLS = os.environ.get("GRIP_REFERENCE_XMLS", "/root/reference/xmls")


@pytest.mark.skipif(not HAVE_MUJOCO, reason="mujoco is not installed (physics parity stays unpinned: DESIGN.md section 2)")
@pytest.mark.parametrize("obj", ["sand_ball", "sugar_cube", "bread_crumb"])      # acorn.stl is missing from the reference checkout
def test_oracle_step_matches_mujoco(orc, obj):
    import mujoco
    xml = os.path.join(XMLS, f"{obj}_env.xml")
    if not os.path.exists(xml):
        pytest.skip(f"{xml} not found (set GRIP_REFERENCE_XMLS)")
    mj = mujoco.MjModel.from_xml_path(xml); d = mujoco.MjData(mj)
    m = orc.Model(obj)
    rng = np.random.default_rng(0)
    e = orc.EnvOracle(m); e.reset()
    ee = mujoco.mj_name2id(mj, mujoco.mjtObj.mjOBJ_BODY, "ee")
    worst_q = worst_a = 0.0
    for k in range(40):
        a = rng.uniform(-1, 1, 6).astype(np.float32); a[0] = abs(a[0])
        if e.step(a).done:
            e.reset()
        s = orc.Sim(m)
        s.qpos[:] = np.array(e.d.qpos); s.qvel[:] = np.array(e.d.qvel); s.ctrl[:] = rng.uniform(-1, 1, 7); s.qacc_warmstart[:] = np.array(e.d.qacc_warmstart)
        s.d.xfrc[1][2] = 0.438 * 9.81
        mujoco.mj_resetData(mj, d)
        d.qpos[:] = s.qpos; d.qvel[:] = s.qvel; d.ctrl[:] = s.ctrl; d.qacc_warmstart[:] = s.qacc_warmstart
        d.xfrc_applied[ee, 2] = 0.438 * 9.81
        s.fwd_position(); s.forward(); mujoco.mj_forward(mj, d)
        assert s.d.ncon == d.ncon, (k, s.d.ncon, d.ncon)
        assert np.abs(s.M - mujoco_full_m(mujoco, mj, d)).max() < 1e-9
        assert np.abs(s.qfrc_bias - d.qfrc_bias).max() < 1e-8
        worst_a = max(worst_a, float(np.abs(s.qacc - d.qacc).max() / (1 + np.abs(d.qacc).max())))
        # dm_control's Physics.step(): mj_step2 then mj_step1 on a forwarded state
        for _ in range(10):
            s.step(1); mujoco.mj_step2(mj, d); mujoco.mj_step1(mj, d)
        worst_q = max(worst_q, float(np.abs(s.qpos - d.qpos).max()))
    print(f"\n[mujoco cross-check] {obj}: worst relative qacc gap {worst_a:.2e}, worst qpos gap after 10 steps {worst_q:.2e}")
    assert worst_a < 1e-4 and worst_q < 1e-5


def mujoco_full_m(mujoco, mj, d):
    M = np.zeros((mj.nv, mj.nv)); mujoco.mj_fullM(mj, M, d.qM)
    return M


def test_crosscheck_is_wired():
    """Runs everywhere: the opt-in test exists, is skipped only for the stated reason, and the reason is recorded."""
    assert HAVE_MUJOCO or True
    import inspect, sys
    src = inspect.getsource(sys.modules[__name__])
    assert "find_spec(\"mujoco\")" in src and "parity stays unpinned" in src
