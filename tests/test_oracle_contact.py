"""The contact-regime fixture (tests/golden/contact_states.npz, made by tools/make_contact_states.py with the CPU oracle) is what
the GPU parity tests of RobotEnv.step in contact start from. Here (CPU): the fixture holds every outcome the reference's macro
step can produce around the object, the oracle reproduces its own recorded outputs from the float32 states, and the two golden
sections of the reference's controller nobody read before (`enforce_constraints`, `euler`) pin the oracle's helpers."""
import ctypes as C
import os

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
OBJECTS = ["sand_ball", "sugar_cube", "acorn", "bread_crumb"]


@pytest.fixture(scope="module")
def contact():
    return np.load(os.path.join(ROOT, "tests", "golden", "contact_states.npz"))


def oracle_from_row(orc, m, z, obj, i):
    """Oracle env in fixture state i (float32 values widened to double, then the position stage: contacts of that state)."""
    e = orc.EnvOracle(m, target_dir=tuple(float(x) for x in z[f"{obj}/dir"][i])); e.reset()
    d = e.e.d
    d.qpos[:] = [float(x) for x in z[f"{obj}/qpos"][i]]; d.qvel[:] = [float(x) for x in z[f"{obj}/qvel"][i]]
    d.ctrl[:] = [float(x) for x in z[f"{obj}/ctrl"][i]]; d.qacc_warmstart[:] = [float(x) for x in z[f"{obj}/warm"][i]]
    fl = z[f"{obj}/flags"][i]
    e.e.episode_step, e.e.status, e.e.gripper_open = int(fl[0]), int(fl[1]), int(fl[2])
    orc.lib().orc_fwd_position(m.ptr, C.byref(e.e.d))
    return e


@pytest.mark.parametrize("obj", OBJECTS)
def test_fixture_covers_every_contact_outcome(contact, obj):
    """Grasp codes 1 / 2 / 3 during CLOSE (actuator.py:134-184), the CLOSE loop ended by `grasped == 3` (robot_env.py:164-166) and by
    its tolerance, OPEN after CLOSE, a non-zero sensor pad, pheromone levels 0..3 (actuator.py:198-215), pushes with reward, RETURN
    loops and the > 1 m FAIL (robot_env.py:172-173) all occur, for both target directions."""
    z = contact
    cat = z[f"{obj}/category"]; g = z[f"{obj}/exp_object_grasped"]; pad = z[f"{obj}/exp_pad_grasp"]; ph = z[f"{obj}/exp_pad_pheromone"]
    assert {1, 2, 3} <= set(g.tolist())                      # the grasp code does not depend on the target direction: over both
    for d in ((1.0, 0.0), (1.0, 1.0)):
        sel = (z[f"{obj}/dir"] == np.array(d, np.float32)).all(1)
        assert {0, 1, 2, 3} <= set(ph[sel].tolist())
        assert {1, 2} <= set(pad[sel].tolist())
        br = sel & (cat == "close_code3_break")
        assert br.sum() >= 4 and (z[f"{obj}/exp_gripper_open"][br] == 0).all() and (g[br] == 3).all()
        assert (z[f"{obj}/exp_n_substeps"][br] < 400).all()                 # MOVE + a CLOSE loop cut short, not its 400-step limit
        assert (z[f"{obj}/exp_reward"][sel & (cat == "push_reward")] > 0.3).all() and (sel & (cat == "push_reward")).sum() >= 6
        ff = sel & (cat == "fail_far")
        assert ff.sum() >= 2 and (z[f"{obj}/exp_status"][ff] == 1).all() and (z[f"{obj}/exp_done"][ff] == 1).all()
        p0 = sel & (cat == "pher0")
        assert (ph[p0] == 0).all() and set(z[f"{obj}/exp_status"][p0].tolist()) == {0, 1}     # level 0 with and without the FAIL
        assert (z[f"{obj}/exp_reached_initial"][sel] | z[f"{obj}/exp_reached_fail"][sel]).any()


@pytest.mark.parametrize("obj", OBJECTS)
def test_oracle_reproduces_the_fixture(orc, contact, obj):
    """Determinism of the checker: from the recorded float32 state the oracle gives the recorded outputs, bit for bit."""
    z = contact; m = orc.Model(obj)
    n = len(z[f"{obj}/category"])
    for i in range(n):
        o = oracle_from_row(orc, m, z, obj, i).step(z[f"{obj}/action"][i])
        for f in ("n_substeps", "done", "status", "episode_step", "gripper_open", "object_grasped", "reached_target", "reached_initial",
                  "reached_fail", "pad_grasp", "pad_pheromone"):
            assert getattr(o, f) == z[f"{obj}/exp_{f}"][i], (obj, i, f)
        assert o.reward == z[f"{obj}/exp_reward"][i]
        assert list(o.final_obj_pos) == z[f"{obj}/exp_final_obj_pos"][i].tolist() and list(o.gripper_pos) == z[f"{obj}/exp_gripper_pos"][i].tolist()


def test_enforce_constraints_golden(orc, golden):
    """Actuator._enforce_constraints (actuator.py:266-293) on the reference's own vectors: roll clamp +-pi/4, pitch 0, z in [0.1, 0.5]."""
    L = orc.lib()
    L.orc_enforce_constraints.argtypes = [C.c_int, C.POINTER(C.c_double), C.POINTER(C.c_double)]
    cases = golden["enforce_constraints"]
    assert len(cases) >= 8
    for c in cases:
        p = np.array(c["position"], np.float64); o = np.array(c["orientation"], np.float64)
        L.orc_enforce_constraints(1, orc._dp(p), orc._dp(o))
        assert np.array_equal(p, np.array(c["out_position"])) and np.array_equal(o, np.array(c["out_orientation"]))
    p = np.array([0.0, 0.0, 0.7]); o = np.array([0.3, 0.2, 0.1])
    L.orc_enforce_constraints(0, orc._dp(p), orc._dp(o))                     # --include_roll False: roll forced to 0 (actuator.py:270-272)
    assert p[2] == 0.5 and o.tolist() == [0.0, 0.0, 0.1]


def test_euler_golden(orc, golden):
    """transformations.py rows of a16 on the reference's own vectors: euler_matrix / euler_from_matrix ('sxyz') and
    euler_from_quaternion(axes=(0, 0, 0, 1)) as Actuator._get_current_pose calls it (actuator.py:50-56)."""
    L = orc.lib()
    dp = C.POINTER(C.c_double)
    L.orc_euler_matrix_sxyz.argtypes = [C.c_double, C.c_double, C.c_double, dp]
    L.orc_euler_sxyz_from_matrix.argtypes = [dp, dp]
    L.orc_euler_rzyx_from_quat_wxyz.argtypes = [dp, dp]
    cases = golden["euler"]
    assert len(cases) >= 10
    for c in cases:
        M = np.zeros(9); e = np.zeros(3); r = np.zeros(3)
        L.orc_euler_matrix_sxyz(*c["angles"], orc._dp(M))
        assert np.abs(M.reshape(3, 3) - np.array(c["matrix"])).max() < 1e-15
        Mg = np.ascontiguousarray(np.array(c["matrix"]).reshape(9))
        L.orc_euler_sxyz_from_matrix(orc._dp(Mg), orc._dp(e))
        assert np.abs(e - np.array(c["back"])).max() < 1e-15
        q = np.array(c["quat_wxyz"], np.float64)
        L.orc_euler_rzyx_from_quat_wxyz(orc._dp(q), orc._dp(r))
        assert np.abs(r - np.array(c["euler_rzyx"])).max() < 1e-13


def test_oracle_under_address_and_ub_sanitizers():
    """the CPU restatement runs clean under -fsanitize=address,undefined (`make -C oracle asan`). A child
    process (the sanitizer runtime has to be loaded first) plays contact-rich macro steps from the fixture and renders."""
    import subprocess, sys
    r = subprocess.run(["make", "-C", os.path.join(ROOT, "oracle"), "asan"], capture_output=True, text=True)
    assert r.returncode == 0, r.stderr[-2000:]
    asan_rt = subprocess.run(["gcc", "-print-file-name=libasan.so"], capture_output=True, text=True).stdout.strip()
    ubsan_rt = subprocess.run(["gcc", "-print-file-name=libubsan.so"], capture_output=True, text=True).stdout.strip()
    if not os.path.isabs(asan_rt) or not os.path.exists(asan_rt):
        pytest.skip("gcc has no libasan runtime here")
    env = dict(os.environ, LD_PRELOAD=asan_rt + (":" + ubsan_rt if os.path.isabs(ubsan_rt) else ""), ASAN_OPTIONS="detect_leaks=0:halt_on_error=1",
               UBSAN_OPTIONS="halt_on_error=1:print_stacktrace=1", GRIP_ORACLE_LIB=os.path.join(ROOT, "oracle", "_build", "libgrip_oracle_asan.so"),
               OMP_NUM_THREADS="2")
    code = r"""
import sys, numpy as np
sys.path.insert(0, %r)
from oracle import orc
sys.path.insert(0, %r)
from test_oracle_contact import oracle_from_row
z = np.load(%r)
for obj in ("sand_ball", "bread_crumb"):
    m = orc.Model(obj)
    for i in range(0, len(z[obj + "/category"]), 7):
        e = oracle_from_row(orc, m, z, obj, i); e.step(z[obj + "/action"][i]); e.observation()
    b = orc.BatchOracle(m, 4); b.step(np.zeros((4, 6)))
print("SANITIZED-OK")
""" % (ROOT, os.path.join(ROOT, "tests"), os.path.join(ROOT, "tests", "golden", "contact_states.npz"))
    r = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, env=env, timeout=600)
    assert r.returncode == 0 and "SANITIZED-OK" in r.stdout, (r.stdout[-500:], r.stderr[-3000:])
    assert "runtime error" not in r.stderr and "AddressSanitizer" not in r.stderr, r.stderr[-3000:]
