#!/usr/bin/env python3
"""Contact-regime fixtures for the macro-step parity tests (tests/test_gpu_contact.py, tests/test_oracle_contact.py).

Every oracle-vs-HIP comparison of RobotEnv.step used to start at reset, 0.6 m from the object: the grasp codes of
Actuator.check_grasp (actuator.py:134-184), the `grasped == 3` break of the CLOSE loop (robot_env.py:150-167), the
pheromone levels below 3 (actuator.py:198-215), pushing rewards and the `> 1.0 m` FAIL (robot_env.py:172-173) were never
produced. This script SEARCHES for them with the CPU oracle: batches of oracle envs play forward-biased random actions
with frequent close / open commands; the state BEFORE every macro step is kept, and when the step produces a wanted
outcome that (state, action) pair is recorded under its category. The state is rounded to float32 (what the device holds)
and the oracle step is repeated from the rounded state: only pairs that still produce the category are kept, together
with the oracle's outputs from the rounded state -- and only WELL-CONDITIONED pairs: the oracle's own outputs must survive fp32-sized
noise injected at every physics.step() (well_conditioned below), otherwise the row sits on a branch point of a stiff contact problem and
cannot tell a wrong implementation from a rounding difference.

Output: tests/golden/contact_states.npz -- per object arrays qpos [k,14], qvel [k,13], ctrl [k,7], warm [k,13],
flags [k,3] (episode_step, status, gripper_open), action [k,6], dir [k,2], category [k] and the oracle outputs exp_*.
Pure oracle data: nothing here reads /root/reference.
"""
import ctypes as C
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from oracle import orc  # noqa: E402

OBJECTS = ("sand_ball", "sugar_cube", "acorn", "bread_crumb")
# category -> how many (state, action) pairs to keep per object and direction
WANT = {"close_code1": 4, "close_code2": 4, "close_code3_break": 6, "close_full_no_grasp": 2, "open_after_close": 3,
        "pad_grasp_nonzero": 6, "push_reward": 8, "hull_contact_move": 6, "pher2": 2, "pher1": 2, "pher0": 2, "fail_far": 2,
        "return_loop": 2}
EXP_INT = ("n_substeps", "done", "status", "episode_step", "gripper_open", "object_grasped", "reached_target", "reached_initial",
           "reached_fail", "pad_grasp", "pad_pheromone")
EXP_F = ("reward", "total_distance", "line_distance")
EXP_V = ("final_obj_pos", "gripper_pos", "init_obj_pos")


def state_of(env):
    d = env.d
    return (np.array(d.qpos, np.float32), np.array(d.qvel, np.float32), np.array(d.ctrl, np.float32), np.array(d.qacc_warmstart, np.float32),
            np.array([env.episode_step, env.status, env.gripper_open], np.int32))


def oracle_from(m, cfg, st):
    """A fresh oracle env put into the (float32) state st: what tests do on both sides."""
    qpos, qvel, ctrl, warm, flags = st
    e = orc.EnvOracle(m, **cfg); e.reset()
    d = e.e.d
    d.qpos[:] = [float(x) for x in qpos]; d.qvel[:] = [float(x) for x in qvel]; d.ctrl[:] = [float(x) for x in ctrl]
    d.qacc_warmstart[:] = [float(x) for x in warm]
    e.e.episode_step, e.e.status, e.e.gripper_open = int(flags[0]), int(flags[1]), int(flags[2])
    orc.lib().orc_fwd_position(m.ptr, C.byref(e.e.d))
    return e


def categories(o, pre_hull, pre_open):
    """Names of the wanted outcomes a finished macro step shows. o: OrcStepOut; pre_hull: gripper-object hull contacts before it."""
    c = []
    closing = pre_open == 1 and o.gripper_open == 0 or (o.object_grasped != 0)
    if o.object_grasped == 1:
        c.append("close_code1")
    if o.object_grasped == 2:
        c.append("close_code2")
    if o.object_grasped == 3 and o.gripper_open == 0:
        c.append("close_code3_break")
    if pre_open == 0 and o.gripper_open == 1 and o.reached_target:
        c.append("open_after_close")
    if o.pad_grasp != 0:
        c.append("pad_grasp_nonzero")
    if o.reward > 0.3 and pre_hull:
        c.append("push_reward")
    if pre_hull and o.object_grasped == 0 and o.reward == 0.0:
        c.append("hull_contact_move")
    if o.reached_initial or o.reached_fail:
        c.append("return_loop")
    del closing
    return c


def well_conditioned(m, cfg, st, act, o_ref, rng, trials=3, amp=5e-7, vamp=3e-5):
    """A contact macro step is hundreds of stiff physics.step() calls. Some states sit on a branch point -- typically a finger resting
    at exactly the 1 mm contact margin of the object while the CLOSE loop presses it against the other finger: whether check_grasp
    sees that contact is decided by the 4th decimal of a millimetre -- where the rounding noise ANY fp32 implementation injects at every
    physics.step() decides the integer outputs. Such rows cannot tell a wrong implementation from a rounding difference. Keep a row
    only if the oracle's own outputs survive that noise: orc_set_step_noise2(amp, vamp) perturbs qpos by amp * (1 + |x|) * U(-1, 1) and
    qvel by vamp * (1 + |v|) * U(-1, 1) after every physics.step(). The amplitudes are the measured one-step errors of the fp32 kernel
    started from the oracle's own states along contact trajectories (measured on MI355X, tests/test_gpu_contact.py::test_one_step_parity_along_contact_trajectories: qpos median 3e-8 / p99 4e-7,
    qvel median 1e-6 / p99 2e-4 -- the velocity error is the Newton solve stopping at the fp32 noise floor of its gradient). Three
    seeds; integer outputs identical, reward within 2e-3, gripper and object within 2e-4 m of the noise-free run."""
    del rng
    L = orc.lib()
    L.orc_set_step_noise.argtypes = [C.c_double, C.c_uint]; L.orc_set_step_noise2.argtypes = [C.c_double, C.c_double, C.c_uint]
    ok = True
    try:
        for seed in range(1, trials + 1):
            L.orc_set_step_noise2(amp, vamp, seed)
            o = oracle_from(m, cfg, st).step(act)
            if any(getattr(o, f) != getattr(o_ref, f) for f in EXP_INT):
                ok = False
            elif max(np.abs(np.array(o.final_obj_pos) - np.array(o_ref.final_obj_pos)).max(), np.abs(np.array(o.gripper_pos) - np.array(o_ref.gripper_pos)).max()) > 2e-4:
                ok = False
            elif abs(o.reward - o_ref.reward) > 2e-3:
                ok = False
            if not ok:
                break
    finally:
        L.orc_set_step_noise(0.0, 0)
    return ok


rejected = [0]


def search(obj, direction, rng, n=128, rounds=700):
    m = orc.Model(obj)
    cfg = dict(target_dir=direction)
    b = orc.BatchOracle(m, n, **cfg)
    found = {k: [] for k in WANT}
    for r in range(rounds):
        acts = rng.uniform(-1, 1, (n, 6)).astype(np.float32)
        acts[:, 0] = np.abs(acts[:, 0]) * rng.choice([1.0, 0.4], n)             # towards the object
        acts[:, 1] *= 0.6; acts[:, 2] *= 0.5
        mode = rng.integers(0, 4, n)
        acts[mode == 0, 5] = -1.0                                              # close
        acts[mode == 1, 5] = 1.0                                               # open
        small = rng.random(n) < 0.3
        acts[small, :5] *= 0.15                                                # close / open nearly in place
        pre = [state_of(b.envs[i]) for i in range(n)]
        pre_hull = [any(b.envs[i].d.con[c].g1 != 0 and b.envs[i].d.con[c].g2 == 6 for c in range(b.envs[i].d.ncon)) for i in range(n)]
        b.step(acts, auto_reset=True)
        for i in range(n):
            o = b.outs[i]
            cats = [c for c in categories(o, pre_hull[i], int(pre[i][4][2])) if len(found[c]) < WANT[c]]
            if not cats:
                continue
            # repeat from the float32-rounded state; keep only if the outcome survives
            e = oracle_from(m, cfg, pre[i])
            hull32 = any(e.d.con[c].g1 != 0 and e.d.con[c].g2 == 6 for c in range(e.d.ncon))
            o2 = e.step(acts[i])
            cats2 = categories(o2, hull32, int(pre[i][4][2]))
            keep = [c for c in cats if c in cats2]
            if keep and not well_conditioned(m, cfg, pre[i], acts[i], o2, rng):
                rejected[0] += 1
                continue
            for c in keep:
                if len(found[c]) < WANT[c]:
                    found[c].append((pre[i], acts[i].copy(), c))
                    break
        if all(len(found[k]) >= WANT[k] for k in WANT if not k.startswith(("pher", "fail", "close_full"))):
            break
    # constructed states: the gripper moved off the target line (pheromone levels 2 / 1 / 0) and far from the object (FAIL)
    base = orc.EnvOracle(m, **cfg); base.reset()
    s0 = state_of(base.e)
    dn = np.array(direction, np.float64) / np.linalg.norm(direction)
    perp = np.array([-dn[1], dn[0]])
    ee0 = np.array([base.d.xpos[1][0], base.d.xpos[1][1]])

    obj0 = np.array([base.d.xpos[7][0], base.d.xpos[7][1]])
    along_obj = float((obj0 - ee0) @ dn)                       # puts the gripper level with the object along the target line
    side = -1.0 if obj0 @ perp < 0 else 1.0                    # the object's side of the line (keeps |ee - obj| < 1 m at |offset| < 1 m)

    def placed(offset, along=0.0):
        q = s0[0].copy()
        tgt = (ee0 @ dn + along) * dn + offset * perp
        q[0:2] = (tgt - ee0).astype(np.float32)
        return (q, s0[1].copy(), s0[2].copy(), s0[3].copy(), s0[4].copy())
    # thresholds of actuator.py:207-215: e^-d > 0.82 / 0.6 / 0.37, i.e. d < 0.198 / 0.511 / 0.994
    for name, off in (("pher2", (0.25, -0.4)), ("pher1", (0.6, -0.8))):
        for k in range(WANT[name]):
            a = rng.uniform(-0.3, 0.3, 6).astype(np.float32)
            found[name].append((placed(off[k % 2]), a, name))
    for k in range(WANT["pher0"]):                              # k even: level 0 while still within 1 m of the object; k odd: beyond it (FAIL too)
        a = rng.uniform(-0.3, 0.3, 6).astype(np.float32); a[:3] = 0.0
        found["pher0"].append((placed(side * 0.998, along_obj) if k % 2 == 0 else placed(-side * 1.05, along_obj), a, "pher0"))
    for k in range(WANT["fail_far"]):
        a = rng.uniform(-0.3, 0.3, 6).astype(np.float32)
        found["fail_far"].append((placed(0.0, along=-0.8 - 0.2 * k), a, "fail_far"))
    # closing in free space: the CLOSE loop runs to its tolerance exit (grasp code 0, gripper_open -> 0)
    for k in range(WANT["close_full_no_grasp"]):
        a = rng.uniform(-0.2, 0.2, 6).astype(np.float32); a[5] = -1.0
        found["close_full_no_grasp"].append((s0, a, "close_full_no_grasp"))
    return m, cfg, found


def main():
    rng = np.random.default_rng(20261004)
    out = {}
    for obj in OBJECTS:
        rows = []
        for direction in ((1.0, 0.0), (1.0, 1.0)):
            m, cfg, found = search(obj, direction, rng)
            print(obj, direction, {k: len(v) for k, v in found.items()}, "ill-conditioned rows rejected so far:", rejected[0], flush=True)
            for name, lst in found.items():
                for st, act, cat in lst:
                    e = oracle_from(m, cfg, st)
                    o = e.step(act)
                    rows.append((st, act, cat, direction, o.__class__.from_buffer_copy(o)))
        k = len(rows)
        out[f"{obj}/qpos"] = np.array([r[0][0] for r in rows], np.float32); out[f"{obj}/qvel"] = np.array([r[0][1] for r in rows], np.float32)
        out[f"{obj}/ctrl"] = np.array([r[0][2] for r in rows], np.float32); out[f"{obj}/warm"] = np.array([r[0][3] for r in rows], np.float32)
        out[f"{obj}/flags"] = np.array([r[0][4] for r in rows], np.int32); out[f"{obj}/action"] = np.array([r[1] for r in rows], np.float32)
        out[f"{obj}/category"] = np.array([r[2] for r in rows]); out[f"{obj}/dir"] = np.array([r[3] for r in rows], np.float32)
        for f in EXP_INT:
            out[f"{obj}/exp_{f}"] = np.array([getattr(r[4], f) for r in rows], np.int32)
        for f in EXP_F:
            out[f"{obj}/exp_{f}"] = np.array([getattr(r[4], f) for r in rows], np.float64)
        for f in EXP_V:
            out[f"{obj}/exp_{f}"] = np.array([list(getattr(r[4], f)) for r in rows], np.float64)
        print(obj, k, "pairs")
    path = os.path.join(ROOT, "tests", "golden", "contact_states.npz")
    np.savez_compressed(path, **out)
    print("wrote", path, os.path.getsize(path), "bytes")


if __name__ == "__main__":
    main()
