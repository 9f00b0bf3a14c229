"""Generate tests/golden/controller_golden.json from the reference's OWN Python.

Runs only in the build container (needs /root/reference). The reference's hot path cannot be
imported as a package (gym, dm_control, cv2, stable_baselines3 are not installed -- ordinary
ModuleNotFoundError), so individual files are loaded *by path* after inert stand-in
modules are placed in sys.modules for those packages. What executes is the reference's own
arithmetic (actuator.py, reward.py, utils.py, transformations.py, feature_extractor.py) against
fake `physics` objects; only inputs and outputs are written out. No reference source or bytecode
is copied (PYTHONDONTWRITEBYTECODE is forced).
"""
import importlib.util
import json
import math
import os
import sys
import types

sys.dont_write_bytecode = True
import numpy as np

REF = os.environ.get("GRIP_REFERENCE", "/root/reference")
OUT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tests", "golden", "controller_golden.json")


def _stub_modules():
    gym = types.ModuleType("gym"); spaces = types.ModuleType("gym.spaces")

    class Box:
        def __init__(self, low, high, shape=None, dtype=np.float32):
            self.low, self.high, self.shape, self.dtype = low, high, shape, dtype

        def sample(self):
            return np.zeros(self.shape, dtype=self.dtype)

    class Dict(dict):
        def __init__(self, d):
            super().__init__(d)
            self.spaces = d
    spaces.Box, spaces.Dict = Box, Dict
    gym.spaces = spaces
    gym.GoalEnv = object
    sys.modules["gym"] = gym; sys.modules["gym.spaces"] = spaces
    cv2 = types.ModuleType("cv2"); sys.modules["cv2"] = cv2
    for name in ["dm_control", "dm_control.mujoco", "dm_control.mujoco.wrapper", "dm_control.mujoco.wrapper.mjbindings"]:
        sys.modules[name] = types.ModuleType(name)
    mjlib = types.SimpleNamespace()

    def mj_jacBody(model_ptr, data_ptr, jacp, jacr, body_id):
        # analytic Jacobian of body `ee` at its origin
        roll = data_ptr.physics.data.qpos[3]
        jacp[:] = 0; jacr[:] = 0
        jacp[0, 0] = jacp[1, 1] = jacp[2, 2] = 1.0
        jacr[:, 3] = (1.0, 0.0, 0.0)
        jacr[:, 4] = (0.0, -math.sin(roll), math.cos(roll))
    mjlib.mj_jacBody = mj_jacBody
    sys.modules["dm_control.mujoco.wrapper.mjbindings"].mjlib = mjlib
    import torch
    sb3 = types.ModuleType("stable_baselines3"); common = types.ModuleType("stable_baselines3.common")
    tl = types.ModuleType("stable_baselines3.common.torch_layers")

    class BaseFeaturesExtractor(torch.nn.Module):
        def __init__(self, observation_space, features_dim=0):
            super().__init__()
            self._observation_space = observation_space
            self._features_dim = features_dim

        @property
        def features_dim(self):
            return self._features_dim
    tl.BaseFeaturesExtractor = BaseFeaturesExtractor
    sys.modules["stable_baselines3"] = sb3; sys.modules["stable_baselines3.common"] = common
    sys.modules["stable_baselines3.common.torch_layers"] = tl
    return spaces


def _load(modname, relpath):
    spec = importlib.util.spec_from_file_location(modname, os.path.join(REF, relpath))
    mod = importlib.util.module_from_spec(spec)
    sys.modules[modname] = mod
    spec.loader.exec_module(mod)
    return mod


class FakePhysics:
    """Just enough of dm_control's Physics for actuator.py: named xpos/xquat, qpos, contacts."""

    def __init__(self):
        self.data = types.SimpleNamespace(qpos=np.zeros(14), ctrl=np.zeros(7), ncon=0, contact=[])
        self.data.ptr = types.SimpleNamespace(physics=self)
        self.model = types.SimpleNamespace(nv=13, ptr=None, name2id=lambda n, t: 3, geom_bodyid=np.array([0, 4, 5, 6, 7, 8, 9]))
        names = {"left_inner_knuckle": 5, "left_inner_finger": 6, "right_inner_knuckle": 7, "right_inner_finger": 8, "object": 9}
        self.named = types.SimpleNamespace(
            data=types.SimpleNamespace(xpos={"ee": np.zeros(3), "object": np.zeros(3)}, xquat={"ee": np.array([1., 0, 0, 0])}),
            model=types.SimpleNamespace(geom_bodyid=names))

    def set_ee(self, slide, roll, yaw):
        self.data.qpos[:3] = slide; self.data.qpos[3] = roll; self.data.qpos[4] = yaw
        self.named.data.xpos["ee"] = np.array([-0.5, 0.0, 0.15]) + np.asarray(slide)
        # quaternion of Rx(roll) * Rz(yaw), wxyz
        qr = np.array([math.cos(roll / 2), math.sin(roll / 2), 0, 0]); qy = np.array([math.cos(yaw / 2), 0, 0, math.sin(yaw / 2)])
        w1, x1, y1, z1 = qr; w2, x2, y2, z2 = qy
        self.named.data.xquat["ee"] = np.array([w1*w2 - x1*x2 - y1*y2 - z1*z2, w1*x2 + x1*w2 + y1*z2 - z1*y2,
                                                 w1*y2 - x1*z2 + y1*w2 + z1*x2, w1*z2 + x1*y2 - y1*x2 + z1*w2])


def main():
    _stub_modules()
    sim = types.ModuleType("simulation"); sim.__path__ = []
    sys.modules["simulation"] = sim
    for pkg in ["simulation.utils", "simulation.controller", "simulation.environment"]:
        p = types.ModuleType(pkg); p.__path__ = []; sys.modules[pkg] = p
    tr = _load("simulation.utils.transformations", "simulation/utils/transformations.py")
    sys.modules["simulation.utils"].transformations = tr
    # utils.py needs cv2 only for make_pdf
    ut = _load("simulation.utils.utils", "simulation/utils/utils.py")
    _load("simulation.controller.sensor", "simulation/controller/sensor.py")
    act_mod = _load("simulation.controller.actuator", "simulation/controller/actuator.py")
    sys.modules["scipy.special"]  # ensure scipy present for reward.py import
    rew_mod = _load("simulation.environment.reward", "simulation/environment/reward.py")
    fe_mod = _load("models.feature_extractor", "models/feature_extractor.py")

    cfg = types.SimpleNamespace(max_rotation=0.15, max_translation=0.05, include_roll=True, width_capture=64,
                                height_capture=64, full_observation=True)
    rng = np.random.default_rng(20221003)
    G = {"meta": {"generator": "tools/make_golden.py", "reference": "kv13arm/mujoco_rl_manipulate_unknown_objects",
                  "note": "inputs/outputs of the reference's own functions; see tool docstring"}}

    phys = FakePhysics()
    A = act_mod.Actuator(robot=phys, config=cfg)
    A.setup_action_space()

    # ---- _normalise_action (+ _clip_translation_vector), float32 and float64 inputs
    cases = []
    acts = [np.array([1, 1, 1, .5, -.5, -1.]), np.array([0.2, -0.1, 0.05, 1.0, -1.0, 0.3])] + [rng.uniform(-1, 1, 6) for _ in range(10)]
    for a in acts:
        for dt in (np.float32, np.float64):
            t, r, oc = A._normalise_action(a.astype(dt))
            cases.append(dict(action=a.astype(dt).astype(float).tolist(), dtype=np.dtype(dt).name,
                              translation=np.asarray(t, dtype=float).tolist(), rotation=np.asarray(r, dtype=float).tolist(),
                              open_close=float(oc), out_dtype=str(np.asarray(t).dtype)))
    G["normalise_action"] = cases

    # ---- scale_control
    cases = []
    for _ in range(8):
        dq = rng.uniform(-0.1, 0.1, 5)
        cases.append(dict(dq=dq.tolist(), open_close=-1.0, ctrl=A.scale_control(dq, -1.0).tolist()))
    G["scale_control"] = cases

    # ---- _enforce_constraints
    cases = []
    for _ in range(8):
        pos = rng.uniform(-0.2, 0.8, 3); ori = rng.uniform(-1.5, 1.5, 3)
        p, o = A._enforce_constraints(pos.copy(), ori.copy())
        cases.append(dict(position=pos.tolist(), orientation=ori.tolist(), out_position=p.tolist(), out_orientation=o.tolist()))
    G["enforce_constraints"] = cases

    # ---- get_target_pose against the fake physics (float32 actions, as SB3 feeds them)
    cases = []
    for i in range(24):
        slide = rng.uniform(-0.3, 0.3, 3); slide[2] = rng.uniform(-0.04, 0.3)
        roll = rng.uniform(-0.7, 0.7) if i % 3 else 0.0
        yaw = rng.uniform(-3.0, 3.0) if i % 4 else 0.0
        phys.set_ee(slide, roll, yaw)
        a = rng.uniform(-1, 1, 6).astype(np.float32)
        tq = A.get_target_pose(a.copy())
        cases.append(dict(slide=slide.tolist(), roll=roll, yaw=yaw, action=a.astype(float).tolist(),
                          xpos=phys.named.data.xpos["ee"].tolist(), xquat=phys.named.data.xquat["ee"].tolist(),
                          target_qpos=np.asarray(tq, dtype=float).tolist()))
    G["get_target_pose"] = cases

    # ---- check_grasp with fake contact lists (geom ids: 0 floor,1 base,2 lk,3 lf,4 rk,5 rf,6 object)
    cases = []
    lists = [[], [(2, 6)], [(6, 3)], [(4, 6)], [(6, 5), (3, 6)], [(0, 6), (1, 6)], [(2, 4), (3, 5)], [(2, 6), (4, 6), (0, 6)], [(1, 3)]]
    for cl in lists:
        phys.data.contact = [types.SimpleNamespace(geom1=a, geom2=b) for a, b in cl]
        phys.data.ncon = len(cl)
        cases.append(dict(contacts=cl, code=int(A.check_grasp("object"))))
    G["check_grasp"] = cases

    # ---- pheromone_level
    cases = []
    for d in ([1, 0], [1, 1]):
        for _ in range(10):
            ee = rng.uniform(-1.2, 1.2, 3)
            phys.named.data.xpos["ee"] = ee
            cases.append(dict(ee=ee.tolist(), dir=d, level=int(A.pheromone_level(np.array(d)))))
    G["pheromone_level"] = cases

    # ---- project_to_target_direction
    G["project"] = [dict(pos=p.tolist(), dir=d, value=float(ut.project_to_target_direction(p, np.array(d))))
                    for d in ([1, 0], [1, 1]) for p in rng.uniform(-1, 1, (5, 2))]

    # ---- Reward.agent_reward
    R = rew_mod.Reward(robot=phys, config=cfg)
    cases = []
    for d in ([1, 0], [1, 1]):
        for _ in range(12):
            p0 = rng.uniform(-0.3, 0.3, 3); p1 = p0 + rng.uniform(-0.05, 0.12, 3) * np.array([1, 0.3, 1])
            go = bool(rng.integers(0, 2)); ctr = rng.choice([0.0, -1.0, 0.5], 2); gr = int(rng.integers(0, 4))
            r = R.agent_reward(p0, p1, np.array(d), go, ctr, gr)
            cases.append(dict(init=p0.tolist(), final=p1.tolist(), dir=d, gripper_open=go, controls=ctr.tolist(), grasped=gr, reward=float(r)))
    # the bonus branch fed directly
    r = R.agent_reward(np.array([0., 0, 0.1]), np.array([0.05, 0, 0.1]), np.array([1, 0]), False, np.array([-1., -1.]), 3)
    cases.append(dict(init=[0., 0, 0.1], final=[0.05, 0, 0.1], dir=[1, 0], gripper_open=False, controls=[-1., -1.], grasped=3, reward=float(r)))
    G["agent_reward"] = cases

    # ---- transform_depth
    cases = []
    for k in range(4):
        dep = (rng.uniform(0.02, 3.0, (8, 8)) if k < 3 else rng.uniform(0.02, 0.6, (8, 8))).astype(np.float32)
        out = ut.transform_depth(dep.copy())
        cases.append(dict(depth=dep.astype(float).tolist(), pixels=np.asarray(out, dtype=float).tolist(),
                          u8=np.asarray(out).astype(np.uint8).tolist()))
    G["transform_depth"] = cases

    # ---- euler helpers used by the controller
    cases = []
    for _ in range(10):
        ang = rng.uniform(-1.4, 1.4, 3)
        M = tr.euler_matrix(*ang, 'sxyz')
        back = tr.euler_from_matrix(M, 'sxyz')
        q = rng.normal(size=4); q /= np.linalg.norm(q)
        e = tr.euler_from_quaternion(q, axes=(0, 0, 0, 1))
        cases.append(dict(angles=ang.tolist(), matrix=M[:3, :3].tolist(), back=list(map(float, back)), quat_wxyz=q.tolist(), euler_rzyx=list(map(float, e))))
    G["euler"] = cases

    # ---- AugmentedNatureCNN: parameter count, output shape, one forward with patterned weights
    import torch
    spaces = sys.modules["gym.spaces"]
    space = spaces.Dict({"observation": spaces.Box(0, 255, shape=(5, 64, 64), dtype=np.uint8)})
    net = fe_mod.AugmentedNatureCNN(space, features_dim=514)
    nparam = sum(p.numel() for p in net.parameters())
    with torch.no_grad():
        for k, (name, p) in enumerate(net.named_parameters()):
            idx = torch.arange(p.numel(), dtype=torch.float64)
            p.copy_((torch.sin(idx * 0.37 + k) * 0.05).reshape(p.shape).float())
        x = (torch.arange(2 * 5 * 64 * 64, dtype=torch.float64) * 0.7919).remainder(256).floor().reshape(2, 5, 64, 64).float() / 255.0
        y = net({"observation": x})
    G["feature_extractor"] = dict(n_params=int(nparam), out_shape=list(y.shape), param_names=[n for n, _ in net.named_parameters()],
                                  out_first8=y[0, :8].double().tolist(), out_last4=y[:, -4:].double().tolist(),
                                  out_sum=float(y.double().sum()))

    os.makedirs(os.path.dirname(OUT), exist_ok=True)
    with open(OUT, "w") as f:
        json.dump(G, f, indent=1)
    print("wrote", OUT, {k: (len(v) if isinstance(v, list) else 1) for k, v in G.items()})


if __name__ == "__main__":
    main()
