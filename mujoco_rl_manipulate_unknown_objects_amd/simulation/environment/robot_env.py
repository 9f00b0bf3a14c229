"""Gym-shaped environment surfaces over the HIP rollout engine.

``BatchedRobotEnv``  N reference environments on one GPU; tensors in, tensors out.
``RobotEnv(config)`` the reference's class (simulation/environment/robot_env.py:16), one env,
                     numpy in / numpy out, same reset / step / compute_reward / render / seed /
                     close members, same ``info`` keys (robot_env.py:226-241) -- a drop-in for
                     train_agent.py:17 and eval_agent.py:32.
Every number comes from ``libgrip_sim.so`` (csrc/); nothing here falls back to the CPU.
"""
import os
from enum import Enum
from types import SimpleNamespace

import numpy as np

from ... import engine
from ..controller.actuator import Actuator
from ..controller.sensor import RGBDSensor
from .reward import Reward, IntrinsicReward  # noqa: F401


def default_config(**overrides):
    """config/base_config.py defaults as a Namespace (handy for tests and bench)."""
    from ...config.base_config import _FLAGS
    ns = SimpleNamespace(**{name: default for name, _, default, _ in _FLAGS})
    for k, v in overrides.items():
        setattr(ns, k, v)
    return ns


def _object_name(sim_env):
    return os.path.basename(sim_env).replace("_env.xml", "")


class Status(Enum):
    RUNNING = 0
    FAIL = 1
    TIME_LIMIT = 2


class BatchedRobotEnv:
    metadata = {'render.modes': ['human', 'rgb_array', 'depth_array']}
    Status = Status

    def __init__(self, config, n_envs=1, device_index=0, auto_reset=False):
        self.config = config
        self.n_envs = int(n_envs)
        self.im_reward = bool(getattr(config, "im_reward", False))          # robot_env.py:39-42
        if config.width_capture != 64 or config.height_capture != 64:
            raise ValueError("the observation kernel renders 64x64 (config/base_config.py:18-19 defaults)")
        self.batch = self._make_batch(config, device_index, auto_reset)
        self.device = self.batch.device
        self._sensor = RGBDSensor(config=config)
        self._actuator = Actuator(config=config)
        self._reward_fn = IntrinsicReward(config=config) if self.im_reward else Reward(config=config)
        self.setup_spaces()
        # TWO observation buffers, used in turn: the dict a reset / step returns stays valid until the step after the next one, so a
        # caller may hold the previous observation across a step (rollout / replay buffers do) without copying 20 KB per env first
        self._obs = self.batch.torch.empty((self.n_envs, self.batch.obs_channels, 64, 64), dtype=self.batch.torch.uint8, device=self.device)
        self._obs_prev = self.batch.torch.empty_like(self._obs)

    @staticmethod
    def _direction_vector(direction):
        if direction == 0:                             # robot_env.py:30-33
            return np.array([1, 0])
        if direction == 45:
            return np.array([1, 1])                    # unnormalised, as the reference has it
        raise ValueError("direction must be 0 or 45 (robot_env.py:30-33)")

    @staticmethod
    def _engine_flags(config, auto_reset):
        return dict(max_steps=config.max_steps, time_horizon=config.time_horizon,
                    include_roll=int(bool(config.include_roll)), full_observation=int(bool(config.full_observation)),
                    her_buffer=int(bool(config.her_buffer)), auto_reset=int(bool(auto_reset)),
                    max_translation=config.max_translation, max_rotation=config.max_rotation,
                    pos_tolerance=config.pos_tolerance, grasp_tolerance=config.grasp_tolerance)

    def _make_batch(self, config, device_index, auto_reset):
        self.target_direction = self._direction_vector(config.direction)
        return engine.Batch(_object_name(config.sim_env), self.n_envs, device_index, target_dir=self.target_direction,
                            **self._engine_flags(config, auto_reset))

    def setup_spaces(self):
        self.action_space = self._actuator.setup_action_space()
        self.observation_space = self._sensor.setup_observation_space()

    def _obs_dict(self, out):
        self._obs, self._obs_prev = self._obs_prev, self._obs          # the buffer handed out last time is left alone for one more call
        self.batch.observe(self._obs)
        # the goals are the engine's result arrays, rewritten in place by the next step: hand out copies (16 B per env)
        return {"observation": self._obs, "achieved_goal": out["achieved_goal"].clone(), "desired_goal": out["desired_goal"].clone()}

    def reset(self, mask=None):
        """robot_env.py:56-75 for every env (or the masked ones)."""
        return self._obs_dict(self.batch.reset(mask))

    def step(self, actions):
        """robot_env.py:77-241 for every env. Returns (obs, reward[N], done[N] bool, info dict of tensors)."""
        out = self.batch.step(actions)
        obs = self._obs_dict(out)                            # robot_env.py:186-197: _obs_prev = old_obs, the observation before this step
        if self.im_reward:
            # with auto-reset the new observation of a finished env is its reset state, as DummyVecEnv would return it
            self.batch.add_intrinsic_reward(self._obs_prev, self._obs, out["reward"])
        return obs, out["reward"], out["done"].bool(), out

    def compute_reward(self, achieved_goal, desired_goal, info):
        """robot_env.py:243-273 on host arrays (HER-style relabelling); the step kernel already returned this."""
        if isinstance(info, np.ndarray):
            info = info[0]
        r = self._reward_fn(info.get("old_obs"), info.get("new_obs"), info["init_obj_pos"], info["final_obj_pos"], info["target_dir"],
                            info["gripper_open"], info["controls"], info["object_grasped"])
        if self.config.her_buffer:
            dist = np.linalg.norm(np.asarray(desired_goal) - np.asarray(achieved_goal), axis=-1)
            r = r + 1 / np.exp(dist)
        return r

    # -- RobotEnv.render (robot_env.py:302-340): camera 0 workbench, 1 upper (both `targetbodycom` on the object), 2 gripper, 3 = all
    def _camera_pose(self, camera_id, env_index):
        """(optical centre, rotation, fovy) of a reference camera for one env; None pose = the model's gripper camera.
        A `targetbodycom` camera keeps its body position, looks at the object's centre of mass (camera -z) and keeps world z up
        (MuJoCo's camera tracking [3P-recall]: z = unit(cam - target), x = unit(z_world x z), y = z x x)."""
        from ...model import blob
        # the env's OWN model: a mixed batch holds one per (object, direction) group, with its own object inertial frame and static cameras
        part = self.batch.part_of(env_index)[0] if hasattr(self.batch, "part_of") else self.batch
        if not hasattr(self, "_mdls"):
            self._mdls = {}
        mdl = self._mdls.get(part.model.path)
        if mdl is None:
            mdl = self._mdls[part.model.path] = blob.read_blob(part.model.path)
        self._mdl = mdl
        if camera_id == 2:
            return None, None, float(self._mdl["cam_fovy"][0])
        q = self.batch.get_state()[0][env_index].astype(np.float64)
        w, x, y, z = q[10:14] / np.linalg.norm(q[10:14])
        Ro = np.array([[1 - 2 * (y * y + z * z), 2 * (x * y - w * z), 2 * (x * z + w * y)],
                       [2 * (x * y + w * z), 1 - 2 * (x * x + z * z), 2 * (y * z - w * x)],
                       [2 * (x * z - w * y), 2 * (y * z + w * x), 1 - 2 * (x * x + y * y)]])
        target = q[7:10] + Ro @ self._mdl["body_ipos"][7]
        pos = self._mdl["static_cam_pos"][camera_id]
        zc = pos - target; zc /= np.linalg.norm(zc)
        xc = np.cross([0.0, 0.0, 1.0], zc)
        xc = np.array([1.0, 0.0, 0.0]) if np.linalg.norm(xc) < 1e-9 else xc / np.linalg.norm(xc)
        yc = np.cross(zc, xc)
        return pos, np.stack([xc, yc, zc], axis=1), float(self._mdl["static_cam_fovy"][camera_id])

    def render_images(self, camera_id, w_zoom=1, h_zoom=1, env_index=0):
        """RGBDSensor.render_images (sensor.py:56-77): (rgb uint8 [H, W, 3], depth through transform_depth [H, W])."""
        from ..utils.utils import transform_depth
        W, H = int(self.config.width_capture * w_zoom), int(self.config.height_capture * h_zoom)
        pos, R, fovy = self._camera_pose(camera_id, env_index)
        rgb = self.batch.render_camera(env_index, pos, R, fovy, W, H).cpu().numpy()
        depth = self.batch.render_camera(env_index, pos, R, fovy, W, H, depth=True).cpu().numpy()
        return rgb, transform_depth(depth)

    def render(self, mode='human', env_index=0):
        """robot_env.py:302-340: the configured camera (config.camera_id in 0..2) or cameras 0..camera_id-1 side by side, at the zoomed
        size (64 * rendering_zoom_width x 64 * rendering_zoom_height); 'rgb_array' / 'depth_array' return the image, 'human' shows it
        (cv2 window when OpenCV is installed; otherwise the frame is kept in `last_frame` for the caller to display or save)."""
        kind = 'depth' if mode == 'depth_array' else 'rgb'
        def view(cam):
            rgb, depth = self.render_images(cam, self.config.rendering_zoom_width, self.config.rendering_zoom_height, env_index)
            return depth if kind == 'depth' else rgb
        cid = int(self.config.camera_id)
        result = view(cid) if cid in (0, 1, 2) else np.hstack([view(c) for c in range(cid)])
        if mode in ('rgb_array', 'depth_array'):
            return result
        if mode == 'human':
            self.last_frame = result
            try:
                import cv2
                cv2.imshow("Camera", cv2.cvtColor(result, cv2.COLOR_BGR2RGB)); cv2.waitKey(1)
            except ImportError:
                pass
            return None
        raise ValueError(f"unknown render mode {mode!r}")

    def seed(self, seed=None):
        self.np_random = np.random.default_rng(seed)
        return self.np_random

    def close(self):
        self.batch.close()


class MixedBatchedRobotEnv(BatchedRobotEnv):
    """Groups of envs with different objects and target directions on one GPU (BASELINE.json configs[3]), sorted by
    (object, direction) so that every wavefront is homogeneous. `config.sim_env` / `config.direction` are ignored in favour
    of `objects` x `directions`; env e belongs to group e // envs_per_group, groups ordered object-major.
    Steps lock-step (reset / step) or time-sliced (PPO(async_slice=...): the ready-list capacity must be a multiple of the
    number of groups)."""

    def __init__(self, config, objects=("acorn", "sand_ball", "sugar_cube", "bread_crumb"), directions=(0, 45), envs_per_group=512,
                 device_index=0, auto_reset=False):
        self.group_specs = [(o, int(envs_per_group), d) for o in objects for d in directions]
        super().__init__(config, n_envs=len(self.group_specs) * int(envs_per_group), device_index=device_index, auto_reset=auto_reset)

    def _make_batch(self, config, device_index, auto_reset):
        groups = [(_object_name(o), n, self._direction_vector(d)) for o, n, d in self.group_specs]
        b = engine.MixedBatch(groups, device_index, **self._engine_flags(config, auto_reset))
        self.target_direction = b.target_dirs.cpu().numpy()                 # [n_envs, 2]
        return b


class RobotEnv(BatchedRobotEnv):
    """One environment with the reference's numpy surface (robot_env.py:16)."""

    def __init__(self, config):
        super().__init__(config, n_envs=1, auto_reset=False)
        self.obs = dict()
        self.episode_rewards = np.zeros(config.time_horizon)
        self.status = Status.RUNNING
        self.episode_step = 0
        self.gripper_open = True

    def _np_obs(self, obs):
        self.obs["observation"] = obs["observation"][0].cpu().numpy()
        self.obs["achieved_goal"] = obs["achieved_goal"][0].cpu().numpy()
        self.obs["desired_goal"] = obs["desired_goal"][0].cpu().numpy()
        return self.obs

    def reset(self):
        obs = super().reset()
        self.episode_step = 0
        self.episode_rewards = np.zeros(self.config.time_horizon)
        self.status = Status.RUNNING
        self.gripper_open = True
        return self._np_obs(obs)

    def step(self, action):
        torch = self.batch.torch
        old_obs = self.obs.get("observation")
        old_obs = None if old_obs is None else old_obs.copy()
        a = torch.as_tensor(np.asarray(action, dtype=np.float32)).reshape(1, -1)
        obs, reward, done, out = super().step(a)
        o = {k: v[0].cpu().numpy() for k, v in out.items()}
        self._np_obs(obs)
        reward = float(o["reward"]); done = bool(o["done"])
        self.status = Status(int(o["status"])); self.gripper_open = bool(o["gripper_open"])
        step_idx = int(o["episode_step"]) - 1
        if 0 <= step_idx < len(self.episode_rewards):
            self.episode_rewards[step_idx] = reward
        self.episode_step = int(o["episode_step"])
        pr = int(o["position_reached"])
        info = {"old_obs": old_obs, "new_obs": self.obs["observation"],
                "init_obj_pos": o["init_obj_pos"].astype(np.float64), "final_obj_pos": o["object_position"].astype(np.float64),
                "target_dir": self.target_direction, "gripper_open": self.gripper_open,
                "controls": np.zeros(2), "object_grasped": int(o["object_grasped"]),
                "episode_step": self.episode_step, "episode_rewards": self.episode_rewards, "status": self.status,
                "gripper_position": o["gripper_position"].astype(np.float64), "object_position": o["object_position"].astype(np.float64),
                "position_reached": {"target": bool(pr & 1), "initial": bool(pr & 2), "fail": bool(pr & 4)},
                "total_distance": float(o["total_distance"]), "line_distance": float(o["line_distance"]),
                "n_substeps": int(o["n_substeps"])}
        return self.obs, reward, done, info

    def get_observation(self):
        return self.batch.observe(self.batch.torch.empty_like(self._obs))[0].cpu().numpy()
