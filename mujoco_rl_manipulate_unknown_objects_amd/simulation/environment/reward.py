"""Host-side mirror of simulation/environment/reward.py (numpy, vectorised over envs).

The macro-step kernel computes this reward on the GPU (csrc/grip_sim.hip: agent_reward); this
class exists so that ``RobotEnv.compute_reward(achieved_goal, desired_goal, info)`` keeps the
reference's signature (robot_env.py:243-273), e.g. for HER-style relabelling on the host.
"""
import numpy as np

from ..utils.utils import project_to_target_direction


class Reward:
    """reward.py:8-41: progress of the object along the target direction."""

    def __init__(self, robot=None, config=None):
        self.physics = robot
        self.config = config

    def __call__(self, obs, new_obs, init_obj_pos, final_obj_pos, target_dir, gripper_open, controls, object_grasped):
        return self.agent_reward(init_obj_pos, final_obj_pos, target_dir, gripper_open, controls, object_grasped)

    def agent_reward(self, init_obj_pos, final_obj_pos, target_dir, gripper_open, controls, object_grasped):
        p0 = np.asarray(init_obj_pos, dtype=np.float64); p1 = np.asarray(final_obj_pos, dtype=np.float64)
        d = np.asarray(target_dir, dtype=np.float64)
        a = project_to_target_direction(p0[..., :2], d); b = project_to_target_direction(p1[..., :2], d)
        lateral = np.linalg.norm(b[..., None] * d - p1[..., :2], axis=-1)
        travel = b - a
        ok = (travel > 0.) & (travel < 0.1) & (lateral < 0.1)
        reward = np.where(ok, travel, 0.)
        controls = np.asarray(controls, dtype=np.float64)
        # `np.all(controls) != 0` of the original: both gripper controls non-zero
        bonus = ok & ~np.asarray(gripper_open, dtype=bool) & np.all(controls != 0, axis=-1) & (np.asarray(object_grasped) == 3)
        reward = np.where(bonus, reward * 2, reward)
        reward = np.where(bonus & (p1[..., 2] > 0), reward * 1.5, reward)
        return reward * 30


class IntrinsicReward(Reward):
    """reward.py:44-77: progress reward + KL-style novelty of the observation (grey-level and depth histograms).

    Host mirror in numpy for ``compute_reward``; the rollout path adds the same quantity on the GPU
    (csrc/grip_render.hip: k_intrinsic_reward). cv2 is not a dependency here: BGR2GRAY on uint8 is OpenCV 4.8's fixed-point
    formula and calcHist an exact count (see oracle/grip_render.c for the provenance)."""

    def __init__(self, robot=None, config=None):
        super().__init__(robot, config)

    def __call__(self, obs, new_obs, init_obj_pos, final_obj_pos, target_dir, gripper_open, controls, object_grasped):
        return (self.agent_reward(init_obj_pos, final_obj_pos, target_dir, gripper_open, controls, object_grasped)
                + self.intrinsic_reward(obs, new_obs))

    @staticmethod
    def _pdf(img):
        hist = np.bincount(np.asarray(img, dtype=np.uint8).reshape(-1), minlength=256).astype(np.float32)
        return hist / hist.sum()

    @staticmethod
    def _rel_entr_sum(p, q):
        with np.errstate(divide="ignore", invalid="ignore"):
            t = np.where((p > 0) & (q > 0), p * np.log(p / np.where(q > 0, q, 1)), 0.0).astype(np.float32)
        return float(t.sum())

    def intrinsic_reward(self, obs, new_obs):
        obs = np.asarray(obs); new_obs = np.asarray(new_obs)                      # CHW uint8
        grey = lambda o: ((o[0].astype(np.int64) * 3735 + o[1].astype(np.int64) * 19235 + o[2].astype(np.int64) * 9798 + (1 << 14)) >> 15)
        reward = self._rel_entr_sum(self._pdf(grey(obs)), self._pdf(grey(new_obs)))
        if self.config is None or getattr(self.config, "full_observation", True):
            reward = (reward + self._rel_entr_sum(self._pdf(obs[3]), self._pdf(new_obs[3]))) / 2
        return float(reward)
