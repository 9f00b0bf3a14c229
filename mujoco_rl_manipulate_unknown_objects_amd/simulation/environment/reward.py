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
    """reward.py:44-77 (KL of grey-level histograms). SURVEY.md §8(f) n2: next, not built yet."""

    def __init__(self, robot=None, config=None):
        raise NotImplementedError("IntrinsicReward (--im_reward) is not built yet: SURVEY.md §8(f) n2")
