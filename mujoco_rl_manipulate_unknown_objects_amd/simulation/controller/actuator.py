"""Action space of the reference (simulation/controller/actuator.py:217-247). The controller
arithmetic itself (get_target_pose, scale_control, check_grasp, pheromone_level) runs inside the
macro-step kernel: csrc/grip_sim.hip."""
import numpy as np

from ... import spaces


class Actuator:
    def __init__(self, robot=None, config=None):
        self.physics = robot
        self.config = config

    def setup_action_space(self):
        shape = (6,) if self.config.include_roll else (5,)
        self.action_space = spaces.Box(-1., 1., shape=shape, dtype=np.float32)
        return self.action_space
