"""Observation space of the reference (simulation/controller/sensor.py:12-54)."""
import numpy as np

from ... import spaces


class RGBDSensor:
    def __init__(self, robot=None, config=None):
        self.physics = robot
        self.config = config

    def setup_observation_space(self):
        c = 5 if self.config.full_observation else 4
        self.state_space = spaces.Dict({
            "observation": spaces.Box(low=0, high=255, shape=(c, self.config.height_capture, self.config.width_capture), dtype=np.uint8),
            "achieved_goal": spaces.Box(low=-np.inf, high=np.inf, shape=(2,), dtype=np.float32),
            "desired_goal": spaces.Box(low=-np.inf, high=np.inf, shape=(2,), dtype=np.float32)})
        return self.state_space
