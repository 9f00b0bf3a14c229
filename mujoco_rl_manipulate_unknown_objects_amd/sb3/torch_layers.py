"""Stand-in for ``stable_baselines3.common.torch_layers`` (SB3 is not installed where this runs).
Same class shape the reference subclasses at models/feature_extractor.py:4,7."""
from torch import nn


class BaseFeaturesExtractor(nn.Module):
    def __init__(self, observation_space, features_dim: int = 0):
        super().__init__()
        assert features_dim > 0
        self._observation_space = observation_space
        self._features_dim = features_dim

    @property
    def features_dim(self) -> int:
        return self._features_dim
