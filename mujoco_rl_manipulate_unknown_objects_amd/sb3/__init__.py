"""Stable-Baselines3-shaped stand-ins (SB3 itself is not installed where this framework runs)."""
from .ppo import PPO  # noqa: F401
from .sac import SAC, HerReplayBuffer, ReplayBuffer, FlatReplayBuffer, FlatHerReplayBuffer, SACPolicy  # noqa: F401
from .policies import ActorCriticPolicy  # noqa: F401
from .torch_layers import BaseFeaturesExtractor  # noqa: F401
from .vec_env import DummyVecEnv, GpuVecEnv, Monitor, VecVideoRecorder  # noqa: F401
from .callbacks import BaseCallback, CallbackList, EvalCallback, evaluate_policy  # noqa: F401
