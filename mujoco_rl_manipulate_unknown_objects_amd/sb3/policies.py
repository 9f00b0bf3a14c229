"""Actor-critic policy for PPO with the reference's ``policy_kwargs`` (train_agent.py:18-20):
``features_extractor_class=None, share_features_extractor=True, net_arch=[256, 256]``.

SB3 conventions kept: the whole uint8 observation is divided by 255 before the extractor (pad scalars
included: quirk Q8), tanh MLPs, state-independent log_std initialised to 0, orthogonal init with gains
sqrt(2) / 0.01 (action head) / 1 (value head). 514 -> 256 -> 256 -> 6 (+ log_std[6]) for the policy,
514 -> 256 -> 256 -> 1 for the value: about 1.0 M parameters with the 602 784 of the extractor.
"""
import math

import numpy as np
import torch as th
from torch import nn



def _mlp(in_dim, arch):
    layers, d = [], in_dim
    for h in arch:
        layers += [nn.Linear(d, h), nn.Tanh()]
        d = h
    return nn.Sequential(*layers), d


class ActorCriticPolicy(nn.Module):
    def __init__(self, observation_space, action_space, features_extractor_class=None,
                 features_extractor_kwargs=None, share_features_extractor=True, net_arch=(256, 256),
                 log_std_init=0.0, ortho_init=True, normalize_images=True):
        super().__init__()
        self.observation_space, self.action_space = observation_space, action_space
        self.normalize_images = normalize_images
        self.share_features_extractor = share_features_extractor
        if features_extractor_class is None:
            from ..models.feature_extractor import AugmentedNatureCNN as features_extractor_class
        kw = features_extractor_kwargs or {}
        self.features_extractor = features_extractor_class(observation_space, **kw)
        self.vf_features_extractor = None if share_features_extractor else features_extractor_class(observation_space, **kw)
        self._fused_preprocess = bool(getattr(self.features_extractor, "accepts_raw_uint8", False))
        fd = self.features_extractor.features_dim
        if isinstance(net_arch, dict):
            pi_arch, vf_arch = net_arch.get("pi", []), net_arch.get("vf", [])
        else:
            pi_arch = vf_arch = list(net_arch)
        self.policy_net, pd = _mlp(fd, pi_arch)
        self.value_net_mlp, vd = _mlp(fd, vf_arch)
        self.action_dim = int(np.prod(action_space.shape))
        self.action_net = nn.Linear(pd, self.action_dim)
        self.value_net = nn.Linear(vd, 1)
        self.log_std = nn.Parameter(th.ones(self.action_dim) * log_std_init)
        if ortho_init:
            gains = [(self.features_extractor, math.sqrt(2)), (self.policy_net, math.sqrt(2)), (self.value_net_mlp, math.sqrt(2)),
                     (self.action_net, 0.01), (self.value_net, 1.0)]
            if self.vf_features_extractor is not None:
                gains.append((self.vf_features_extractor, math.sqrt(2)))
            for mod, g in gains:
                for m in mod.modules():
                    if isinstance(m, (nn.Linear, nn.Conv2d)):
                        nn.init.orthogonal_(m.weight, gain=g)
                        if m.bias is not None:
                            m.bias.data.fill_(0.0)

    # ---- helpers
    def _prep(self, obs):
        o = obs["observation"]
        if o.dtype == th.uint8:
            if o.is_cuda and self.normalize_images and getattr(self, "_fused_preprocess", False):
                return {"observation": o}        # AugmentedNatureCNN normalises and lays out raw uint8 in one kernel
            o = o.float()
            if self.normalize_images:
                o = o / 255.0
        return {"observation": o}

    def _latents(self, obs):
        p = self._prep(obs)
        f = self.features_extractor(p)
        fv = f if self.vf_features_extractor is None else self.vf_features_extractor(p)
        return self.policy_net(f), self.value_net_mlp(fv)

    @staticmethod
    def _log_prob(mean, log_std, actions):
        var = th.exp(2 * log_std)
        return (-((actions - mean) ** 2) / (2 * var) - log_std - 0.5 * math.log(2 * math.pi)).sum(-1)

    def forward(self, obs, deterministic=False):
        lp, lv = self._latents(obs)
        mean = self.action_net(lp).float(); values = self.value_net(lv).float().squeeze(-1)
        log_std = self.log_std.float()
        actions = mean if deterministic else mean + th.randn_like(mean) * th.exp(log_std)
        return actions, values, self._log_prob(mean, log_std, actions)

    def forward_parts(self, obs):
        """(mean, log_std, values): the Gaussian head left un-sampled, for callers that sample and score in a fused kernel."""
        lp, lv = self._latents(obs)
        return self.action_net(lp).float(), self.log_std.float(), self.value_net(lv).float().squeeze(-1)

    def evaluate_actions(self, obs, actions):
        lp, lv = self._latents(obs)
        mean = self.action_net(lp).float(); values = self.value_net(lv).float().squeeze(-1)
        log_std = self.log_std.float()
        entropy = (0.5 + 0.5 * math.log(2 * math.pi) + log_std).sum(-1).expand(mean.shape[0])
        return values, self._log_prob(mean, log_std, actions), entropy

    def predict_values(self, obs):
        _, lv = self._latents(obs)
        return self.value_net(lv).float().squeeze(-1)
