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


class _MergedHeads(th.autograd.Function):
    """features [n, F] -> (mean [n, A], values [n]) through the policy | value MLPs (two tanh layers each, same shapes) and the action / value heads, as
    ONE chain with a hand-written backward: first layers = one GEMM with the weights concatenated, second layers and the two heads = batch-of-two
    GEMMs (the heads zero-padded to 8 outputs), and in the backward the tanh derivative, the bias gradient and the change of layout between the
    batch-of-two and the concatenated form are one pass per layer (engine.tanh_backward_colsum). The tensor library's fp32 GEMMs, the same sums as the
    separate modules up to their order (tests/test_gpu_train_kernels.py); about 0.1 ms less per 4096-row minibatch than autograd over the merged forward."""

    @staticmethod
    def forward(ctx, f, w0p, b0p, w0v, b0v, w1p, b1p, w1v, b1v, wa, ba, wv, bv):
        n, A, H = f.shape[0], wa.shape[0], w1p.shape[0]
        W0 = th.cat((w0p, w0v)); h1 = th.tanh_(th.addmm(th.cat((b0p, b0v)), f, W0.t()))                    # [n, 2 H0]
        H0 = w0p.shape[0]
        W1 = th.stack((w1p, w1v))                                                                              # [2, H, H0]
        h2 = th.tanh_(th.baddbmm(th.stack((b1p, b1v)).unsqueeze(1), h1.view(n, 2, H0).transpose(0, 1), W1.transpose(1, 2)))      # [2, n, H]
        Wo = f.new_zeros((2, 8, H)); Wo[0, :A] = wa; Wo[1, 0] = wv[0]
        bo = f.new_zeros((2, 1, 8)); bo[0, 0, :A] = ba; bo[1, 0, 0] = bv[0]
        o = th.baddbmm(bo, h2, Wo.transpose(1, 2))                                                             # [2, n, 8]
        ctx.save_for_backward(f, h1, h2, W0, W1, Wo)
        ctx.A = A
        return o[0, :, :A], o[1, :, 0]

    @staticmethod
    def backward(ctx, g_mean, g_values):
        from ..engine import tanh_backward_colsum
        f, h1, h2, W0, W1, Wo = ctx.saved_tensors
        n, A, H0 = f.shape[0], ctx.A, W0.shape[0] // 2
        go = f.new_zeros((2, n, 8))
        if g_mean is not None:
            go[0, :, :A] = g_mean
        if g_values is not None:
            go[1, :, 0] = g_values
        gWo = th.bmm(go.transpose(1, 2), h2); gbo = go.sum(1)                                                  # [2, 8, H], [2, 8]
        gz2, gb1 = tanh_backward_colsum(th.bmm(go, Wo), h2)                                                    # [2, n, H], [2 H]
        h1b = h1.view(n, 2, H0).transpose(0, 1)
        gW1 = th.bmm(gz2.transpose(1, 2), h1b)                                                                 # [2, H, H0]
        gz1, gb0 = tanh_backward_colsum(th.bmm(gz2, W1), h1, batch_major_to_rows=True)                         # [n, 2 H0] row-major, [2 H0]
        gW0 = th.mm(gz1.t(), f); gf = th.mm(gz1, W0)
        H = W1.shape[1]
        return (gf, gW0[:H0], gb0[:H0], gW0[H0:], gb0[H0:], gW1[0], gb1[:H], gW1[1], gb1[H:], gWo[0, :A], gbo[0, :A], gWo[1, :1], gbo[1, :1])


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
            if hasattr(o, "materialize"):        # rows handed over in place (engine.RecordRows / IndexedRows): the tensor path wants a tensor
                o = o.materialize()
            o = o.float()
            if self.normalize_images:
                o = o / 255.0
        return {"observation": o}

    # the update's forward with the two MLPs as ONE chain (autograd through it): first layers = one GEMM on the shared features with the weights
    # concatenated, deeper layers = a batch-of-two GEMM; backward likewise (one weight-gradient GEMM, one input-gradient GEMM and one bias sum per
    # layer instead of two). The same fp32 sums as the separate modules; [4096 x 256] x [256 x 256] GEMMs alone fill a quarter of the chip.
    merged_heads_training = True
    fused_heads_training = True      # ... and with the hand-written backward of _MergedHeads where the MLPs are two tanh layers each (the default net_arch)

    def _merged_ok(self):
        ok = getattr(self, "_merged_ok_cache", None)
        if ok is None:
            pl = [m for m in self.policy_net if isinstance(m, nn.Linear)]; vl = [m for m in self.value_net_mlp if isinstance(m, nn.Linear)]
            ok = (self.vf_features_extractor is None and len(pl) == len(vl) >= 1 and all(a.weight.shape == b.weight.shape for a, b in zip(pl, vl))
                  and all(isinstance(m, (nn.Linear, nn.Tanh)) for m in list(self.policy_net) + list(self.value_net_mlp))
                  and len(list(self.policy_net)) == 2 * len(pl) and len(list(self.value_net_mlp)) == 2 * len(vl))
            self._merged_ok_cache = (pl, vl) if ok else False
            ok = self._merged_ok_cache
        return ok

    def _mean_values(self, obs):
        """(mean [n, A], values [n]) of the Gaussian policy and the value function"""
        p = self._prep(obs)
        f = self.features_extractor(p)
        m = self._merged_ok() if (self.merged_heads_training and th.is_grad_enabled() and f.is_cuda) else False
        if m and self.fused_heads_training and len(m[0]) == 2 and self.action_net.weight.shape[0] <= 8 and f.dtype == th.float32 and not th.is_autocast_enabled():
            pl, vl = m
            mean, values = _MergedHeads.apply(f, pl[0].weight, pl[0].bias, vl[0].weight, vl[0].bias, pl[1].weight, pl[1].bias, vl[1].weight, vl[1].bias,
                                              self.action_net.weight, self.action_net.bias, self.value_net.weight, self.value_net.bias)
            return mean, values
        lp, lv = self._latents_from(f, p, m)
        return self.action_net(lp).float(), self.value_net(lv).float().squeeze(-1)

    def _latents(self, obs):
        p = self._prep(obs)
        f = self.features_extractor(p)
        return self._latents_from(f, p, self._merged_ok() if (self.merged_heads_training and th.is_grad_enabled() and f.is_cuda) else False)

    def _latents_from(self, f, p, m):
        if m:
            pl, vl = m
            h = th.tanh(th.addmm(th.cat((pl[0].bias, vl[0].bias)), f, th.cat((pl[0].weight, vl[0].weight)).t()))      # [n, 2 o]
            n, o = h.shape[0], pl[0].weight.shape[0]
            hb = h.view(n, 2, o).transpose(0, 1)                                                                         # [2, n, o], a view
            for a, b in zip(pl[1:], vl[1:]):
                hb = th.tanh(th.baddbmm(th.stack((a.bias, b.bias)).unsqueeze(1), hb, th.stack((a.weight, b.weight)).transpose(1, 2)))
            return hb[0], hb[1]
        fv = f if self.vf_features_extractor is None else self.vf_features_extractor(p)
        return self.policy_net(f), self.value_net_mlp(fv)

    @staticmethod
    def _log_prob(mean, log_std, actions):
        var = th.exp(2 * log_std)
        return (-((actions - mean) ** 2) / (2 * var) - log_std - 0.5 * math.log(2 * math.pi)).sum(-1)

    def forward(self, obs, deterministic=False):
        mean, values = self._mean_values(obs)
        log_std = self.log_std.float()
        actions = mean if deterministic else mean + th.randn_like(mean) * th.exp(log_std)
        return actions, values, self._log_prob(mean, log_std, actions)

    def accepts_record_rows(self):
        """can forward_parts() take engine.RecordRows (the tick's observations read in place from the trainer's record rows)?"""
        fe = self.features_extractor
        return (self._rollout_cache is not None and getattr(fe, "_wl_nhwc", None) is not None and getattr(self, "_fused_preprocess", False)
                and bool(getattr(self, "normalize_images", True)))

    def forward_parts(self, obs):
        """(mean, log_std, values): the Gaussian head left un-sampled, for callers that sample and score in a fused kernel."""
        if self._rollout_cache is not None and not th.is_grad_enabled():
            return self._forward_parts_merged(obs)
        o = obs["observation"]
        if hasattr(o, "materialize") and not (hasattr(o, "index") and th.is_grad_enabled() and getattr(self.features_extractor, "accepts_indexed_rows", False)
                                              and getattr(self, "_fused_preprocess", False) and self.vf_features_extractor is None):
            obs = {"observation": o.materialize()}
        mean, values = self._mean_values(obs)
        return mean, self.log_std.float(), values

    # ---- rollout-side forward with merged heads (no autograd): the policy and value MLPs have the same shape, so their first layers
    # are one GEMM on the shared features, the deeper ones block-diagonal, the action and value heads one [A + 1]-row GEMM; the
    # extractor hands over its features without the NCHW copy nn.Flatten makes of a channels_last tensor. Half the launches of the
    # module-by-module forward (232 -> 192 us per 1024 rows on MI355X, tools/policy_fwd_bench.py), the same sums up to fp32 rounding.
    # The merged weights live at fixed addresses (a captured rollout tick keeps reading them): refresh_rollout_cache() after
    # every change of the parameters -- PPO calls it at the start of each collect_rollouts().
    _rollout_cache = None

    def enable_rollout_cache(self):
        """Switch forward_parts() under no_grad to the merged-heads path when the architecture allows it; returns whether it did."""
        pl = [m for m in self.policy_net if isinstance(m, nn.Linear)]; vl = [m for m in self.value_net_mlp if isinstance(m, nn.Linear)]
        ok = (self.vf_features_extractor is None and len(pl) == len(vl) >= 1 and all(a.weight.shape == b.weight.shape for a, b in zip(pl, vl))
              and all(isinstance(m, (nn.Linear, nn.Tanh)) for m in list(self.policy_net) + list(self.value_net_mlp))
              and hasattr(self.features_extractor, "rollout_features") and self.action_net.weight.is_cuda)
        if not ok:
            return False
        dev, A = self.action_net.weight.device, self.action_dim
        c = {"W": [], "b": [], "pl": pl, "vl": vl}
        for i, (a, b) in enumerate(zip(pl, vl)):
            o, k = a.weight.shape
            c["W"].append(th.zeros(2 * o, k if i == 0 else 2 * k, device=dev)); c["b"].append(th.zeros(2 * o, device=dev))
        hd = pl[-1].weight.shape[0]
        c["Wh"] = th.zeros(A + 1, 2 * hd, device=dev); c["bh"] = th.zeros(A + 1, device=dev)
        self._rollout_cache = c
        self.refresh_rollout_cache()
        return True

    @th.no_grad()
    def refresh_rollout_cache(self):
        c = self._rollout_cache
        if c is None:
            return
        for i, (a, b) in enumerate(zip(c["pl"], c["vl"])):
            o, k = a.weight.shape
            if i == 0:
                c["W"][i][:o].copy_(a.weight); c["W"][i][o:].copy_(b.weight)
            else:
                c["W"][i][:o, :k].copy_(a.weight); c["W"][i][o:, k:].copy_(b.weight)
            c["b"][i][:o].copy_(a.bias); c["b"][i][o:].copy_(b.bias)
        A, hd = self.action_dim, c["pl"][-1].weight.shape[0]
        c["Wh"][:A, :hd].copy_(self.action_net.weight); c["Wh"][A, hd:].copy_(self.value_net.weight[0])
        c["bh"][:A].copy_(self.action_net.bias); c["bh"][A:].copy_(self.value_net.bias)
        self.features_extractor.refresh_rollout_cache()
        c["stamp"] = self._param_stamp()

    def _param_stamp(self):
        """changes whenever a parameter is written in place (optimizer step, load_state_dict, copy_) or replaced"""
        return tuple((id(p), p._version) for p in self.parameters())

    def _forward_parts_merged(self, obs):
        c = self._rollout_cache
        # stale copies are refreshed here too (train() followed by an evaluation, load_state_dict, a foreign optimizer step ...), except
        # inside a stream capture, where PPO has refreshed them before capturing and the graph must keep reading the same addresses
        if c.get("stamp") != self._param_stamp() and not (self.action_net.weight.is_cuda and th.cuda.is_current_stream_capturing()):
            self.refresh_rollout_cache()
        h = self.features_extractor.rollout_features(self._prep(obs))
        if h is None:                                               # not the observation layout the extractor's fast path handles
            if hasattr(obs["observation"], "materialize"):
                obs = {"observation": obs["observation"].materialize()}
            mean, values = self._mean_values(obs)
            return mean, self.log_std.float(), values
        for W, b in zip(c["W"], c["b"]):
            h = th.tanh_(th.addmm(b, h, W.t()))
        o = th.addmm(c["bh"], h, c["Wh"].t())
        return o[:, :self.action_dim], self.log_std.float(), o[:, self.action_dim]

    def evaluate_actions(self, obs, actions):
        mean, values = self._mean_values(obs)
        log_std = self.log_std.float()
        entropy = (0.5 + 0.5 * math.log(2 * math.pi) + log_std).sum(-1).expand(mean.shape[0])
        return values, self._log_prob(mean, log_std, actions), entropy

    def predict_values(self, obs):
        _, lv = self._latents(obs)
        return self.value_net(lv).float().squeeze(-1)
