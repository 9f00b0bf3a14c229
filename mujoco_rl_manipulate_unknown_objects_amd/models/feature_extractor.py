"""AugmentedNatureCNN (reference models/feature_extractor.py:7-49), same module names so that a
state_dict of the reference loads unchanged: ``cnn.{0,2,4}`` convolutions, ``linear.0``.

NatureCNN over the image channels (all but the last) -> 512 features, concatenated with the two
scalars the environment writes into the sensor-pad channel (robot_env.py:281-283): 514 features,
602 784 parameters for the default 5 x 64 x 64 observation. Runs on PyTorch-ROCm (MIOpen /
hipBLASLt); ``channels_last`` + bf16 autocast are applied by the PPO policy, not here.
"""
import torch as th
from torch import nn

from ..sb3.torch_layers import BaseFeaturesExtractor


class _Conv1U8(th.autograd.Function):
    """First layer of the extractor for TRAINING on raw uint8 CUDA observations: forward = grip_conv1_u8 (normalisation, Conv2d(4, 32, 8, 4),
    bias and ReLU as one launch, the kernel the rollouts use: 78 us per 4096 samples against 64 + 256 + 30 us for the uint8 ->
    float pass, the tensor library's implicit GEMM and the ReLU); backward = ReLU mask, then the library's weight-gradient kernel on the
    float image, which is recomputed from the bytes (64 us) instead of being kept from the forward (268 MB per 4096-sample minibatch).
    The input needs no gradient. Same arithmetic as the reference path up to summation order (tests/test_gpu_env_api.py)."""

    @staticmethod
    def forward(ctx, obs, weight, bias):
        from ..engine import conv1_u8
        out, other = conv1_u8(obs, weight, bias)
        ctx.save_for_backward(obs, weight, out)
        ctx.mark_non_differentiable(other)
        return out, other

    @staticmethod
    def backward(ctx, gout, _gother):
        from ..engine import obs_preprocess
        obs, weight, out = ctx.saved_tensors
        g = th.ops.aten.threshold_backward(gout.contiguous(memory_format=th.channels_last), out, 0.0)      # ReLU
        x, _ = obs_preprocess(obs)
        _, gw, gb = th.ops.aten.convolution_backward(g, x, weight, [weight.shape[0]], [4, 4], [0, 0], [1, 1], False, [0, 0], 1, [False, True, True])
        return None, gw, gb


class _CnnTrunk(th.autograd.Function):
    """The three convolutions + ReLUs for TRAINING on raw uint8 CUDA observations, hand-written in both directions (csrc/grip_policy.hip,
    csrc/grip_train.hip; fp32 products and sums on the matrix cores -- the tensor library's fp32 arithmetic up to summation order,
    tests/test_gpu_train_kernels.py). Forward: grip_conv1_u8 and grip_conv23 as in the rollouts, additionally writing y2 and the three ReLU masks
    as bits. Backward: one launch (grip_trunk_backward) for both data gradients, the three ReLU masks and the first layer's weight and bias
    gradient straight from the observation bytes and all three bias gradients; the other two layers' weight gradients stay with the tensor
    library (at 65 % of the fp32 MFMA rate they are not where the time goes). Per 4096-sample minibatch on MI355X: forward 198 us (tensor library 497), backward of these layers ~0.44 ms (~1.0 ms)."""

    @staticmethod
    def forward(ctx, obs, index, w1, b1, w2, b2, w3, b3):
        from ..engine import conv1_u8, conv23_prep, conv23, IndexedRows
        ctx.indexed = index is not None                  # obs = the whole observation store, index = the minibatch's rows of it
        y1, other, m1 = conv1_u8(IndexedRows(obs, index) if ctx.indexed else obs, w1, b1, with_mask=True)
        b2m, b3m = conv23_prep(w2, w3)
        y3, y2, m2, m3 = conv23(y1, b2m, b2, b3m, b3, train=True)
        ctx.save_for_backward(obs, index if ctx.indexed else obs.new_empty(0), w1, w2, w3, y1, y2, m1, m2, m3, b2m, b3m)
        ctx.mark_non_differentiable(other)
        return y3, other

    @staticmethod
    def backward(ctx, g3, _gother):
        from ..engine import trunk_backward, IndexedRows
        obs, index, w1, w2, w3, y1, y2, m1, m2, m3, b2m, b3m = ctx.saved_tensors
        g3m, g2m, gw1, (gb1, gb2, gb3), _ = trunk_backward(g3.contiguous(memory_format=th.channels_last), m3, m2, m1, IndexedRows(obs, index) if ctx.indexed else obs,
                                                           b3m, b2m, w1)
        cb = th.ops.aten.convolution_backward
        gw3 = cb(g3m, y2, w3, None, [1, 1], [0, 0], [1, 1], False, [0, 0], 1, [False, True, False])[1]
        gw2 = cb(g2m, y1, w2, None, [2, 2], [0, 0], [1, 1], False, [0, 0], 1, [False, True, False])[1]
        return None, None, gw1, gb1, gw2, gb2, gw3, gb3


class _LinearReluCat(th.autograd.Function):
    """cat(relu(x @ W^T + b), other) with the ReLU derivative and the bias gradient as one pass over the leading columns of the incoming gradient
    (engine.relu_backward_colsum) instead of a slice copy, a threshold_backward and a column reduction; `other` gets no gradient."""

    @staticmethod
    def forward(ctx, x, W, b, other):
        h = th.relu_(th.addmm(b, x, W.t()))
        ctx.save_for_backward(x, W, h)
        return th.cat((h, other.to(h.dtype)), dim=1)

    @staticmethod
    def backward(ctx, g):
        from ..engine import relu_backward_colsum
        x, W, h = ctx.saved_tensors
        gz, gb = relu_backward_colsum(g[:, :h.shape[1]], h)
        return th.mm(gz, W), th.mm(gz.t(), x), gb, None


class AugmentedNatureCNN(BaseFeaturesExtractor):
    accepts_raw_uint8 = True        # forward() normalises raw uint8 CUDA observations itself (one fused kernel)
    fused_first_layer_training = True   # the update's first layer through grip_conv1_u8 too (_Conv1U8); False: the tensor library's convolution
    fused_trunk_training = True         # the update's three convolutions hand-written in both directions (_CnnTrunk); False: the above
    accepts_indexed_rows = True         # forward() takes engine.IndexedRows (a minibatch as row numbers of the rollout storage) where the fused trunk applies

    def __init__(self, observation_space, features_dim: int = 514):
        super().__init__(observation_space, features_dim)
        shape = observation_space["observation"].shape
        n_input_channels = shape[0] - 1
        self.cnn = nn.Sequential(
            nn.Conv2d(n_input_channels, 32, kernel_size=8, stride=4, padding=0), nn.ReLU(),
            nn.Conv2d(32, 64, kernel_size=4, stride=2, padding=0), nn.ReLU(),
            nn.Conv2d(64, 64, kernel_size=3, stride=1, padding=0), nn.ReLU(),
            nn.Flatten())
        with th.no_grad():
            n_flatten = self.cnn(th.zeros(1, n_input_channels, shape[1], shape[2])).shape[1]
        self.linear = nn.Sequential(nn.Linear(n_flatten, features_dim - 2), nn.ReLU())

    # ---- rollout-side features for ActorCriticPolicy's merged-heads forward (no autograd): the fused first layer, the library's
    # convolutions, and the linear layer applied to the NHWC tensor as it lies in memory (its weight's columns re-ordered once per
    # refresh) instead of to the NCHW copy nn.Flatten makes of a channels_last tensor.
    _wl_nhwc = None
    _b23 = None             # second / third convolution's weights as grip_conv23's GEMM operands (fixed addresses)
    _b23_store = None
    _c1_terms = None        # the first convolution's weight as grip_conv1_u8's B operand (three bf16 terms), split once per refresh

    @th.no_grad()
    def refresh_rollout_cache(self):
        lw = self.linear[0].weight
        c3 = self.cnn[4].out_channels
        hw = int(round((lw.shape[1] // c3) ** 0.5))
        if self._wl_nhwc is None or self._wl_nhwc.device != lw.device:
            self._wl_nhwc = th.empty_like(lw)
        self._wl_nhwc.copy_(lw.view(lw.shape[0], c3, hw, hw).permute(0, 2, 3, 1).reshape(lw.shape[0], -1))
        c2, c3m = self.cnn[2], self.cnn[4]
        self._b23 = None
        c0 = self.cnn[0]
        if lw.is_cuda and tuple(c0.weight.shape) == (32, 4, 8, 8) and c0.weight.dtype == th.float32:
            from ..engine import conv1_prep
            self._c1_terms = conv1_prep(c0.weight, self._c1_terms if (self._c1_terms is not None and self._c1_terms.device == c0.weight.device) else None)
        else:
            self._c1_terms = None
        if (lw.is_cuda and tuple(c2.weight.shape) == (64, 32, 4, 4) and c2.stride == (2, 2) and c2.padding == (0, 0) and tuple(c3m.weight.shape) == (64, 64, 3, 3)
                and c3m.stride == (1, 1) and c3m.padding == (0, 0) and c2.weight.dtype == th.float32):
            from ..engine import conv23_prep
            self._b23_store = conv23_prep(c2.weight, c3m.weight, *(self._b23_store or (None, None)))
            self._b23 = self._b23_store

    def rollout_features(self, observations):
        """[B, features_dim] for raw uint8 CUDA observations of the default layout, None otherwise (the caller then uses forward())."""
        obs = observations["observation"]
        c0 = self.cnn[0]
        if not (self._wl_nhwc is not None and obs.dtype == th.uint8 and obs.is_cuda and not th.is_grad_enabled() and tuple(obs.shape[1:]) == (5, 64, 64)
                and tuple(c0.weight.shape) == (32, 4, 8, 8) and c0.stride == (4, 4) and c0.padding == (0, 0) and c0.weight.dtype == th.float32
                and not th.is_autocast_enabled()):
            return None
        from ..engine import conv1_u8
        x, other = conv1_u8(obs.contiguous(), c0.weight, c0.bias, prepared=self._c1_terms)      # (a tensor, or engine.RecordRows: the tick's record rows, read in place)
        if self._b23 is not None:                                   # both remaining convolutions + ReLUs as one f32-MFMA launch
            from ..engine import conv23
            x = conv23(x, self._b23[0], self.cnn[2].bias, self._b23[1], self.cnn[4].bias)
        else:
            x = th.relu_(self.cnn[2](x)); x = th.relu_(self.cnn[4](x))
        xf = x.permute(0, 2, 3, 1).reshape(x.shape[0], -1)          # a view: the tensor is channels_last
        return th.cat((th._addmm_activation(self.linear[0].bias, xf, self._wl_nhwc.t()), other), dim=1)      # (the ReLU in the GEMM's epilogue)

    def forward(self, observations, num_direct_features: int = 2) -> th.Tensor:
        obs = observations["observation"]
        c0 = self.cnn[0]
        index = None
        if hasattr(obs, "index") and hasattr(obs, "records"):          # engine.IndexedRows
            if self.fused_trunk_training and th.is_grad_enabled() and not th.is_autocast_enabled() and num_direct_features == 2:
                obs, index = obs.records, obs.index
            else:
                obs = obs.materialize()
        if (index is None and obs.dtype == th.uint8 and obs.is_cuda and not th.is_grad_enabled() and num_direct_features == 2 and tuple(obs.shape[1:]) == (5, 64, 64)
                and tuple(c0.weight.shape) == (32, 4, 8, 8) and c0.stride == (4, 4) and c0.padding == (0, 0) and c0.weight.dtype == th.float32):
            # rollouts (no autograd): normalisation, first convolution, bias and ReLU in one f32-MFMA launch (csrc/grip_policy.hip)
            from ..engine import conv1_u8
            x, other = conv1_u8(obs.contiguous(), c0.weight, c0.bias)
            for layer in list(self.cnn)[2:]:
                x = layer(x)
            return th.cat((self.linear(x), other.to(x.dtype)), dim=1)
        c2, c3m = self.cnn[2], self.cnn[4]
        if (self.fused_trunk_training and obs.dtype == th.uint8 and obs.is_cuda and th.is_grad_enabled() and num_direct_features == 2
                and tuple(obs.shape[1:]) == (5, 64, 64) and tuple(c0.weight.shape) == (32, 4, 8, 8) and c0.stride == (4, 4) and c0.padding == (0, 0)
                and c0.weight.dtype == th.float32 and not th.is_autocast_enabled() and tuple(c2.weight.shape) == (64, 32, 4, 4) and c2.stride == (2, 2)
                and c2.padding == (0, 0) and tuple(c3m.weight.shape) == (64, 64, 3, 3) and c3m.stride == (1, 1) and c3m.padding == (0, 0)):
            y3, other = _CnnTrunk.apply(obs.contiguous(), index, c0.weight, c0.bias, c2.weight, c2.bias, c3m.weight, c3m.bias)
            # the linear layer on the NHWC tensor as it lies in memory, its weight's columns re-ordered to match (2 MB moved instead of the
            # activations' 17 MB in each direction; the gradient comes back already channels-last)
            lw = self.linear[0].weight
            w_nhwc = lw.view(lw.shape[0], 64, 4, 4).permute(0, 2, 3, 1).reshape(lw.shape[0], -1)
            return _LinearReluCat.apply(y3.permute(0, 2, 3, 1).reshape(y3.shape[0], -1), w_nhwc, self.linear[0].bias, other)
        if index is not None:                          # the fused trunk did not apply after all (another architecture): the gathered rows, below
            obs = obs[index]
        if (self.fused_first_layer_training and obs.dtype == th.uint8 and obs.is_cuda and th.is_grad_enabled() and num_direct_features == 2
                and tuple(obs.shape[1:]) == (5, 64, 64) and tuple(c0.weight.shape) == (32, 4, 8, 8) and c0.stride == (4, 4) and c0.padding == (0, 0)
                and c0.weight.dtype == th.float32 and not th.is_autocast_enabled()):
            x, other = _Conv1U8.apply(obs.contiguous(), c0.weight, c0.bias)
            for layer in list(self.cnn)[2:]:
                x = layer(x)
            return th.cat((self.linear(x), other.to(x.dtype)), dim=1)
        if obs.dtype == th.uint8 and obs.is_cuda and num_direct_features == 2 and obs.shape[2:] == (64, 64):
            # raw uint8 observation on the GPU (the policies pass it through un-normalised): cast, / 255 and the NHWC layout
            # in one kernel instead of three passes (csrc/grip_render.hip: k_obs_preprocess)
            from ..engine import obs_preprocess
            x, other = obs_preprocess(obs.contiguous())
            return th.cat((self.linear(self.cnn(x)), other), dim=1)
        other = obs[:, -1, 0, :num_direct_features]        # grasp code, pheromone level (already / 255)
        # the channel slice is a strided view: MIOpen only has a naive kernel for non-packed inputs (157 ms vs 2 ms
        # per 4096-sample fwd+bwd on MI355X), so pack it first
        x = obs[:, :-1]
        x = x.contiguous(memory_format=th.channels_last) if x.is_cuda else x.contiguous()
        img = self.linear(self.cnn(x))
        return th.cat((img, other.to(img.dtype)), dim=1)
