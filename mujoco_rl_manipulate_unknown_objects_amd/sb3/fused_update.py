"""One PPO minibatch step -- forward, loss, backward -- of the default policy as an EXPLICIT launch sequence instead of an autograd graph.

What stable_baselines3's ``PPO.train`` does per minibatch for the reference's configuration (train_agent.py:18-47: ``AugmentedNatureCNN`` features,
``net_arch=[256, 256]``, shared extractor, fp32) is a fixed chain of ~20 GEMM-sized kernels. Under autograd the chain drags ~50 launches of a few
microseconds each along -- concatenating the policy | value weights for the merged GEMMs, zero-filling and slicing the padded heads, four gathers of the
minibatch's samples, scaling the loss kernel's gradients by 1.0, accumulating 21 gradients into ``.grad``, zeroing them again -- about a sixth of the
captured update's 1.19 ms (profiles/r03_update_path.txt). Here (1.19 -> 1.11 ms per 4096-sample minibatch on MI355X, tools/update_time.py)

* the PARAMETERS live in one flat buffer laid out so that the merged operands are views, not copies: ``[policy_net.0.weight | value_net_mlp.0.weight]`` IS the
  first layers' [2 H0, F] matrix, the second layers' weights are a [2, H, H0] batch, ``action_net.weight`` / ``value_net.weight`` are rows 0..A-1 / row 0 of
  the two zero-padded [8, H] head matrices (the padding rows are not parameters: no optimiser touches them). ``state_dict()`` is unchanged -- same names,
  same shapes, same values;
* the GRADIENTS live in a second buffer of the same layout (every ``.grad`` is a view of it: it is also the one bucket the multi-GPU path all-reduces) and every
  kernel of the backward writes its gradient STRAIGHT into its slot -- GEMMs through ``out=``, the hand-written kernels through their output pointers -- so
  nothing is accumulated and nothing has to be zeroed: each slot is overwritten once per step;
* the loss kernel reads mean / value where the heads' GEMM left them and writes d loss / d (heads' output) in that GEMM's layout, with the heads' bias
  gradients (grip_ppo_loss_heads); the minibatch's samples are gathered by one launch;
* (an option, off: the four weight-gradient GEMMs of the MLP part on a second stream beside the data-gradient chain -- they fill a quarter of the chip each, but
  the fork / join events cost more than the overlap buys: 1.14-1.17 ms per minibatch against 1.11.)

The arithmetic is that of the autograd path (the same kernels on the same operands; tests/test_gpu_train_kernels.py compares every gradient). Anything
else -- another extractor or net_arch, autocast, CPU -- keeps the autograd path (``PPO._loss_backward``)."""
import os
import torch as th
from torch import nn

from .policies import ActorCriticPolicy

_WGRAD23_LIBRARY = os.environ.get("GRIP_WGRAD23_LIBRARY", "0") == "1"      # comparison switch: convolution_backward instead of grip_wgrad23


class FusedUpdate:
    @staticmethod
    def applies(ppo):
        pol = ppo.policy
        if not (ppo.device.type == "cuda" and ppo.fused_loss and ppo.autocast_dtype is None and ppo.indexed_minibatches and isinstance(pol, ActorCriticPolicy)):
            return False
        from ..models.feature_extractor import AugmentedNatureCNN
        fe = pol.features_extractor
        if not (type(fe) is AugmentedNatureCNN and fe.fused_trunk_training and pol.vf_features_extractor is None and pol.normalize_images
                and pol.merged_heads_training and pol.fused_heads_training and getattr(pol, "_fused_preprocess", False)):
            return False
        m = pol._merged_ok()
        if not m or len(m[0]) != 2 or pol.action_dim > 8:
            return False
        c0, c2, c4, lin = fe.cnn[0], fe.cnn[2], fe.cnn[4], fe.linear[0]
        if not (tuple(c0.weight.shape) == (32, 4, 8, 8) and c0.stride == (4, 4) and c0.padding == (0, 0) and tuple(c2.weight.shape) == (64, 32, 4, 4)
                and c2.stride == (2, 2) and c2.padding == (0, 0) and tuple(c4.weight.shape) == (64, 64, 3, 3) and c4.stride == (1, 1) and c4.padding == (0, 0)
                and tuple(lin.weight.shape) == (lin.weight.shape[0], 1024) and tuple(ppo.env.observation_space["observation"].shape) == (5, 64, 64)):
            return False
        mine = FusedUpdate._params(pol)
        everything = list(pol.parameters())
        return (len(mine) == len(everything) and {id(p) for p in mine} == {id(p) for p in everything}
                and all(p.is_cuda and p.dtype == th.float32 and p.requires_grad for p in everything))

    @staticmethod
    def _params(pol):
        fe = pol.features_extractor
        pl, vl = pol._merged_ok()
        c0, c2, c4, lin = fe.cnn[0], fe.cnn[2], fe.cnn[4], fe.linear[0]
        return [c0.weight, c0.bias, c2.weight, c2.bias, c4.weight, c4.bias, lin.weight, lin.bias, pl[0].weight, vl[0].weight, pl[0].bias, vl[0].bias,
                pl[1].weight, vl[1].weight, pl[1].bias, vl[1].bias, pol.action_net.weight, pol.value_net.weight, pol.action_net.bias, pol.value_net.bias, pol.log_std]

    fork_weight_grads = False       # the MLP part's weight-gradient GEMMs on a second stream: measured slower (above)

    def __init__(self, ppo):
        pol = self.pol = ppo.policy
        self.ppo = ppo
        dev = self.dev = ppo.device
        fe = pol.features_extractor
        pl, vl = pol._merged_ok()
        self.c0, self.c2, self.c4, self.lin = fe.cnn[0], fe.cnn[2], fe.cnn[4], fe.linear[0]
        A = self.A = pol.action_dim
        H0, F = pl[0].weight.shape
        H = pl[1].weight.shape[0]
        self.H0, self.F, self.H, self.L = int(H0), int(F), int(H), int(self.lin.weight.shape[0])
        # ---- layout (float offsets; every group starts on a 16-byte boundary)
        plan, off = [], 0

        def place(p, numel_slot=None, at=None):
            nonlocal off
            o = off if at is None else at
            plan.append((p, o))
            if at is None:
                off += p.numel() if numel_slot is None else numel_slot
            return o

        def align():
            nonlocal off
            off = (off + 3) // 4 * 4
        for p in (self.c0.weight, self.c0.bias, self.c2.weight, self.c2.bias, self.c4.weight, self.c4.bias, self.lin.weight, self.lin.bias):
            align(); place(p)
        align(); o_w0 = place(pl[0].weight); place(vl[0].weight)
        align(); o_b0 = place(pl[0].bias); place(vl[0].bias)
        align(); o_w1 = place(pl[1].weight); place(vl[1].weight)
        align(); o_b1 = place(pl[1].bias); place(vl[1].bias)
        align(); o_wo = off; place(pol.action_net.weight, at=o_wo); place(pol.value_net.weight, at=o_wo + 8 * self.H); off += 2 * 8 * self.H
        align(); o_bo = off; place(pol.action_net.bias, at=o_bo); place(pol.value_net.bias, at=o_bo + 8); off += 16
        align(); o_ls = place(pol.log_std, numel_slot=8)
        align()
        total = off
        P = self.P = th.zeros(total, dtype=th.float32, device=dev)
        G = self.G = th.zeros(total, dtype=th.float32, device=dev)
        self._plan = plan
        with th.no_grad():
            for p, o in plan:
                v = th.as_strided(P, p.size(), p.stride(), storage_offset=o)
                v.copy_(p.data)
                p.data = v
        self.bind()
        seg = lambda buf, o, shape: buf[o:o + int(th.Size(shape).numel())].view(shape)
        two = lambda buf: dict(W0=seg(buf, o_w0, (2 * H0, F)), b0=seg(buf, o_b0, (2 * H0,)), W1=seg(buf, o_w1, (2, H, H0)), b1=seg(buf, o_b1, (2 * H,)),
                               Wo=seg(buf, o_wo, (2, 8, H)), bo=seg(buf, o_bo, (16,)), ls=seg(buf, o_ls, (8,)))
        self.p, self.g = two(P), two(G)
        self.wl_nhwc = th.empty_like(self.lin.weight)            # the extractor's linear weight with its columns in the NHWC order of y3 as it lies in memory
        self.b23 = (None, None)
        self.side = th.cuda.Stream(dev)
        ppo._flat_grad = G                                       # the multi-GPU path's one bucket

    def bind(self):
        """every parameter's .grad = its view of the gradient buffer (again, should somebody have set it to None or replaced it)"""
        for p, o in self._plan:
            g = p.grad
            if g is None or g.data_ptr() != self.G.data_ptr() + 4 * o or g.stride() != p.stride():
                p.grad = th.as_strided(self.G, p.size(), p.stride(), storage_offset=o)

    def intact(self):
        """are the parameters still the views of the flat buffer (a caller may have re-assigned .data, or moved the module)?"""
        base = self.P.data_ptr()
        return all(p.data_ptr() == base + 4 * o for p, o in self._plan)

    @th.no_grad()
    def loss_backward(self, src, idx):
        from ..engine import conv1_u8, conv23_prep, conv23, trunk_backward, conv23_weight_gradients, tanh_backward_colsum, relu_backward_colsum, ppo_loss_heads, bias_tanh_, IndexedRows
        obs, actions, old_logp, adv, ret = src
        ppo, p, g = self.ppo, self.p, self.g
        n, A, H0, H, L = int(idx.numel()), self.A, self.H0, self.H, self.L
        dev = self.dev
        cur = th.cuda.current_stream(dev)
        fork = self.fork_weight_grads
        rows = IndexedRows(obs, idx)

        def aside(fn):                       # a weight-gradient GEMM beside the chain: the side stream picks up after everything issued so far
            if not fork:
                return fn()
            self.side.wait_stream(cur)
            with th.cuda.stream(self.side):
                return fn()

        # ---- forward
        c0, c2, c4, lin = self.c0, self.c2, self.c4, self.lin
        y1, other, m1 = conv1_u8(rows, c0.weight, c0.bias, with_mask=True)
        self.b23 = conv23_prep(c2.weight, c4.weight, *self.b23)
        b2m, b3m = self.b23
        y3, y2, m2, m3 = conv23(y1, b2m, c2.bias, b3m, c4.bias, train=True)
        self.wl_nhwc.view(L, 4, 4, 64).copy_(lin.weight.view(L, 64, 4, 4).permute(0, 2, 3, 1))
        x = y3.permute(0, 2, 3, 1).reshape(n, 1024)                                              # a view: y3 is channels-last
        h = th._addmm_activation(lin.bias, x, self.wl_nhwc.t())                                  # relu(x W^T + b), the ReLU in the GEMM's epilogue where the library has one
        feat = th.cat((h, other), dim=1)                                                         # [n, F]
        h1 = th.tanh_(th.addmm(p["b0"], feat, p["W0"].t()))                                      # [n, 2 H0]: policy | value first layers
        h1b = h1.view(n, 2, H0).transpose(0, 1)                                                  # [2, n, H0], a view
        h2 = bias_tanh_(th.bmm(h1b, p["W1"].transpose(1, 2)), p["b1"]) if H % 4 == 0 else th.tanh_(th.baddbmm(p["b1"].view(2, 1, H), h1b, p["W1"].transpose(1, 2)))   # [2, n, H]
        o = th.bmm(h2, p["Wo"].transpose(1, 2))                                                  # [2, n, 8]: mean in [0, :, :A], value in [1, :, 0] -- less the heads' biases, which the loss kernel adds
        # ---- loss and d loss / d o (+ the heads' bias gradients and log_std's)
        out, go = ppo_loss_heads(o, p["bo"], p["ls"][:A], actions, old_logp, adv, ret, idx, ppo.clip_range, ppo.ent_coef, ppo.vf_coef, g["bo"], g["ls"])
        # ---- backward of the heads and the two MLPs (weight gradients beside the chain)
        aside(lambda: th.bmm(go.transpose(1, 2), h2, out=g["Wo"]))
        gz2, _ = tanh_backward_colsum(th.bmm(go, p["Wo"]), h2, gb_out=g["b1"])                   # [2, n, H]
        aside(lambda: th.bmm(gz2.transpose(1, 2), h1b, out=g["W1"]))
        gz1, _ = tanh_backward_colsum(th.bmm(gz2, p["W1"]), h1, batch_major_to_rows=True, gb_out=g["b0"])      # [n, 2 H0]
        aside(lambda: th.mm(gz1.t(), feat, out=g["W0"]))
        gf = th.mm(gz1, p["W0"])                                                                 # [n, F]
        # ---- the extractor's linear layer
        gz, _ = relu_backward_colsum(gf[:, :L], h, gb_out=lin.bias.grad)
        aside(lambda: lin.weight.grad.view(L, 64, 4, 4).copy_(th.mm(gz.t(), x).view(L, 4, 4, 64).permute(0, 3, 1, 2)))
        g3 = th.mm(gz, self.wl_nhwc).view(n, 4, 4, 64).permute(0, 3, 1, 2)                       # d loss / d y3, channels-last
        # ---- the three convolutions
        g3m, g2m, _, _, _ = trunk_backward(g3, m3, m2, m1, rows, b3m, b2m, c0.weight, gw_out=c0.weight.grad, gb_out=(c0.bias.grad, c2.bias.grad, c4.bias.grad))
        if _WGRAD23_LIBRARY:                 # the comparison path: the tensor library's two weight-gradient kernels (until round 5 the only one)
            cb = th.ops.aten.convolution_backward
            c4.weight.grad.copy_(cb(g3m, y2, c4.weight, None, [1, 1], [0, 0], [1, 1], False, [0, 0], 1, [False, True, False])[1])
            c2.weight.grad.copy_(cb(g2m, y1, c2.weight, None, [2, 2], [0, 0], [1, 1], False, [0, 0], 1, [False, True, False])[1])
        else:
            conv23_weight_gradients(y1, g2m, y2, g3m, gw2_out=c2.weight.grad, gw3_out=c4.weight.grad)
        if fork:
            cur.wait_stream(self.side)
        # (every intermediate stays referenced up to here: nothing the side stream reads is handed back to the allocator before the join)
        return out[1], out[2], out[0]
