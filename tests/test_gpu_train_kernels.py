"""Hand-written backward of AugmentedNatureCNN's convolutions (csrc/grip_train.hip) against the tensor library's fp32 autograd on the same
inputs (reference models/feature_extractor.py:14-22 trained by stable_baselines3 in fp32). Tolerances: both sides sum fp32 products of the same
operands in different orders, over up to 576 (data gradients) / 4096 x 225 (weight gradients) terms."""
import pytest
import torch as th

pytestmark = pytest.mark.gpu


def _trunk(n, seed=0):
    g = th.Generator(device="cuda").manual_seed(seed)
    dev = "cuda"
    y1 = th.relu(th.randn(n, 32, 15, 15, device=dev, generator=g)).contiguous(memory_format=th.channels_last)
    w2 = (th.randn(64, 32, 4, 4, device=dev, generator=g) * 0.05).contiguous(memory_format=th.channels_last)
    w3 = (th.randn(64, 64, 3, 3, device=dev, generator=g) * 0.05).contiguous(memory_format=th.channels_last)
    b2 = th.randn(64, device=dev, generator=g) * 0.1
    b3 = th.randn(64, device=dev, generator=g) * 0.1
    return y1, w2, b2, w3, b3, g


def _reference(n, seed):
    """fp32 autograd through the three layers on a random uint8 observation: everything the kernels are compared with"""
    y1_, w2, b2, w3, b3, g = _trunk(1, seed)
    obs = th.randint(0, 256, (n, 5, 64, 64), device="cuda", dtype=th.uint8, generator=g)
    w1 = (th.randn(32, 4, 8, 8, device="cuda", generator=g) * 0.05).contiguous(memory_format=th.channels_last).requires_grad_(True)
    b1 = (th.randn(32, device="cuda", generator=g) * 0.1).requires_grad_(True)
    x = (obs[:, :4].float() / 255.0).contiguous(memory_format=th.channels_last)
    p1 = th.nn.functional.conv2d(x, w1, b1, stride=4); y1 = th.relu(p1)
    p2 = th.nn.functional.conv2d(y1, w2, b2, stride=2); y2 = th.relu(p2)
    p3 = th.nn.functional.conv2d(y2, w3, b3); y3 = th.relu(p3)
    g3 = th.randn(y3.shape, device="cuda", generator=g).contiguous(memory_format=th.channels_last)
    for t in (p1, p2, p3): t.retain_grad()
    y3.backward(g3)
    cl = lambda t: t.detach().contiguous(memory_format=th.channels_last)
    return dict(obs=obs, w1=w1, b1=b1, b2=b2, b3=b3, w2=w2, w3=w3, gb2=p2.grad.sum((0, 2, 3)), gb3=p3.grad.sum((0, 2, 3)), y1=cl(y1), y2=cl(y2), y3=cl(y3), g3=g3, g3m=cl(p3.grad), g2m=cl(p2.grad), g1m=cl(p1.grad), gw1=w1.grad, gb1=b1.grad)


def _close(a, b, rel):
    err = (a - b).abs().max().item()
    assert err <= rel * max(1.0, b.abs().max().item()), (err, b.abs().max().item())


def _bits(mask, nbits):
    """[n, P] integer words -> bool [n, nbits, P]"""
    sh = th.arange(nbits, device=mask.device, dtype=mask.dtype).view(1, nbits, 1)
    return ((mask.unsqueeze(1) >> sh) & 1).bool()


@pytest.mark.parametrize("n", [1, 2, 3, 64, 257, 1031])
def test_trunk_backward_matches_autograd(n):
    from mujoco_rl_manipulate_unknown_objects_amd.engine import conv23_prep, conv1_u8, conv23, trunk_backward
    R = _reference(n, seed=n)
    b2m, b3m = conv23_prep(R["w2"], R["w3"])
    # the training forward: activations as the tensor library's, masks = their signs
    y1, other, m1 = conv1_u8(R["obs"], R["w1"].detach(), R["b1"].detach(), with_mask=True)
    y3, y2, m2, m3 = conv23(y1, b2m, R["b2"], b3m, R["b3"], train=True)
    th.cuda.synchronize()
    _close(y1, R["y1"], 1e-5); _close(y2, R["y2"], 1e-5); _close(y3, R["y3"], 1e-5)
    assert th.equal(_bits(m1, 32).view(n, 32, 15, 15), y1 > 0) and th.equal(_bits(m2, 64).view(n, 64, 6, 6), y2 > 0) and th.equal(_bits(m3, 64).view(n, 64, 4, 4), y3 > 0)
    # masks of the REFERENCE activations for the comparison of the gradients (a pre-activation within rounding of zero may flip between the two forwards)
    pack = lambda y, dt: (((y.flatten(2) > 0).to(dt)) << th.arange(y.shape[1], device="cuda", dtype=dt).view(1, -1, 1)).sum(1).contiguous()
    m1r, m2r, m3r = pack(R["y1"], th.int64).to(th.int32), pack(R["y2"], th.int64), pack(R["y3"], th.int64)
    g3m, g2m, gw, gb, g1m = trunk_backward(R["g3"], m3r, m2r, m1r, R["obs"], b3m, b2m, R["w1"].detach(), want_g1m=True)
    th.cuda.synchronize()
    assert th.equal(g3m, R["g3m"])                                       # a mask: exact
    _close(g2m, R["g2m"], 2e-5); _close(g1m, R["g1m"], 2e-5)
    assert ((g2m == 0) == (R["g2m"] == 0)).float().mean().item() > 0.999 and ((g1m == 0) == (R["g1m"] == 0)).float().mean().item() > 0.999       # same masks
    assert gw.stride() == R["w1"].stride()
    _close(gw, R["gw1"], 1e-4); _close(gb[0], R["gb1"], 1e-4); _close(gb[1], R["gb2"], 1e-4); _close(gb[2], R["gb3"], 1e-4)      # sums over n x 225 positions
    # without the on-chip consumer: data gradients only, same values; and run-to-run bit-identical
    a = trunk_backward(R["g3"], m3r, m2r, m1r, None, b3m, b2m)
    assert th.equal(a[1], g2m) and th.equal(a[4], g1m) and a[2] is None
    b = trunk_backward(R["g3"], m3r, m2r, m1r, R["obs"], b3m, b2m, R["w1"].detach())
    assert th.equal(b[2], gw) and all(th.equal(x, y) for x, y in zip(b[3], gb)) and b[4] is None
    # the minibatch as rows of a larger store (engine.IndexedRows): the same bits as on the gathered copy, forward and backward
    from mujoco_rl_manipulate_unknown_objects_amd.engine import IndexedRows
    store = th.randint(0, 256, (2 * n + 3, 5, 64, 64), device="cuda", dtype=th.uint8)
    index = th.randperm(2 * n + 3, device="cuda")[:n]
    store[index] = R["obs"]
    y1i, otheri, m1i = conv1_u8(IndexedRows(store, index), R["w1"].detach(), R["b1"].detach(), with_mask=True)
    assert th.equal(y1i, y1) and th.equal(otheri, other) and th.equal(m1i, m1)
    c = trunk_backward(R["g3"], m3r, m2r, m1r, IndexedRows(store, index), b3m, b2m, R["w1"].detach())
    assert th.equal(c[2], gw) and all(th.equal(x, y) for x, y in zip(c[3], gb))


@pytest.mark.parametrize("n", [1, 2, 3, 5, 64, 257, 1031, 4096])
def test_second_and_third_weight_gradients_match_the_tensor_library(n):
    """grip_wgrad23 (k_wgrad23_b3: bf16 matrix pipe, both operands as three bf16 terms) against convolution_backward in fp64 on the same operands, beside the
    tensor library's own fp32 kernels: fp32-equivalent means the kernel's error is of the fp32 library's size (sums over n x 36 / n x 16 positions in another order)."""
    from mujoco_rl_manipulate_unknown_objects_amd.engine import conv23_weight_gradients
    g = th.Generator(device="cuda").manual_seed(100 + n)
    rnd = lambda *s: th.randn(*s, device="cuda", generator=g)
    cl = lambda t: t.contiguous(memory_format=th.channels_last)
    y1, y2 = cl(th.relu(rnd(n, 32, 15, 15))), cl(th.relu(rnd(n, 64, 6, 6)))
    g2m, g3m = cl(rnd(n, 64, 6, 6) * (rnd(n, 64, 6, 6) > 0)), cl(rnd(n, 64, 4, 4) * (rnd(n, 64, 4, 4) > 0))         # masked gradients: about half zeros
    w2, w3 = cl(rnd(64, 32, 4, 4)), cl(rnd(64, 64, 3, 3))
    cb = th.ops.aten.convolution_backward
    ref = lambda gm, y, w, st, dt: cb(gm.to(dt), y.to(dt), w.to(dt), None, [st, st], [0, 0], [1, 1], False, [0, 0], 1, [False, True, False])[1]
    r2, r3 = ref(g2m, y1, w2, 2, th.float64), ref(g3m, y2, w3, 1, th.float64)
    l2, l3 = ref(g2m, y1, w2, 2, th.float32), ref(g3m, y2, w3, 1, th.float32)
    gw2, gw3 = conv23_weight_gradients(y1, g2m, y2, g3m)
    th.cuda.synchronize()
    for name, mine, lib32, r in (("w2", gw2, l2, r2), ("w3", gw3, l3, r3)):
        scale = r.abs().max().item()
        e_mine, e_lib = (mine.double() - r).abs().max().item() / scale, (lib32.double() - r).abs().max().item() / scale
        print(f"n {n} d{name}: max |error| / max |gradient|: kernel {e_mine:.2e}, tensor library fp32 {e_lib:.2e}")
        assert e_mine <= max(2.0 * e_lib, 2e-6), (name, e_mine, e_lib)
    # into given tensors of other strides (the update's flat gradient buffer holds channels-last views), and run-to-run bit-identical
    o2, o3 = cl(th.empty(64, 32, 4, 4, device="cuda")), cl(th.empty(64, 64, 3, 3, device="cuda"))
    conv23_weight_gradients(y1, g2m, y2, g3m, gw2_out=o2, gw3_out=o3)
    assert th.equal(o2, gw2) and th.equal(o3, gw3) and o2.stride() != gw2.stride()


def test_extractor_trains_the_same_through_the_fused_trunk():
    """AugmentedNatureCNN.forward under autograd: the hand-written trunk (_CnnTrunk + the NHWC linear layer) against the tensor library's modules on
    the same parameters and uint8 observations -- features to 2e-5, every parameter's gradient under a random linear loss to 2e-4 of its largest
    entry (sums over 4096 x up to 225 positions in different orders; a ReLU within rounding of zero may flip between the two forwards)."""
    import numpy as np
    from mujoco_rl_manipulate_unknown_objects_amd import spaces
    from mujoco_rl_manipulate_unknown_objects_amd.models.feature_extractor import AugmentedNatureCNN
    osp = spaces.Dict({"observation": spaces.Box(0, 255, shape=(5, 64, 64), dtype=np.uint8)})
    th.manual_seed(0)
    net = AugmentedNatureCNN(osp).cuda().to(memory_format=th.channels_last)
    with th.no_grad():
        for p in net.parameters():
            if p.ndim == 1:
                p.add_(0.05 * th.randn_like(p))
    obs = th.randint(0, 256, (520, 5, 64, 64), dtype=th.uint8, device="cuda")
    G = th.randn(520, 514, device="cuda")
    out, grads = {}, {}
    for fused in (True, False):
        net.fused_trunk_training = fused; net.fused_first_layer_training = False
        net.zero_grad(set_to_none=True)
        f = net({"observation": obs})
        (f * G).sum().backward()
        out[fused] = f.detach(); grads[fused] = {k: p.grad.clone() for k, p in net.named_parameters()}
    _close(out[True], out[False], 2e-5)
    for k in grads[False]:
        assert grads[True][k].shape == grads[False][k].shape and grads[True][k].stride() == grads[False][k].stride(), k
        err = (grads[True][k] - grads[False][k]).abs().max().item()
        assert err <= 2e-4 * grads[False][k].abs().max().item(), (k, err, grads[False][k].abs().max().item())


def test_merged_heads_training_gradients_match_the_separate_modules():
    """ActorCriticPolicy's update forward with the policy | value MLPs as one chain (cat / batch-of-two GEMMs under autograd) against the separate
    modules: outputs to 2e-5, every parameter's gradient under a random linear loss to 2e-4 of its largest entry (fp32 both ways, sums re-associated)."""
    from mujoco_rl_manipulate_unknown_objects_amd.sb3.policies import ActorCriticPolicy
    from mujoco_rl_manipulate_unknown_objects_amd.models.feature_extractor import AugmentedNatureCNN
    from mujoco_rl_manipulate_unknown_objects_amd.simulation.controller.sensor import RGBDSensor
    from mujoco_rl_manipulate_unknown_objects_amd.simulation.controller.actuator import Actuator
    from mujoco_rl_manipulate_unknown_objects_amd.simulation.environment.robot_env import default_config
    cfg = default_config()
    th.manual_seed(5)
    pol = ActorCriticPolicy(RGBDSensor(config=cfg).setup_observation_space(), Actuator(config=cfg).setup_action_space(),
                            features_extractor_class=AugmentedNatureCNN, net_arch=[256, 256]).cuda().to(memory_format=th.channels_last)
    with th.no_grad():
        for p in pol.parameters():
            if p.ndim == 1:
                p.add_(0.1 * th.randn_like(p))
        pol.action_net.weight.mul_(30.0)
    assert pol._merged_ok()
    obs = {"observation": th.randint(0, 256, (300, 5, 64, 64), dtype=th.uint8, device="cuda")}
    Gm, Gv = th.randn(300, pol.action_dim, device="cuda"), th.randn(300, device="cuda")
    out, grads = {}, {}
    for name, merged, fused in (("hand-written backward", True, True), ("autograd over the merged forward", True, False), ("separate modules", False, False)):
        pol.merged_heads_training = merged; pol.fused_heads_training = fused
        pol.zero_grad(set_to_none=True)
        mean, log_std, values = pol.forward_parts(obs)
        ((mean * Gm).sum() + (values * Gv).sum()).backward()
        out[name] = (mean.detach(), values.detach()); grads[name] = {k: p.grad.clone() for k, p in pol.named_parameters() if p.grad is not None}
    ref = "separate modules"
    for name in out:
        _close(out[name][0], out[ref][0], 2e-5); _close(out[name][1], out[ref][1], 2e-5)
        assert grads[name].keys() == grads[ref].keys()
        for k in grads[ref]:
            err = (grads[name][k] - grads[ref][k]).abs().max().item()
            assert err <= 2e-4 * max(grads[ref][k].abs().max().item(), 1e-6), (name, k, err, grads[ref][k].abs().max().item())
    # the value function alone (predict_values / a loss without the policy term): the unused output's gradient is None
    pol.merged_heads_training = pol.fused_heads_training = True
    pol.zero_grad(set_to_none=True)
    _, _, values = pol.forward_parts(obs)
    (values * Gv).sum().backward()
    assert pol.value_net.weight.grad is not None and th.isfinite(pol.value_net.weight.grad).all()


def test_clip_adam_matches_the_tensor_library():
    """engine.ClipAdam against clip_grad_norm_ + torch.optim.Adam(fused, capturable) on copies of the same parameters and gradients, three steps, once with
    the clip active and once not: parameters, moments, step counters and the clipped gradients to 1e-6 relative (same fp32 formulas, the norm summed in
    a different order)."""
    from mujoco_rl_manipulate_unknown_objects_amd.engine import ClipAdam
    th.manual_seed(1)
    shapes = [(32, 4, 8, 8), (32,), (64, 32, 4, 4), (512, 1024), (512,), (4,), (1, 256), (5000, 3)]
    for scale in (10.0, 1e-3):                          # gradient norm far above / below max_norm = 0.5
        pa = [th.randn(s, device="cuda") for s in shapes]
        pa = [p.contiguous(memory_format=th.channels_last) if p.dim() == 4 else p for p in pa]
        pb = [p.clone(memory_format=th.preserve_format) for p in pa]
        for p in pa + pb: p.requires_grad_(True)
        oa = th.optim.Adam(pa, lr=3e-4, eps=1e-5, capturable=True, fused=True); ob = th.optim.Adam(pb, lr=3e-4, eps=1e-5, capturable=True, fused=True)
        ca = ClipAdam(oa, 0.5)
        for it in range(3):
            for x, y in zip(pa, pb):
                g = th.randn(x.shape, device="cuda") * scale
                if x.dim() == 4: g = g.contiguous(memory_format=th.channels_last)
                x.grad = g.clone(memory_format=th.preserve_format); y.grad = g.clone(memory_format=th.preserve_format)
            assert ca.step()
            th.nn.utils.clip_grad_norm_(pb, 0.5); ob.step()
            th.cuda.synchronize()
            for x, y in zip(pa, pb):
                _close(x.detach(), y.detach(), 1e-6); _close(x.grad, y.grad, 1e-6)
                _close(oa.state[x]["exp_avg"], ob.state[y]["exp_avg"], 1e-6); _close(oa.state[x]["exp_avg_sq"], ob.state[y]["exp_avg_sq"], 1e-6)
                assert float(oa.state[x]["step"]) == float(ob.state[y]["step"]) == it + 1
    sd = oa.state_dict()                                # still a torch Adam state
    assert len(sd["state"]) == len(shapes) and set(sd["state"][0]) == {"step", "exp_avg", "exp_avg_sq"}


def _synthetic_rollout(R, A, seed):
    g = th.Generator(device="cuda").manual_seed(seed)
    obs = th.randint(0, 256, (R, 5, 64, 64), dtype=th.uint8, device="cuda", generator=g)
    actions = th.rand(R, A, device="cuda", generator=g) * 2 - 1
    logp = -A * (0.9 + 0.3 * th.rand(R, device="cuda", generator=g))
    adv = th.randn(R, device="cuda", generator=g); ret = th.randn(R, device="cuda", generator=g)
    return obs, actions, logp, adv, ret


def _default_ppo(seed, **kw):
    from mujoco_rl_manipulate_unknown_objects_amd.sb3 import PPO, GpuVecEnv
    from mujoco_rl_manipulate_unknown_objects_amd.models.feature_extractor import AugmentedNatureCNN
    from mujoco_rl_manipulate_unknown_objects_amd.simulation.environment.robot_env import BatchedRobotEnv, default_config
    env = BatchedRobotEnv(default_config(sim_env="/xmls/sand_ball_env.xml"), n_envs=64, auto_reset=True)
    model = PPO("MultiInputPolicy", GpuVecEnv(env), n_steps=2, batch_size=64, n_epochs=1, seed=seed, async_slice=32, async_capacity=32, async_budget_us=1000,
                policy_kwargs=dict(features_extractor_class=AugmentedNatureCNN, share_features_extractor=True, net_arch=[256, 256]), **kw)
    with th.no_grad():
        for p in model.policy.parameters():
            if p.ndim == 1:
                p.add_(0.05 * th.randn_like(p))
        model.policy.action_net.weight.mul_(20.0)
    return model, env


@pytest.mark.parametrize("n", [64, 1000])
def test_explicit_update_gradients_match_the_autograd_path(n):
    """sb3/fused_update.py -- a minibatch's forward, loss and backward as an explicit launch sequence over flat parameter / gradient buffers -- against
    PPO._loss_backward's autograd graph over the same kernels, on the same parameters and the same synthetic minibatch: the three loss numbers to 1e-6
    relative, every parameter's gradient to 1e-5 of its largest entry (the same kernels on the same operands; only the heads' bias gradients and
    log_std's are summed by another kernel). The re-laid parameters keep names, shapes, strides and values (state_dict() unchanged), and the weight
    gradients computed on a second stream (an option that measured slower: off) are the same as without it."""
    model, env = _default_ppo(3)
    pol = model.policy
    A = pol.action_dim
    src = _synthetic_rollout(1500, A, 11)
    idx = th.randperm(1500, device="cuda")[:n].contiguous()
    before = {k: v.clone() for k, v in pol.state_dict().items()}
    strides = {k: v.stride() for k, v in pol.named_parameters()}
    assert model._fused is not None and model._fused.intact()      # laid out at construction, before anything can capture the parameters' addresses
    model.explicit_update = False
    model._select_update_path(src, idx)
    assert model._fused is None
    pol.zero_grad(set_to_none=True)
    pl0, vl0, loss0 = (float(x) for x in model._loss_backward(src, idx))
    ref = {k: p.grad.clone() for k, p in pol.named_parameters()}
    assert all(v is not None for v in ref.values())
    model.explicit_update = True
    model._select_update_path(src, idx)
    fu = model._fused
    assert fu is not None and fu.intact()
    after = pol.state_dict()
    assert before.keys() == after.keys() and all(th.equal(before[k], after[k]) for k in before)
    assert all(p.stride() == strides[k] and p.grad.stride() == strides[k] for k, p in pol.named_parameters())
    for fork in (True, False):
        fu.fork_weight_grads = fork
        fu.G.fill_(float("nan"))                         # every slot must be written
        pl1, vl1, loss1 = (float(x) for x in model._loss_backward(src, idx))
        th.cuda.synchronize()
        for a, b in ((pl0, pl1), (vl0, vl1), (loss0, loss1)):
            assert abs(a - b) <= 1e-6 * max(1.0, abs(a)), (a, b)
        for k, p in pol.named_parameters():
            assert p.grad.data_ptr() >= fu.G.data_ptr() and th.isfinite(p.grad).all(), k
            err = (p.grad - ref[k]).abs().max().item()
            assert err <= 1e-5 * max(ref[k].abs().max().item(), 1e-6), (fork, k, err, ref[k].abs().max().item())
        pad = fu.g["Wo"].clone(); pad[0, :A] = 0; pad[1, :1] = 0
        assert th.isfinite(fu.G[:-8]).all() and (pad == 0).all()          # the heads' padding rows carry zero gradient (log_std's two pad words are never written)
    # moving a parameter away (a caller's .to(), a re-assigned .data) is noticed: the parameters are laid out again (and a captured graph dropped)
    pol.log_std.data = pol.log_std.data.clone()
    assert not fu.intact()
    model._upd = {"stale": True}
    model._select_update_path(src, idx)
    assert model._fused is not None and model._fused is not fu and model._fused.intact() and model._upd is None
    pl2, vl2, loss2 = (float(x) for x in model._loss_backward(src, idx))
    assert abs(loss2 - loss0) <= 1e-6 * max(1.0, abs(loss0))
    # ... and a storage the sequence does not handle (float observations) goes back to autograd
    model._select_update_path((src[0].float(),) + src[1:], idx)
    assert model._fused is None and all(p.grad is None for p in pol.parameters())
    env.close()


def test_explicit_update_trains_like_the_autograd_path():
    """Two PPO objects from the same seed on the same synthetic rollout, one on the explicit launch sequence (captured into hipGraphs after two eager
    steps, like the bench), one on the autograd path: after six optimiser steps the losses agree to 1e-4 and every parameter to 1e-5 + 1e-2 of the distance it moved."""
    out = {}
    for explicit in (True, False):
        model, env = _default_ppo(7)
        model.explicit_update = explicit
        A = model.policy.action_dim
        src = _synthetic_rollout(768, A, 5)
        start = {k: v.clone() for k, v in model.policy.state_dict().items()}
        g = th.Generator(device="cuda").manual_seed(1)
        losses = []
        for _ in range(6):
            idx = th.randperm(768, device="cuda", generator=g)[:256].contiguous()
            losses.append(float(model._minibatch_update(src, idx)[2]))
        th.cuda.synchronize()
        assert (model._fused is not None) == explicit and model._upd["fwd"] is not None
        out[explicit] = (losses, {k: v.clone() for k, v in model.policy.state_dict().items()}, start)
        env.close()
    (la, pa, s0), (lb, pb, _) = out[True], out[False]
    assert all(abs(a - b) <= 1e-4 * max(1.0, abs(b)) for a, b in zip(la, lb)), (la, lb)
    for k in pb:
        moved = (pb[k] - s0[k]).abs().max().item()
        assert moved > 0 and (pa[k] - pb[k]).abs().max().item() <= 1e-5 + 1e-2 * moved, (k, moved, (pa[k] - pb[k]).abs().max().item())
