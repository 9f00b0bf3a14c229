"""Rollout-side actor-critic forward at the tick's batch size: what the per-tick policy phase costs, and variants."""
import sys, os, time
sys.path.insert(0, os.path.join(os.path.dirname(__file__), ".."))
import torch, torch.nn as nn
from mujoco_rl_manipulate_unknown_objects_amd.simulation.environment.robot_env import default_config
from mujoco_rl_manipulate_unknown_objects_amd.simulation.controller.sensor import RGBDSensor
from mujoco_rl_manipulate_unknown_objects_amd.simulation.controller.actuator import Actuator
from mujoco_rl_manipulate_unknown_objects_amd.sb3.policies import ActorCriticPolicy
from mujoco_rl_manipulate_unknown_objects_amd.models.feature_extractor import AugmentedNatureCNN

B = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
cfg = default_config()
pol = ActorCriticPolicy(RGBDSensor(config=cfg).setup_observation_space(), Actuator(config=cfg).setup_action_space(),
                        features_extractor_class=AugmentedNatureCNN, net_arch=[256, 256]).cuda().to(memory_format=torch.channels_last)
obs = torch.randint(0, 256, (B, 5, 64, 64), dtype=torch.uint8, device="cuda")

def timeit(name, fn, n=50):
    with torch.no_grad():
        for _ in range(5): fn()
        g = torch.cuda.CUDAGraph()
        s = torch.cuda.Stream(); s.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(s):
            fn()
        torch.cuda.current_stream().wait_stream(s); torch.cuda.synchronize()
        with torch.cuda.graph(g):
            out = fn()
        torch.cuda.synchronize(); t = time.perf_counter()
        for _ in range(n): g.replay()
        torch.cuda.synchronize(); dt = (time.perf_counter() - t) / n * 1e6
    print(f"{name:40s} {dt:8.1f} us (graph replay)", flush=True)
    return out

ref = timeit("current forward_parts", lambda: pol.forward_parts({"observation": obs}))

from mujoco_rl_manipulate_unknown_objects_amd.engine import conv1_u8
c0 = pol.features_extractor.cnn[0]
timeit("conv1_u8 alone (prep + f32 MFMA)", lambda: conv1_u8(obs, c0.weight, c0.bias))
fe = pol.features_extractor
convs = [fe.cnn[0], fe.cnn[2], fe.cnn[4]]
from mujoco_rl_manipulate_unknown_objects_amd.engine import obs_preprocess
def fused_relu():
    x, other = obs_preprocess(obs)
    for c in convs:
        x = torch.miopen_convolution_relu(x, c.weight, c.bias, c.stride, c.padding, c.dilation, 1)
    f = torch.cat((fe.linear(x.flatten(1)), other), dim=1)
    return pol.action_net(pol.policy_net(f)), pol.log_std, pol.value_net(pol.value_net_mlp(f)).squeeze(-1)
try:
    out = timeit("miopen_convolution_relu", fused_relu)
    print("max |d mean|", (out[0] - ref[0]).abs().max().item(), "max |d value|", (out[2] - ref[2]).abs().max().item())
except Exception as ex:
    print("miopen_convolution_relu failed:", repr(ex)[:300])

torch.backends.cudnn.benchmark = True
pol2 = ActorCriticPolicy(RGBDSensor(config=cfg).setup_observation_space(), Actuator(config=cfg).setup_action_space(),
                         features_extractor_class=AugmentedNatureCNN, net_arch=[256, 256]).cuda().to(memory_format=torch.channels_last)
timeit("forward_parts, cudnn.benchmark=True", lambda: pol2.forward_parts({"observation": obs}))
with torch.autocast("cuda", dtype=torch.bfloat16):
    timeit("forward_parts bf16 autocast", lambda: pol2.forward_parts({"observation": obs}))
