#!/usr/bin/env python3
"""Build a diagnostic / experimental variant of the library next to the shipped one:
    python tools/build_variant.py stamps -DGRIP_STAMPS           -> csrc/libgrip_sim_stamps.so (tools/stamp_*.py)
    python tools/build_variant.py exp1 -DSOME_EXPERIMENT ...      -> csrc/libgrip_sim_exp1.so   (tools/flag_sweep.py, async_bench.py --lib)
The shipped libgrip_sim.so / libgrip_sim_cold.so are built by engine.build_library()."""
import os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "mujoco_rl_manipulate_unknown_objects_amd", "csrc")
name, extra = sys.argv[1], sys.argv[2:]
out = os.path.join(CSRC, f"libgrip_sim_{name}.so")
cmd = ["/opt/rocm/bin/hipcc", "-O3", "--offload-arch=gfx950", "-std=c++17", "-fno-slp-vectorize", "-fno-strict-aliasing", "-fno-hip-fp32-correctly-rounded-divide-sqrt",
       "-fno-signed-zeros", "-freciprocal-math", "-fgpu-flush-denormals-to-zero", "-mllvm", "-amdgpu-sched-strategy=iterative-ilp", "-shared", "-fPIC"] + extra + ["-o", out] + \
      [os.path.join(CSRC, f) for f in ("grip_sim.hip", "grip_render.hip", "grip_rollout.hip", "grip_policy.hip", "grip_train.hip")]
subprocess.run(cmd, check=True)
print(out)
