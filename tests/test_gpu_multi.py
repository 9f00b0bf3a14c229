"""-m gpu, needs TWO GPUs (skipped on the one-GPU test box; the driver's 8-GPU node runs it): the N > 1 path over RCCL.

`python bench.py --gpus 2` without a launcher must spawn one rank per GPU itself (a child process, before any GPU call), run the
time-sliced rollouts + PPO updates with one flattened-gradient all-reduce per minibatch over RCCL, print ONE JSON line and keep
the two policy replicas bit-identical."""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_two_ranks_over_rccl_stay_bit_identical():
    import torch
    if torch.cuda.device_count() < 2:            # counting devices does not initialise the GPU in this process
        pytest.skip("needs two GPUs")
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "4", "--warmup", "2", "--envs", "512", "--rollout", "2",
                        "--minibatch", "512", "--preroll", "4", "--no-cpu-baseline", "--object", "sand_ball"], capture_output=True, text=True, timeout=900, env=env)
    assert r.returncode == 0, r.stderr[-3000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, r.stdout[-2000:]
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and d["scaling"] == "weak" and d["value"] > 0
    assert d["replicas_identical"] is True
    assert d["env_steps_counted"] == 2 * 512 * 4 and d["short_rollouts"] == 0 and d["update_path"].startswith("explicit")


def test_two_ranks_rehearsed_on_one_gpu_with_gloo():
    """The same N > 1 code path on the one-GPU box: `bench.py --gpus 2` without a launcher (it spawns its ranks), both ranks on device 0,
    gloo instead of RCCL (GRIP_BENCH_REHEARSAL=1: never a headline number). Exercises the self-launch, the flat gradient bucket as
    .grad storage under the captured update graphs (channels_last parameters included), the learn-loop agreement and the replica check."""
    import torch
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    env.update(GRIP_BENCH_REHEARSAL="1", HSA_ENABLE_IPC_MODE_LEGACY="0")
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "6", "--warmup", "2", "--envs", "1024", "--rollout", "2",
                        "--minibatch", "1024", "--preroll", "8", "--no-cpu-baseline", "--object", "sand_ball"], capture_output=True, text=True, timeout=600, env=env)
    assert r.returncode == 0, r.stderr[-3000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, r.stdout[-2000:]
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and d["replicas_identical"] is True and d["env_steps_counted"] == 2 * 1024 * 6 and d["short_rollouts"] == 0
    assert d["update_path"].startswith("explicit")          # the flat gradient buffer of sb3/fused_update.py is the all-reduce's bucket
