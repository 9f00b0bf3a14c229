"""ctypes binding of ``libgrip_sim.so`` (include/grip_sim.h) + torch-tensor plumbing.

The shared library is the product: hand-written HIP kernels for gfx950 behind a
plain C ABI. This module only moves pointers: device memory, streams and
``torch.distributed`` come from PyTorch-ROCm, nothing is computed here and there is
NO CPU fallback -- if the library or a GPU is missing every call raises.
"""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(_HERE, "csrc")
LIB_PATH = os.path.join(CSRC, "libgrip_sim.so")
ASSETS = os.path.join(_HERE, "assets")
OBJECTS = ("acorn", "sand_ball", "sugar_cube", "bread_crumb")
MAXCON = 14


class GripError(RuntimeError):
    pass


COLD_LIB_PATH = os.path.join(CSRC, "libgrip_sim_cold.so")      # -DGRIP_COLD_PORTAL comparison build: see select_library()


def build_library(force=False, verbose=False):
    """hipcc --offload-arch=gfx950 of csrc/*.hip into csrc/libgrip_sim.so and the comparison build csrc/libgrip_sim_cold.so (cross-compiles
    without a GPU)."""
    srcs = [os.path.join(CSRC, f) for f in sorted(os.listdir(CSRC)) if f.endswith((".hip", ".h"))]
    srcs.append(os.path.join(_HERE, "..", "include", "grip_sim.h"))
    newest = max(os.path.getmtime(s) for s in srcs)
    hipcc = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
    procs = []
    for path, extra in ((os.path.join(CSRC, "libgrip_sim.so"), []), (COLD_LIB_PATH, ["-DGRIP_COLD_PORTAL"])):
        if not force and os.path.exists(path) and os.path.getmtime(path) >= newest:
            continue
        tmp = path + f".tmp{os.getpid()}"
        # -fno-hip-fp32-correctly-rounded-divide-sqrt: `/` and sqrtf as v_rcp / v_sqrt + one Newton step (<= 2.5 ulp) instead of the
        # ~10-instruction correctly rounded sequences; -fgpu-flush-denormals-to-zero: no denormal fix-ups. Together +2.6 % physics rate
        # (tools/physics_rate.py); every oracle-parity tolerance holds unchanged. -amdgpu-sched-strategy=iterative-ilp: the physics kernel is bound by
        # instruction issue (two waves per SIMD, each issuing at most one instruction per ~5 cycles: DESIGN.md), and the ILP-first scheduler's order leaves
        # fewer s_waitcnt / s_nop slots in the long dependent chains through LDS: +4.1 % physics rate
        # over the default one (max-ilp +-0, max-memory-clause -0.4 %, -O2 +0.3 %; round 3, tools/physics_rate.py on one box; round 4 on the final kernel: default / max-ilp /
        # max-memory-clause -2.5 %, iterative-maxocc -0.8 %, iterative-minreg -7.5 %).
        # -fno-signed-zeros -freciprocal-math (round 4): x / y may become x * (1 / y), -0 need not be kept apart from +0: +0.6 % physics rate on the same box, 371 against
        # 372 of 379 exactly matching contact-fixture lanes, every parity test unchanged. -fassociative-math is NOT harmless: the solver's compensated sums break (43 M
        # physics.step()/s and 1 % of the macro steps faulting).
        cmd = [hipcc, "-O3", "--offload-arch=gfx950", "-std=c++17", "-fno-slp-vectorize", "-fno-strict-aliasing", "-fno-hip-fp32-correctly-rounded-divide-sqrt",
               "-fno-signed-zeros", "-freciprocal-math", "-fgpu-flush-denormals-to-zero", "-mllvm", "-amdgpu-sched-strategy=iterative-ilp", "-shared", "-fPIC"] + extra + ["-o", tmp,
               os.path.join(CSRC, "grip_sim.hip"), os.path.join(CSRC, "grip_render.hip"), os.path.join(CSRC, "grip_rollout.hip"),
               os.path.join(CSRC, "grip_policy.hip"), os.path.join(CSRC, "grip_train.hip")]
        procs.append((path, tmp, subprocess.Popen(cmd, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True)))
    for path, tmp, pr in procs:                                  # the two builds run side by side
        out, err = pr.communicate()
        if verbose or pr.returncode:
            print(out, err)
        if pr.returncode:
            raise GripError("hipcc failed:\n" + err[-4000:])
        os.replace(tmp, path)      # atomic: concurrent builders never see a half-written library
    return LIB_PATH


def source_fingerprint():
    """sha256 (first 16 hex digits) over the sources the library is built from -- csrc/*.hip, csrc/*.h, include/grip_sim.h, names and contents in sorted order.
    tools/pmc_run.sh records it beside the counters it collects; bench.py attaches a committed counter summary to its line only when the running tree has the same
    fingerprint (round 4's summary had been taken two commits before the shipped build)."""
    import hashlib
    h = hashlib.sha256()
    files = sorted(os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith((".hip", ".h")))
    files.append(os.path.normpath(os.path.join(_HERE, "..", "include", "grip_sim.h")))
    for f in files:
        h.update(os.path.basename(f).encode()); h.update(b"\0")
        with open(f, "rb") as fh:
            h.update(fh.read())
    return h.hexdigest()[:16]


def select_library(cold_portal=False):
    """Choose the build of the library for this process -- before the first call into it. cold_portal: the comparison build whose
    narrow phase starts every portal refinement from scratch, as libccd / MuJoCo do (the shipped build starts a touching pair's
    refinement from the portal it converged to one physics.step() earlier, checked for validity: 20-35 % cheaper steps in
    contact-rich states, trajectories within micrometres of the cold start over tens of steps: tools/warm_portal_probe.py,
    tests/test_gpu_parity.py). GRIP_COLD_PORTAL=1 in the environment selects it too."""
    global LIB_PATH
    if _lib is not None:
        raise GripError("select_library() must come before the first use of the library")
    LIB_PATH = COLD_LIB_PATH if cold_portal else os.path.join(CSRC, "libgrip_sim.so")


class RolloutTickC(C.Structure):
    """GripRolloutTick (include/grip_sim.h)."""
    _fields_ = ([("n_envs", C.c_int32), ("capacity", C.c_int32), ("action_dim", C.c_int32), ("n_records", C.c_int64)] +
                [(n, C.c_void_p) for n in ("ready_list", "ready_count", "base", "reward", "done", "n_substeps", "actions", "values", "log_probs",
                                           "low", "high", "slot_actions", "rec_of_env", "rewards", "dones", "next_rec", "prev_rec", "rec_env",
                                           "completed", "is_rec", "actions_buf", "log_probs_buf", "values_buf", "n_completed", "substeps_total",
                                           "ep_ret", "ep_len", "ep_ret_sum", "ep_len_sum", "ep_count", "noise", "log_std", "rng_count")] +
                [("rng_seed", C.c_uint64), ("mean_stride", C.c_int32), ("value_stride", C.c_int32)])


class EnvConfigC(C.Structure):
    _fields_ = [("max_steps", C.c_int32), ("time_horizon", C.c_int32), ("include_roll", C.c_int32),
                ("full_observation", C.c_int32), ("her_buffer", C.c_int32), ("auto_reset", C.c_int32),
                ("max_translation", C.c_float), ("max_rotation", C.c_float), ("pos_tolerance", C.c_float),
                ("grasp_tolerance", C.c_float), ("target_dir", C.c_float * 2)]


_OUT_FIELDS = [("reward", "float32", ()), ("done", "uint8", ()), ("achieved_goal", "float32", (2,)),
               ("desired_goal", "float32", (2,)), ("status", "int32", ()), ("episode_step", "int32", ()),
               ("gripper_open", "int32", ()), ("object_grasped", "int32", ()), ("position_reached", "int32", ()),
               ("total_distance", "float32", ()), ("line_distance", "float32", ()), ("gripper_position", "float32", (3,)),
               ("object_position", "float32", (3,)), ("init_obj_pos", "float32", (3,)), ("n_substeps", "int32", ()),
               ("fault", "int32", ())]


class StepOutC(C.Structure):
    _fields_ = [(n, C.c_void_p) for n, _, _ in _OUT_FIELDS]


_lib = None
EXPORTS = ["grip_last_error", "grip_model_load", "grip_model_free", "grip_model_nvert", "grip_batch_create",
           "grip_batch_destroy", "grip_batch_set_config", "grip_batch_num_envs", "grip_batch_reset", "grip_batch_step",
           "grip_batch_observe", "grip_batch_get_state", "grip_batch_set_state", "grip_batch_get_flags",
           "grip_batch_set_flags", "grip_batch_substep", "grip_batch_debug_forward", "grip_batch_target_pose",
           "grip_batch_kernel_time", "grip_batch_device_time", "grip_selftest_cholesky", "grip_batch_advance", "grip_batch_observe_list", "grip_rollout_tick", "grip_rollout_gae", "grip_intrinsic_reward", "grip_obs_preprocess",
           "grip_batch_set_state_storage", "grip_batchset_create", "grip_batchset_destroy", "grip_batchset_refresh", "grip_batchset_num_envs",
           "grip_batchset_step", "grip_batchset_advance", "grip_batchset_observe", "grip_batchset_observe_list", "grip_conv1_u8", "grip_conv1_u8_rows", "grip_batch_render_camera", "grip_ppo_loss", "grip_conv23_prep", "grip_conv23",
           "grip_trunk_backward", "grip_trunk_backward_parts", "grip_conv1_u8_train", "grip_conv23_train", "grip_clip_adam", "grip_clip_adam_chunks", "grip_tanh_backward_colsum", "grip_relu_backward_colsum", "grip_ppo_loss_heads", "grip_bias_tanh", "grip_conv1_prep", "grip_wgrad23", "grip_wgrad23_scratch_floats"]


def lib():
    global _lib, LIB_PATH
    if _lib is not None:
        return _lib
    if os.environ.get("GRIP_COLD_PORTAL") == "1" and LIB_PATH == os.path.join(CSRC, "libgrip_sim.so"):
        LIB_PATH = COLD_LIB_PATH
    variant = os.environ.get("GRIP_LIB_VARIANT")           # development only: an experimental build made by tools/build_variant.py
    if variant and LIB_PATH == os.path.join(CSRC, "libgrip_sim.so"):
        LIB_PATH = os.path.join(CSRC, f"libgrip_sim_{variant}.so")
    if not os.path.exists(LIB_PATH):
        raise GripError(f"{LIB_PATH} is not built: run __graft_entry__.build() (hipcc --offload-arch=gfx950). "
                        "There is no CPU fallback for the product path.")
    # PyTorch-ROCm first: it carries its own HIP runtime, and the library must bind to THAT copy (loaded second, libgrip_sim.so would pull
    # /opt/rocm's libamdhip64 in as a second runtime that sees no device once torch has initialised the first)
    import torch  # noqa: F401
    L = C.CDLL(LIB_PATH)
    vp = C.c_void_p
    L.grip_last_error.restype = C.c_char_p
    L.grip_model_load.argtypes = [C.c_char_p, C.POINTER(vp)]
    L.grip_model_free.argtypes = [vp]
    L.grip_model_nvert.argtypes = [vp]
    L.grip_batch_create.argtypes = [vp, C.c_int, C.c_int, C.POINTER(vp)]
    L.grip_batch_destroy.argtypes = [vp]
    L.grip_batch_set_config.argtypes = [vp, C.POINTER(EnvConfigC)]
    L.grip_batch_num_envs.argtypes = [vp]
    L.grip_batch_reset.argtypes = [vp, vp, C.POINTER(StepOutC), vp]
    L.grip_batch_step.argtypes = [vp, vp, C.POINTER(StepOutC), vp]
    L.grip_batch_observe.argtypes = [vp, vp, vp]
    L.grip_batch_get_state.argtypes = [vp, vp, vp, vp, vp, C.c_int, vp]
    L.grip_batch_set_state.argtypes = [vp, vp, vp, vp, vp, C.c_int, vp]
    L.grip_batch_get_flags.argtypes = [vp, vp, vp, vp, vp]
    L.grip_batch_set_flags.argtypes = [vp, vp, vp, vp, vp]
    L.grip_batch_substep.argtypes = [vp, C.c_int, vp]
    L.grip_batch_set_state_storage.argtypes = [vp, C.c_int, vp]
    L.grip_batch_debug_forward.argtypes = [vp] + [vp] * 7 + [vp]
    L.grip_batch_target_pose.argtypes = [vp, vp, vp, vp]
    L.grip_batch_kernel_time.argtypes = [vp, C.c_int, C.POINTER(C.c_float), C.POINTER(C.c_int)]
    if hasattr(L, "grip_batch_device_time"):             # (absent from comparison builds of earlier rounds' sources: tools/physics_rate.py <variant>)
        L.grip_batch_device_time.argtypes = [vp, C.c_int, C.POINTER(C.c_double), C.POINTER(C.c_longlong), vp]
    L.grip_selftest_cholesky.argtypes = [vp, vp, vp, C.c_int, vp]
    L.grip_batch_advance.argtypes = [vp, vp, C.c_int, C.c_int, C.c_int, C.c_int, vp, vp, vp, vp]
    L.grip_batch_observe_list.argtypes = [vp, vp, vp, C.c_int, vp, vp, vp, vp]
    L.grip_intrinsic_reward.argtypes = [vp, vp, vp, vp, vp, C.c_int, C.c_int, C.c_int, vp, vp]
    L.grip_obs_preprocess.argtypes = [vp, C.c_int, C.c_int, vp, vp, vp]
    L.grip_batchset_create.argtypes = [C.POINTER(vp), C.c_int, C.POINTER(StepOutC), C.POINTER(vp)]
    L.grip_batchset_destroy.argtypes = [vp]
    L.grip_batchset_refresh.argtypes = [vp]
    L.grip_batchset_num_envs.argtypes = [vp]
    L.grip_batchset_step.argtypes = [vp, vp, vp]
    L.grip_batchset_advance.argtypes = [vp, vp, C.c_int, C.c_int, C.c_int, C.c_int, vp, vp, vp]
    L.grip_batchset_observe.argtypes = [vp, vp, vp]
    L.grip_batchset_observe_list.argtypes = [vp, vp, C.c_int, vp, vp, vp, vp]
    L.grip_conv1_u8.argtypes = [vp, C.c_int, C.c_int, vp, C.POINTER(C.c_int64), vp, vp, vp, vp, vp]
    L.grip_conv1_u8_rows.argtypes = [vp, vp, C.c_int, C.c_int, vp, C.POINTER(C.c_int64), vp, vp, vp, vp, vp]
    L.grip_conv23_prep.argtypes = [vp, C.POINTER(C.c_int64), vp, C.POINTER(C.c_int64), vp, vp, vp]
    L.grip_conv23.argtypes = [vp, C.c_int, vp, vp, vp, vp, vp, vp]
    L.grip_conv23_train.argtypes = [vp, C.c_int, vp, vp, vp, vp, vp, vp, vp, vp, vp]
    L.grip_conv1_u8_train.argtypes = [vp, vp, vp, C.c_int, C.c_int, vp, C.POINTER(C.c_int64), vp, vp, vp, vp, vp, vp]
    L.grip_trunk_backward.argtypes = [vp] * 6 + [C.c_int, vp, vp, C.c_int] + [vp] * 5 + [C.POINTER(C.c_int64), vp, vp, vp, vp]
    L.grip_trunk_backward_parts.argtypes = [C.c_int]
    L.grip_wgrad23_scratch_floats.argtypes = [C.c_int]; L.grip_wgrad23_scratch_floats.restype = C.c_longlong
    L.grip_wgrad23.argtypes = [vp, vp, vp, vp, C.c_int, vp, vp, C.POINTER(C.c_int64), vp, C.POINTER(C.c_int64), vp]
    L.grip_clip_adam_chunks.argtypes = [C.c_int, C.POINTER(C.c_int64)]
    L.grip_relu_backward_colsum.argtypes = [vp, C.c_int, vp, vp, C.c_int, C.c_int, vp, vp, vp]
    L.grip_tanh_backward_colsum.argtypes = [vp, vp, vp, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int64, vp, vp, vp]
    L.grip_clip_adam.argtypes = [C.c_int, C.POINTER(C.c_int64)] + [C.POINTER(vp)] * 5 + [C.c_float] * 5 + [vp, vp, vp]
    L.grip_ppo_loss.argtypes = [vp] * 7 + [C.c_int, C.c_int, C.c_float, C.c_float, C.c_float] + [vp] * 5
    L.grip_ppo_loss_heads.argtypes = [vp] * 8 + [C.c_int, C.c_int, C.c_float, C.c_float, C.c_float] + [vp] * 6
    L.grip_bias_tanh.argtypes = [vp, vp, C.c_int, C.c_int, C.c_int, vp]
    L.grip_conv1_prep.argtypes = [vp, C.POINTER(C.c_int64), vp, vp]
    L.grip_batch_render_camera.argtypes = [vp, C.c_int, vp, C.c_float, C.c_int, C.c_int, vp, vp, vp]
    L.grip_rollout_tick.argtypes = [vp, vp]
    L.grip_rollout_gae.argtypes = [C.c_int, vp, vp, vp, vp, vp, C.c_float, C.c_float, vp, vp, vp]
    _lib = L
    return L


def _chk(rc):
    if rc != 0:
        raise GripError(lib().grip_last_error().decode())


def asset_path(obj):
    obj = os.path.basename(obj).replace("_env.xml", "").replace("_env.grpm", "")
    p = os.path.join(ASSETS, f"{obj}_env.grpm")
    if not os.path.exists(p):
        raise GripError(f"no compiled model for '{obj}' under {ASSETS}")
    return p


def obs_preprocess(obs):
    """uint8 CUDA observation [n, C, 64, 64] -> (float32 / 255 image channels as a channels-last [n, C - 1, 64, 64] tensor,
    the two sensor-pad scalars / 255 as [n, 2]) in one kernel (grip_obs_preprocess)."""
    import torch
    assert obs.is_cuda and obs.dtype == torch.uint8 and obs.is_contiguous() and obs.shape[2:] == (64, 64)
    n, ch = int(obs.shape[0]), int(obs.shape[1])
    img = torch.empty((n, ch - 1, 64, 64), dtype=torch.float32, device=obs.device, memory_format=torch.channels_last)
    other = torch.empty((n, 2), dtype=torch.float32, device=obs.device)
    stream = C.c_void_p(torch.cuda.current_stream(obs.device).cuda_stream)
    if lib().grip_obs_preprocess(C.c_void_p(obs.data_ptr()), n, ch, C.c_void_p(img.data_ptr()), C.c_void_p(other.data_ptr()), stream) != 0:
        raise GripError("grip_obs_preprocess failed")
    return img, other


class RecordRows:
    """`n` consecutive rows of a uint8 observation store [R, 5, 64, 64], starting at the row a DEVICE scalar holds (int64 [1]): what the
    time-sliced trainer hands to the policy instead of a staging copy of the tick's observations -- the observation kernel renders them once,
    into the record rows, and the first layer (conv1_u8) reads them there. Quacks like the tensor of those rows where the policy looks."""
    dtype_name = "uint8"

    def __init__(self, records, row0, n):
        import torch
        assert records.is_cuda and records.dtype == torch.uint8 and records.is_contiguous() and row0.dtype == torch.int64 and row0.is_cuda
        self.records, self.row0, self.n = records, row0, int(n)
        self.dtype, self.is_cuda, self.device = torch.uint8, True, records.device
        self.shape = (self.n,) + tuple(records.shape[1:])
        self.ndim = records.ndim

    def contiguous(self):
        return self

    def materialize(self):
        """the rows as an ordinary tensor: a gathered copy, without a host sync (the row base stays a device scalar), so that a fallback that
        needs real tensors -- the module-by-module forward under autocast, say -- still works inside a stream capture"""
        import torch
        return self.records.index_select(0, self.row0.reshape(1) + torch.arange(self.n, device=self.records.device))


class IndexedRows:
    """Rows `index` (int64 [n], device) of a uint8 observation store [R, 5, 64, 64]: a minibatch of the update handed to the policy where it lies in the
    rollout storage, instead of a gathered copy (84 MB moved per 4096-sample minibatch). Quacks like the tensor of those rows where the policy looks."""

    def __init__(self, records, index):
        import torch
        assert records.is_cuda and records.dtype == torch.uint8 and records.is_contiguous() and index.dtype == torch.int64 and index.is_cuda and index.dim() == 1
        self.records, self.index, self.n = records, index.contiguous(), int(index.numel())
        self.dtype, self.is_cuda, self.device = torch.uint8, True, records.device
        self.shape = (self.n,) + tuple(records.shape[1:])
        self.ndim = records.ndim

    def contiguous(self):
        return self

    def materialize(self):
        return self.records[self.index]


def conv1_prep(weight, out=None):
    """The first layer's weight / 255 as three bf16 terms, the B operand of grip_conv1_u8 (float32 [12288] scratch; `out`: rewrite it in place -- fixed address for
    a captured rollout tick). conv1_u8(..., prepared=...) then skips the split: once per policy update instead of once per tick."""
    import torch
    assert weight.is_cuda and weight.dtype == torch.float32 and tuple(weight.shape) == (32, 4, 8, 8)
    if out is None:
        out = torch.empty(12288, dtype=torch.float32, device=weight.device)
    strides = (C.c_int64 * 4)(*weight.stride())
    _chk(lib().grip_conv1_prep(C.c_void_p(weight.data_ptr()), strides, C.c_void_p(out.data_ptr()), C.c_void_p(torch.cuda.current_stream(weight.device).cuda_stream)))
    return out


def conv1_u8(obs, weight, bias, with_mask=False, prepared=None):
    """First layer of AugmentedNatureCNN for rollouts (grip_conv1_u8, csrc/grip_policy.hip): uint8 CUDA observations
    [n, 5, 64, 64] -> (relu(conv2d(obs[:, :4] / 255, weight, bias, stride 4)) as a channels-last float32 [n, 32, 15, 15]
    tensor, the two sensor-pad scalars / 255 as [n, 2]) in one launch on the matrix cores (fp32 arithmetic, computed exactly on the bf16 pipe:
    csrc/grip_policy.hip). No autograd. prepared: conv1_prep(weight) of the SAME weight values (the split is then not repeated)."""
    import torch
    rows = obs if isinstance(obs, RecordRows) else None
    idx = obs if isinstance(obs, IndexedRows) else None
    if rows is not None or idx is not None:
        obs = (rows or idx).records
    assert obs.is_cuda and obs.dtype == torch.uint8 and obs.is_contiguous() and tuple(obs.shape[1:]) == (5, 64, 64)
    assert weight.dtype == torch.float32 and tuple(weight.shape) == (32, 4, 8, 8) and bias.dtype == torch.float32 and bias.is_contiguous()
    n = rows.n if rows is not None else idx.n if idx is not None else int(obs.shape[0])
    out = torch.empty((n, 32, 15, 15), dtype=torch.float32, device=obs.device, memory_format=torch.channels_last)
    other = torch.empty((n, 2), dtype=torch.float32, device=obs.device)
    if prepared is not None:
        assert prepared.is_cuda and prepared.dtype == torch.float32 and prepared.numel() == 12288 and prepared.is_contiguous()
    scratch = prepared if prepared is not None else torch.empty(12288, dtype=torch.float32, device=obs.device)
    strides = (C.c_int64 * 4)(*weight.stride())
    stream = C.c_void_p(torch.cuda.current_stream(obs.device).cuda_stream)
    # with_mask (the update's forward): also the layer's ReLU mask, int32 [n, 225], bit c of word (image, position) = channel c is active
    mask = torch.empty((n, 225), dtype=torch.int32, device=obs.device) if with_mask else None
    _chk(lib().grip_conv1_u8_train(C.c_void_p(obs.data_ptr()), None if rows is None else C.c_void_p(rows.row0.data_ptr()),
                                   None if idx is None else C.c_void_p(idx.index.data_ptr()), n, 5, None if prepared is not None else C.c_void_p(weight.data_ptr()), strides,
                                   C.c_void_p(bias.data_ptr()), C.c_void_p(scratch.data_ptr()), C.c_void_p(out.data_ptr()), C.c_void_p(other.data_ptr()),
                                   None if mask is None else C.c_void_p(mask.data_ptr()), stream))
    return (out, other, mask) if with_mask else (out, other)


CONV23_B2_ROWS, CONV23_B3_ROWS = 1024 + 768 + 768, 1152 + 864 + 864     # fp32 operand in both layouts + the bf16 fragments of the forward + those of the data gradients (k_conv23_prep)


def conv23_prep(w2, w3, b2_mat=None, b3_mat=None):
    """The weights of AugmentedNatureCNN's second and third convolutions as the GEMM operands grip_conv23 / grip_trunk_backward read: float32 [CONV23_B2_ROWS, 64]
    and [CONV23_B3_ROWS, 64] -- the matrix k-major, then channel-major (2 x 512 / 2 x 576 rows), then the bf16 kernel's operand fragments (three bf16 terms per weight:
    196 608 / 221 184 bytes = 768 / 864 rows); pass the previous pair to rewrite it in place (fixed addresses for a captured rollout tick)."""
    import torch
    assert w2.is_cuda and w2.dtype == torch.float32 and tuple(w2.shape) == (64, 32, 4, 4) and w3.dtype == torch.float32 and tuple(w3.shape) == (64, 64, 3, 3)
    if b2_mat is None:
        b2_mat = torch.empty((CONV23_B2_ROWS, 64), dtype=torch.float32, device=w2.device); b3_mat = torch.empty((CONV23_B3_ROWS, 64), dtype=torch.float32, device=w2.device)
    s2 = (C.c_int64 * 4)(*w2.stride()); s3 = (C.c_int64 * 4)(*w3.stride())
    stream = C.c_void_p(torch.cuda.current_stream(w2.device).cuda_stream)
    _chk(lib().grip_conv23_prep(C.c_void_p(w2.data_ptr()), s2, C.c_void_p(w3.data_ptr()), s3, C.c_void_p(b2_mat.data_ptr()), C.c_void_p(b3_mat.data_ptr()), stream))
    return b2_mat, b3_mat


def conv23(y1, b2_mat, bias2, b3_mat, bias3, train=False):
    """relu(conv3(relu(conv2(y1)))) of AugmentedNatureCNN for rollouts in one launch (grip_conv23, csrc/grip_policy.hip): y1 = conv1_u8's
    channels-last float32 [n, 32, 15, 15] -> channels-last float32 [n, 64, 4, 4]. No autograd."""
    import torch
    n = int(y1.shape[0])
    assert y1.is_cuda and y1.dtype == torch.float32 and tuple(y1.shape[1:]) == (32, 15, 15) and y1.is_contiguous(memory_format=torch.channels_last)
    assert tuple(b2_mat.shape) == (CONV23_B2_ROWS, 64) and tuple(b3_mat.shape) == (CONV23_B3_ROWS, 64) and bias2.is_contiguous() and bias3.is_contiguous()
    out = torch.empty((n, 64, 4, 4), dtype=torch.float32, device=y1.device, memory_format=torch.channels_last)
    stream = C.c_void_p(torch.cuda.current_stream(y1.device).cuda_stream)
    # train (the update's forward): also y2 (channels-last [n, 64, 6, 6]) and the two layers' ReLU masks, int64 [n, 36] / [n, 16], bit c = channel c is active
    y2 = torch.empty((n, 64, 6, 6), dtype=torch.float32, device=y1.device, memory_format=torch.channels_last) if train else None
    m2 = torch.empty((n, 36), dtype=torch.int64, device=y1.device) if train else None
    m3 = torch.empty((n, 16), dtype=torch.int64, device=y1.device) if train else None
    vp = lambda t: None if t is None else C.c_void_p(t.data_ptr())
    _chk(lib().grip_conv23_train(C.c_void_p(y1.data_ptr()), n, C.c_void_p(b2_mat.data_ptr()), C.c_void_p(bias2.data_ptr()), C.c_void_p(b3_mat.data_ptr()),
                                 C.c_void_p(bias3.data_ptr()), C.c_void_p(out.data_ptr()), vp(y2), vp(m2), vp(m3), stream))
    return (out, y2, m2, m3) if train else out


def _nhwc(t, c, hw):
    import torch
    return t.is_cuda and t.dtype == torch.float32 and tuple(t.shape[1:]) == (c, hw, hw) and t.is_contiguous(memory_format=torch.channels_last)


def trunk_backward(g3, mask3, mask2, mask1, obs, b3_mat, b2_mat, w1=None, want_g1m=False, gw_out=None, gb_out=None):
    """Backward of AugmentedNatureCNN's convolutions below the third layer's output, one launch (grip_trunk_backward, csrc/grip_train.hip):
    g3 = d loss / d y3 (channels-last float32 [n, 64, 4, 4]), the ReLU masks of the training forward (conv23(train=True): mask3 int64 [n, 16], mask2
    int64 [n, 36]; conv1_u8(with_mask=True): mask1 int32 [n, 225]), the weight matrices of conv23_prep, the uint8 observations [n, 5, 64, 64] (None:
    no first-layer weight gradient) and the first layer's weight (for the gradient's shape and strides) -> (g3 * mask3, d loss / d conv2's
    pre-activation, d loss / d w1, (d loss / d b1, d loss / d b2, d loss / d b3), d loss / d conv1's pre-activation or None), the data gradients
    channels-last. gw_out / gb_out: write d loss / d w1 (strides of w1) and the three bias gradients there instead of into fresh tensors."""
    import torch
    n = int(g3.shape[0])
    assert _nhwc(g3, 64, 4)
    for m, shape, dt in ((mask3, (n, 16), torch.int64), (mask2, (n, 36), torch.int64), (mask1, (n, 225), torch.int32)):
        assert m.is_cuda and m.dtype == dt and tuple(m.shape) == shape and m.is_contiguous()
    assert tuple(b2_mat.shape) == (CONV23_B2_ROWS, 64) and tuple(b3_mat.shape) == (CONV23_B3_ROWS, 64) and b2_mat.is_contiguous() and b3_mat.is_contiguous()
    g3m = torch.empty_like(g3)
    g2m = torch.empty((n, 64, 6, 6), dtype=torch.float32, device=g3.device, memory_format=torch.channels_last)
    g1m = torch.empty((n, 32, 15, 15), dtype=torch.float32, device=g3.device, memory_format=torch.channels_last) if want_g1m or obs is None else None
    gw = gb = part = None
    vp = lambda t: None if t is None else C.c_void_p(t.data_ptr())
    strides = None
    obs_rows = None
    if isinstance(obs, IndexedRows):
        assert obs.n == n
        obs_rows, obs = obs.index, obs.records
    if obs is not None:
        assert obs.is_cuda and obs.dtype == torch.uint8 and obs.is_contiguous() and tuple(obs.shape[1:]) == (5, 64, 64) and (obs_rows is not None or obs.shape[0] == n)
        assert w1 is not None and tuple(w1.shape) == (32, 4, 8, 8) and w1.dtype == torch.float32
        gw = torch.empty_like(w1) if gw_out is None else gw_out
        gb = tuple(torch.empty(k, dtype=torch.float32, device=g3.device) for k in (32, 64, 64)) if gb_out is None else tuple(gb_out)
        assert gw.stride() == w1.stride() and gw.dtype == torch.float32 and all(t.is_contiguous() and t.numel() == k and t.dtype == torch.float32 for t, k in zip(gb, (32, 64, 64)))
        part = torch.empty((int(lib().grip_trunk_backward_parts(n)), 8352), dtype=torch.float32, device=g3.device)
        strides = (C.c_int64 * 4)(*gw.stride())
    stream = C.c_void_p(torch.cuda.current_stream(g3.device).cuda_stream)
    _chk(lib().grip_trunk_backward(vp(g3), vp(mask3), vp(mask2), vp(mask1), vp(obs), vp(obs_rows), 5, vp(b3_mat), vp(b2_mat), n, vp(g3m), vp(g2m), vp(g1m), vp(part), vp(gw), strides,
                                   *([vp(t) for t in gb] if gb else [None] * 3), stream))
    return g3m, g2m, gw, gb, g1m


def conv23_weight_gradients(y1, g2m, y2, g3m, gw2_out=None, gw3_out=None):
    """d loss / d w2 [64, 32, 4, 4] and d loss / d w3 [64, 64, 3, 3] of AugmentedNatureCNN's second and third convolution (grip_wgrad23, csrc/grip_train.hip:
    k_wgrad23_b3 on the bf16 matrix pipe in fp32-equivalent arithmetic) from the layers' inputs y1 [n, 32, 15, 15], y2 [n, 64, 6, 6] and the gradients at their
    pre-activations g2m [n, 64, 6, 6], g3m [n, 64, 4, 4] (trunk_backward) -- all channels-last float32. gw?_out: write there (any strides) instead of into fresh tensors."""
    import torch
    n = int(y1.shape[0])
    assert _nhwc(y1, 32, 15) and _nhwc(g2m, 64, 6) and _nhwc(y2, 64, 6) and _nhwc(g3m, 64, 4) and all(int(t.shape[0]) == n for t in (g2m, y2, g3m))
    gw2 = torch.empty((64, 32, 4, 4), dtype=torch.float32, device=y1.device) if gw2_out is None else gw2_out
    gw3 = torch.empty((64, 64, 3, 3), dtype=torch.float32, device=y1.device) if gw3_out is None else gw3_out
    assert tuple(gw2.shape) == (64, 32, 4, 4) and tuple(gw3.shape) == (64, 64, 3, 3) and gw2.dtype == torch.float32 and gw3.dtype == torch.float32 and gw2.is_cuda and gw3.is_cuda
    words = int(lib().grip_wgrad23_scratch_floats(n))
    assert words > 0, "grip_wgrad23_scratch_floats: no device"
    scratch = torch.empty(words, dtype=torch.float32, device=y1.device)
    vp = lambda t: C.c_void_p(t.data_ptr())
    stream = C.c_void_p(torch.cuda.current_stream(y1.device).cuda_stream)
    _chk(lib().grip_wgrad23(vp(y1), vp(g2m), vp(y2), vp(g3m), n, vp(scratch), vp(gw2), (C.c_int64 * 4)(*gw2.stride()), vp(gw3), (C.c_int64 * 4)(*gw3.stride()), stream))
    return gw2, gw3


def tanh_backward_colsum(g, h, batch_major_to_rows=False, gb_out=None):
    """gz = g * (1 - h^2) and its column sums (the tanh layer's bias gradient) in one pass (grip_tanh_backward_colsum, csrc/grip_train.hip).
    g float32 [B, n, C] contiguous (B = 1 for a 2-D g). batch_major_to_rows=False: h is [B, n, C] like g, gz comes back in that layout. True: h is the
    row-major [n, B * C] activation whose column blocks are the batch entries, and gz comes back row-major [n, B * C]. Returns (gz, grad_bias [B * C])."""
    import torch
    g3 = g if g.dim() == 3 else g.unsqueeze(0)
    B, n, Cc = (int(x) for x in g3.shape)
    assert g3.is_cuda and g3.dtype == torch.float32 and g3.is_contiguous() and h.dtype == torch.float32 and h.is_contiguous()
    if batch_major_to_rows:
        assert tuple(h.shape) == (n, B * Cc)
        gz = torch.empty_like(h); row_stride, bstride = B * Cc, Cc
    else:
        assert h.numel() == g3.numel()
        gz = torch.empty_like(g); row_stride, bstride = Cc, n * Cc
    gb = torch.empty(B * Cc, dtype=torch.float32, device=g.device) if gb_out is None else gb_out
    assert gb.is_contiguous() and gb.numel() == B * Cc and gb.dtype == torch.float32
    sc = torch.empty(((n + 31) // 32) * B * Cc, dtype=torch.float32, device=g.device)
    stream = C.c_void_p(torch.cuda.current_stream(g.device).cuda_stream)
    _chk(lib().grip_tanh_backward_colsum(C.c_void_p(g3.data_ptr()), C.c_void_p(h.data_ptr()), C.c_void_p(gz.data_ptr()), B, n, Cc, row_stride, bstride,
                                         C.c_void_p(sc.data_ptr()), C.c_void_p(gb.data_ptr()), stream))
    return gz, gb


def relu_backward_colsum(g, h, gb_out=None):
    """gz = g * (h > 0) and its column sums (the ReLU layer's bias gradient) in one pass (grip_relu_backward_colsum): g float32 [n, C], rows possibly strided
    (a column slice of a wider gradient), h float32 [n, C] contiguous. Returns (gz [n, C], grad_bias [C])."""
    import torch
    n, Cc = (int(x) for x in h.shape)
    assert g.is_cuda and g.dtype == torch.float32 and tuple(g.shape) == (n, Cc) and g.stride(1) == 1 and g.stride(0) >= Cc and h.dtype == torch.float32 and h.is_contiguous()
    gz = torch.empty_like(h); gb = torch.empty(Cc, dtype=torch.float32, device=g.device) if gb_out is None else gb_out
    assert gb.is_contiguous() and gb.numel() == Cc and gb.dtype == torch.float32
    sc = torch.empty(((n + 31) // 32) * Cc, dtype=torch.float32, device=g.device)
    stream = C.c_void_p(torch.cuda.current_stream(g.device).cuda_stream)
    _chk(lib().grip_relu_backward_colsum(C.c_void_p(g.data_ptr()), int(g.stride(0)), C.c_void_p(h.data_ptr()), C.c_void_p(gz.data_ptr()), n, Cc, C.c_void_p(sc.data_ptr()),
                                         C.c_void_p(gb.data_ptr()), stream))
    return gz, gb


class ClipAdam:
    """clip_grad_norm_(parameters, max_norm) followed by optimizer.step() of a torch.optim.Adam, in two launches (grip_clip_adam, csrc/grip_train.hip):
    works on the optimiser's own state tensors (step, exp_avg, exp_avg_sq: its state_dict stays what torch would have written) and hyper-parameters.
    step() returns False -- nothing done -- when the optimiser or its tensors are not of the kind the kernel handles (the caller then runs the
    tensor library's pair)."""

    def __init__(self, optimizer, max_norm):
        self.opt, self.max_norm, self._partials = optimizer, float(max_norm), None

    def step(self):
        import torch
        if type(self.opt) is not torch.optim.Adam or len(self.opt.param_groups) != 1:
            return False
        g = self.opt.param_groups[0]
        ps = [p for p in g["params"] if p.grad is not None]
        if (not ps or len(ps) > 48 or g.get("weight_decay", 0) != 0 or g.get("amsgrad") or g.get("maximize") or g.get("differentiable")
                or not isinstance(g["lr"], float) or not (g.get("capturable") or g.get("fused"))):
            return False
        for p in ps:
            dense = p.is_contiguous() or (p.dim() == 4 and p.is_contiguous(memory_format=torch.channels_last))
            if not (p.is_cuda and p.dtype == torch.float32 and dense and p.grad.dtype == torch.float32 and p.grad.stride() == p.stride() and not p.grad.is_sparse):
                return False
        for p in ps:                                    # state as torch.optim.Adam._init_group creates it for capturable / fused optimisers
            st = self.opt.state[p]
            if len(st) == 0:
                st["step"] = torch.zeros((), dtype=torch.float32, device=p.device)
                st["exp_avg"] = torch.zeros_like(p, memory_format=torch.preserve_format)
                st["exp_avg_sq"] = torch.zeros_like(p, memory_format=torch.preserve_format)
            if not (st["step"].is_cuda and st["step"].dtype == torch.float32 and st["exp_avg"].stride() == p.stride() and st["exp_avg_sq"].stride() == p.stride()):
                return False
        n = len(ps)
        numel = (C.c_int64 * n)(*[p.numel() for p in ps])
        arr = lambda ts: (C.c_void_p * n)(*[t.data_ptr() for t in ts])
        nchunks = int(lib().grip_clip_adam_chunks(n, numel))
        if self._partials is None or self._partials.numel() < nchunks or self._partials.device != ps[0].device:
            self._partials = torch.empty(nchunks, dtype=torch.float32, device=ps[0].device)
        b1, b2 = g["betas"]
        stream = C.c_void_p(torch.cuda.current_stream(ps[0].device).cuda_stream)
        _chk(lib().grip_clip_adam(n, numel, arr(ps), arr([p.grad for p in ps]), arr([self.opt.state[p]["exp_avg"] for p in ps]),
                                  arr([self.opt.state[p]["exp_avg_sq"] for p in ps]), arr([self.opt.state[p]["step"] for p in ps]), float(g["lr"]), float(b1), float(b2),
                                  float(g["eps"]), self.max_norm, C.c_void_p(self._partials.data_ptr()), None, stream))
        self.bump_versions(ps)
        return True

    @staticmethod
    def bump_versions(params):
        """The kernel writes the parameters through raw pointers: tell autograd's version counters (host side, legal during a stream capture), so
        that whoever caches something derived from the parameters -- the policy's merged rollout weights -- sees that they changed. A replayed
        graph does not come through here: PPO.train() calls this once more after its last replay."""
        import torch
        for p in params:
            torch.autograd.graph.increment_version(p)


def ppo_loss(mean, log_std, values, actions, old_log_prob, advantages, returns, clip_range, ent_coef, vf_coef):
    """PPO loss of one minibatch and its gradients in one launch (grip_ppo_loss, csrc/grip_policy.hip): float32 CUDA tensors in,
    (out[3] = loss / policy loss / value loss, d loss / d mean [n, A], d loss / d values [n], d loss / d log_std [A]) out."""
    import torch
    ts = [t.float().contiguous() for t in (mean, log_std, values, actions, old_log_prob, advantages, returns)]
    n, A = int(ts[0].shape[0]), int(ts[0].shape[1])
    assert all(t.is_cuda for t in ts) and ts[3].shape == (n, A) and ts[1].shape == (A,) and all(t.shape == (n,) for t in (ts[2], ts[4], ts[5], ts[6]))
    dev = ts[0].device
    out = torch.empty(3, dtype=torch.float32, device=dev); gm = torch.empty((n, A), dtype=torch.float32, device=dev)
    gv = torch.empty(n, dtype=torch.float32, device=dev); gl = torch.empty(A, dtype=torch.float32, device=dev)
    stream = C.c_void_p(torch.cuda.current_stream(dev).cuda_stream)
    _chk(lib().grip_ppo_loss(*[C.c_void_p(t.data_ptr()) for t in ts], n, A, float(clip_range), float(ent_coef), float(vf_coef),
                             C.c_void_p(out.data_ptr()), C.c_void_p(gm.data_ptr()), C.c_void_p(gv.data_ptr()), C.c_void_p(gl.data_ptr()), stream))
    return out, gm, gv, gl


def bias_tanh_(z, bias):
    """z = tanh(z + bias) in place (grip_bias_tanh): z float32 [B, n, C] contiguous, bias [B * C]; returns z"""
    import torch
    B, n, Cc = (int(x) for x in z.shape)
    assert z.is_cuda and z.dtype == torch.float32 and z.is_contiguous() and bias.dtype == torch.float32 and bias.is_contiguous() and bias.numel() == B * Cc and Cc % 4 == 0
    _chk(lib().grip_bias_tanh(C.c_void_p(z.data_ptr()), C.c_void_p(bias.data_ptr()), B, n, Cc, C.c_void_p(torch.cuda.current_stream(z.device).cuda_stream)))
    return z


def ppo_loss_heads(heads_out, head_bias, log_std, actions, old_log_prob, advantages, returns, rows, clip_range, ent_coef, vf_coef, grad_head_bias, grad_log_std):
    """The minibatch loss for the update's explicit launch sequence (grip_ppo_loss_heads, two launches): heads_out float32 [2, n, 8] (the merged heads' last
    batch-of-two GEMM WITHOUT its bias, head_bias [2, 8] beside it: [0, i, :A] + bias = mean, [1, i, 0] + bias = value), the ROLLOUT's sample arrays (actions [R, A], old_log_prob / advantages / returns [R]) and the
    minibatch's rows of them (int64 [n]) -> (out[3] = loss / policy loss / value loss, d loss / d heads_out [2, n, 8]); the heads' bias gradients [2, 8] and
    d loss / d log_std [A] are written into grad_head_bias / grad_log_std."""
    import torch
    n, A = int(heads_out.shape[1]), int(log_std.numel())
    f32 = lambda t, shape: t.is_cuda and t.dtype == torch.float32 and t.is_contiguous() and (shape is None or tuple(t.shape) == shape)
    R = int(actions.shape[0])
    assert f32(heads_out, (2, n, 8)) and f32(head_bias, None) and head_bias.numel() == 16 and f32(log_std, (A,)) and f32(actions, (R, A)) and all(f32(t, None) and t.numel() == R for t in (old_log_prob, advantages, returns))
    assert rows.is_cuda and rows.dtype == torch.int64 and rows.is_contiguous() and rows.numel() == n
    assert f32(grad_head_bias, None) and grad_head_bias.numel() == 16 and f32(grad_log_std, None) and grad_log_std.numel() >= A
    dev = heads_out.device
    out = torch.empty(3, dtype=torch.float32, device=dev); go = torch.empty_like(heads_out)
    samples = torch.empty(n * (A + 3) + 324, dtype=torch.float32, device=dev)     # the gathered samples, the loss kernel's 16 x 20 partial sums, its arrival counter (one word, per call)
    stream = C.c_void_p(torch.cuda.current_stream(dev).cuda_stream)
    p = lambda t: C.c_void_p(t.data_ptr())
    _chk(lib().grip_ppo_loss_heads(p(heads_out), p(head_bias), p(log_std), p(actions), p(old_log_prob), p(advantages), p(returns), p(rows), n, A, float(clip_range), float(ent_coef),
                                   float(vf_coef), p(samples), p(out), p(go), p(grad_head_bias), p(grad_log_std), stream))
    return out, go


class Model:
    def __init__(self, obj_or_path):
        path = obj_or_path if os.path.isfile(obj_or_path) else asset_path(obj_or_path)
        self.ptr = C.c_void_p()
        _chk(lib().grip_model_load(path.encode(), C.byref(self.ptr)))
        self.path = path

    def __del__(self):
        if getattr(self, "ptr", None) and _lib is not None:
            _lib.grip_model_free(self.ptr)
            self.ptr = None


def _np(a, dt):
    a = np.ascontiguousarray(a, dtype=dt)
    return a, a.ctypes.data_as(C.c_void_p)


class Batch:
    """N environments of one model on one GPU (GripBatch). All tensors are torch CUDA tensors."""

    def __init__(self, model, n_envs, device_index=0, out=None, **cfg):
        """`out`: optional dict of preallocated result tensors (one per _OUT_FIELDS entry, first dimension n_envs, contiguous)
        to write into -- MixedBatch hands every group a slice of one shared set."""
        import torch
        if not torch.cuda.is_available():
            raise GripError("no GPU visible: the rollout engine has no CPU path")
        self.torch = torch
        self.model = model if isinstance(model, Model) else Model(model)
        self.n = int(n_envs)
        self.device = torch.device("cuda", device_index)
        self.ptr = C.c_void_p()
        _chk(lib().grip_batch_create(self.model.ptr, self.n, device_index, C.byref(self.ptr)))
        self.cfg = EnvConfigC(400, 400, 1, 1, 0, 0, 0.05, 0.15, 0.002, 0.03, (C.c_float * 2)(1.0, 0.0))
        self.out = {}
        for name, dt, shp in _OUT_FIELDS:
            if out is not None:
                v = out[name]
                if tuple(v.shape) != (self.n,) + shp or v.dtype != getattr(torch, dt) or not v.is_contiguous() or v.device != self.device:
                    raise GripError(f"result tensor '{name}' must be contiguous {dt} {(self.n,) + shp} on {self.device}")
                self.out[name] = v
            else:
                self.out[name] = torch.zeros((self.n,) + shp, dtype=getattr(torch, dt), device=self.device)
        self._outc = StepOutC(**{n: self.out[n].data_ptr() for n, _, _ in _OUT_FIELDS})
        if cfg:
            self.set_config(**cfg)

    # -- config ---------------------------------------------------------------------------------
    def set_config(self, **kw):
        for k, v in kw.items():
            if k == "target_dir":
                self.cfg.target_dir[0], self.cfg.target_dir[1] = float(v[0]), float(v[1])
            elif hasattr(self.cfg, k):
                setattr(self.cfg, k, v)
            else:
                raise GripError(f"unknown config field {k}")
        _chk(lib().grip_batch_set_config(self.ptr, C.byref(self.cfg)))

    @property
    def action_dim(self):
        return 6 if self.cfg.include_roll else 5

    @property
    def obs_channels(self):
        return 5 if self.cfg.full_observation else 4

    def _stream(self):
        return C.c_void_p(self.torch.cuda.current_stream(self.device).cuda_stream)

    # -- env surface ----------------------------------------------------------------------------
    def reset(self, mask=None):
        mp = None
        if mask is not None:
            mask = mask.to(device=self.device, dtype=self.torch.uint8).contiguous()
            mp = C.c_void_p(mask.data_ptr())
        _chk(lib().grip_batch_reset(self.ptr, mp, C.byref(self._outc), self._stream()))
        return self.out

    def step(self, actions):
        a = actions.to(device=self.device, dtype=self.torch.float32).contiguous()
        if a.shape != (self.n, self.action_dim):
            raise GripError(f"actions must be [{self.n},{self.action_dim}], got {tuple(a.shape)}")
        _chk(lib().grip_batch_step(self.ptr, C.c_void_p(a.data_ptr()), C.byref(self._outc), self._stream()))
        return self.out

    def observe(self, obs=None):
        if obs is None:
            obs = self.torch.empty((self.n, self.obs_channels, 64, 64), dtype=self.torch.uint8, device=self.device)
        _chk(lib().grip_batch_observe(self.ptr, C.c_void_p(obs.data_ptr()), self._stream()))
        return obs

    def render_camera(self, env_index=0, cam_pos=None, cam_R=None, fovy=None, width=64, height=64, depth=False):
        """RobotEnv.render's camera view (grip_batch_render_camera): uint8 [H, W, 3] RGB, or float32 [H, W] metres with depth=True.
        cam_pos / cam_R (3 and 3 x 3, columns = camera axes in world coordinates, looking along -z) default to the gripper camera."""
        t = self.torch
        cam = None
        if cam_pos is not None:
            cam = t.tensor(list(np.asarray(cam_pos, np.float32).reshape(3)) + list(np.asarray(cam_R, np.float32).reshape(9)), dtype=t.float32, device=self.device)
        out = t.empty((height, width) if depth else (height, width, 3), dtype=t.float32 if depth else t.uint8, device=self.device)
        _chk(lib().grip_batch_render_camera(self.ptr, int(env_index), None if cam is None else C.c_void_p(cam.data_ptr()), float(fovy if fovy is not None else self.gripper_fovy),
                                            int(width), int(height), None if depth else C.c_void_p(out.data_ptr()), C.c_void_p(out.data_ptr()) if depth else None, self._stream()))
        return out

    @property
    def gripper_fovy(self):
        from .model import blob
        if not hasattr(self, "_fovy"):
            self._fovy = float(blob.read_blob(self.model.path)["cam_fovy"][0])
        return self._fovy

    # -- asynchronous stepping (grip_sim.h: grip_batch_advance) ------------------------------------
    def advance(self, slot_actions, slice_len, ready_list, ready_count, budget_us=0, lag=1):
        """One time slice: at most `slice_len` calls of physics.step() per env. slot_actions float32 [capacity, action_dim]
        are the actions for the envs the previous call listed; ready_list int32 [capacity] / ready_count int32 [1] receive
        the envs now waiting for an action. Finished envs' rows of self.out are refreshed."""
        t = self.torch
        cap = int(ready_list.numel())
        if slot_actions.dtype != t.float32 or not slot_actions.is_contiguous() or tuple(slot_actions.shape) != (cap, self.action_dim):
            raise GripError(f"slot_actions must be contiguous float32 [{cap},{self.action_dim}]")
        if ready_list.dtype != t.int32 or ready_count.dtype != t.int32:
            raise GripError("ready_list / ready_count must be int32")
        _chk(lib().grip_batch_advance(self.ptr, C.c_void_p(slot_actions.data_ptr()), int(slice_len), int(budget_us), int(lag), cap, C.byref(self._outc),
                                      C.c_void_p(ready_list.data_ptr()), C.c_void_p(ready_count.data_ptr()), self._stream()))
        return self.out

    def observe_list(self, ready_list, ready_count, obs, records=None, record_row=None):
        """Render the listed envs into obs rows 0..count-1; with `records` (uint8 [R, C, 64, 64]) and `record_row` (int64 [1])
        also into records[record_row + r]."""
        cap = int(ready_list.numel())
        if obs is None and records is None:
            raise GripError("observe_list needs obs rows, record rows, or both")
        if obs is not None and (obs.dtype != self.torch.uint8 or not obs.is_contiguous() or obs.shape[0] < cap):
            raise GripError("obs must be contiguous uint8 [capacity, C, 64, 64]")
        rp = None if records is None else C.c_void_p(records.data_ptr())
        rr = None if record_row is None else C.c_void_p(record_row.data_ptr())
        _chk(lib().grip_batch_observe_list(self.ptr, C.c_void_p(ready_list.data_ptr()), C.c_void_p(ready_count.data_ptr()), cap,
                                           None if obs is None else C.c_void_p(obs.data_ptr()), rp, rr, self._stream()))
        return obs

    def add_intrinsic_reward(self, old_obs, new_obs, reward, old_rows=None, ready_list=None, ready_count=None):
        """reward[e] += IntrinsicReward.intrinsic_reward(old, new) (reward.py:57-77) for every pair; see grip_sim.h."""
        t = self.torch
        assert old_obs.dtype == t.uint8 and new_obs.dtype == t.uint8 and reward.dtype == t.float32
        n_pairs = int(new_obs.shape[0]) if ready_list is None else int(ready_list.numel())
        p = lambda x: None if x is None else C.c_void_p(x.data_ptr())
        rc = lib().grip_intrinsic_reward(p(old_obs), p(old_rows), p(new_obs), p(ready_list), p(ready_count), n_pairs, int(new_obs.shape[1]),
                                         int(self.cfg.full_observation), p(reward), self._stream())
        if rc != 0:
            raise GripError("grip_intrinsic_reward failed")

    # -- low-level hooks ------------------------------------------------------------------------
    def get_state(self):
        qpos = np.zeros((self.n, 14), np.float32); qvel = np.zeros((self.n, 13), np.float32)
        ctrl = np.zeros((self.n, 7), np.float32); warm = np.zeros((self.n, 13), np.float32)
        _chk(lib().grip_batch_get_state(self.ptr, qpos.ctypes.data, qvel.ctypes.data, ctrl.ctypes.data, warm.ctypes.data, 0, self._stream()))
        return qpos, qvel, ctrl, warm

    def set_state(self, qpos=None, qvel=None, ctrl=None, warm=None):
        keep = []
        ptrs = []
        for a, w in ((qpos, 14), (qvel, 13), (ctrl, 7), (warm, 13)):
            if a is None:
                ptrs.append(None)
            else:
                arr, p = _np(a, np.float32)
                assert arr.shape == (self.n, w), (arr.shape, w)
                keep.append(arr); ptrs.append(p)
        _chk(lib().grip_batch_set_state(self.ptr, *ptrs, 0, self._stream()))

    def set_state_storage(self, dtype):
        """'f16': keep qpos / qvel / ctrl as IEEE half in HBM (BASELINE.json configs[4]; fp32 arithmetic, rounded on every
        state store); 'f32': the default."""
        if dtype not in ("f16", "f32"):
            raise GripError("state storage is 'f32' or 'f16'")
        _chk(lib().grip_batch_set_state_storage(self.ptr, int(dtype == "f16"), self._stream()))

    def get_flags(self):
        es = np.zeros(self.n, np.int32); st = np.zeros(self.n, np.int32); go = np.zeros(self.n, np.int32)
        _chk(lib().grip_batch_get_flags(self.ptr, es.ctypes.data, st.ctypes.data, go.ctypes.data, self._stream()))
        return es, st, go

    def set_flags(self, episode_step=None, status=None, gripper_open=None):
        keep, ptrs = [], []
        for a in (episode_step, status, gripper_open):
            if a is None:
                ptrs.append(None)
            else:
                arr, p = _np(a, np.int32); keep.append(arr); ptrs.append(p)
        _chk(lib().grip_batch_set_flags(self.ptr, *ptrs, self._stream()))

    def substep(self, k=1):
        _chk(lib().grip_batch_substep(self.ptr, int(k), self._stream()))

    def debug_forward(self):
        n = self.n
        ncon = np.zeros(n, np.int32); con = np.zeros((n, MAXCON, 10), np.float32); xpos = np.zeros((n, 8, 3), np.float32)
        qacc = np.zeros((n, 13), np.float32); qs = np.zeros((n, 13), np.float32); M = np.zeros((n, 13, 13), np.float32)
        bias = np.zeros((n, 13), np.float32)
        _chk(lib().grip_batch_debug_forward(self.ptr, ncon.ctypes.data, con.ctypes.data, xpos.ctypes.data, qacc.ctypes.data,
                                            qs.ctypes.data, M.ctypes.data, bias.ctypes.data, self._stream()))
        return dict(ncon=ncon, con=con, xpos=xpos, qacc=qacc, qacc_smooth=qs, M=M, bias=bias)

    def target_pose(self, actions):
        a = actions.to(device=self.device, dtype=self.torch.float32).contiguous()
        t = np.zeros((self.n, 5), np.float32)
        _chk(lib().grip_batch_target_pose(self.ptr, C.c_void_p(a.data_ptr()), t.ctypes.data, self._stream()))
        return t

    def kernel_time(self, reset=True):
        ms = C.c_float(); n = C.c_int()
        _chk(lib().grip_batch_kernel_time(self.ptr, int(reset), C.byref(ms), C.byref(n)))
        return ms.value, n.value

    def device_time(self, reset=True):
        """(mean duration in ms, number) of ALL time-slice launches since the last reset, replayed graphs' included, from the device's own
        clock stamps (grip_batch_device_time); kernel_time() is the host-event figure of the eager launches only."""
        ms = C.c_double(); n = C.c_longlong()
        _chk(lib().grip_batch_device_time(self.ptr, int(reset), C.byref(ms), C.byref(n), self._stream()))
        return ms.value, n.value

    def close(self):
        if getattr(self, "ptr", None) and _lib is not None:
            _lib.grip_batch_destroy(self.ptr)
            self.ptr = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


class MixedBatch:
    """Several groups of environments on one GPU, each group with its own object model and target direction, behind the
    surface of one Batch (BASELINE.json configs[3]: {acorn, sand_ball, sugar_cube, bread_crumb} x direction {0, 45}).

    Envs are sorted by group, so every wavefront is homogeneous: group g owns the contiguous env range
    [offsets[g], offsets[g + 1]). Each group is a GripBatch of its own that writes straight into its slice of the shared
    result / observation tensors; the groups form a GripBatchSet (include/grip_sim.h), i.e. ONE launch per phase covers all
    of them -- the per-group kernel arguments sit in constant memory and a workgroup picks its group from its index -- so the
    chip is filled by one grid on one stream. In the time-sliced schedule every group keeps its own decision list and owns a
    fixed segment of the caller's ready list / action / observation rows (advance, observe_list below).
    """

    def __init__(self, groups, device_index=0, **cfg):
        """groups: sequence of (object name or .grpm path, n_envs, target_dir (x, y))."""
        import torch
        if not torch.cuda.is_available():
            raise GripError("no GPU visible: the rollout engine has no CPU path")
        if not groups:
            raise GripError("MixedBatch needs at least one group")
        self.torch = torch
        self.device = torch.device("cuda", device_index)
        self.groups = [(g[0], int(g[1]), (float(g[2][0]), float(g[2][1]))) for g in groups]
        self.offsets = [0]
        for _, n, _ in self.groups:
            if n <= 0:
                raise GripError("every group needs at least one env")
            self.offsets.append(self.offsets[-1] + n)
        self.n = self.offsets[-1]
        self.out = {name: torch.zeros((self.n,) + shp, dtype=getattr(torch, dt), device=self.device) for name, dt, shp in _OUT_FIELDS}
        self._outc = StepOutC(**{n: self.out[n].data_ptr() for n, _, _ in _OUT_FIELDS})
        cfg.pop("target_dir", None)
        models = {}
        self.parts = []
        for g, (obj, n, d) in enumerate(self.groups):
            lo, hi = self.offsets[g], self.offsets[g + 1]
            model = models.setdefault(obj, Model(obj))
            self.parts.append(Batch(model, n, device_index, out={k: v[lo:hi] for k, v in self.out.items()}, target_dir=d, **cfg))
        self.cfg = self.parts[0].cfg                       # the flags every group shares (action / observation layout)
        self.target_dirs = torch.tensor([d for _, n, d in self.groups for _ in range(n)], dtype=torch.float32, device=self.device)
        self.ptr = C.c_void_p()
        arr = (C.c_void_p * len(self.parts))(*[p.ptr for p in self.parts])
        _chk(lib().grip_batchset_create(arr, len(self.parts), C.byref(self._outc), C.byref(self.ptr)))

    action_dim = Batch.action_dim
    obs_channels = Batch.obs_channels
    _stream = Batch._stream

    def set_config(self, **kw):
        if "target_dir" in kw:
            raise GripError("target directions of a MixedBatch are fixed per group")
        for p in self.parts:
            p.set_config(**kw)
        _chk(lib().grip_batchset_refresh(self.ptr))

    def set_state_storage(self, dtype):
        for p in self.parts:
            p.set_state_storage(dtype)
        _chk(lib().grip_batchset_refresh(self.ptr))

    def part_of(self, env_index):
        """(part, index inside it) of global env `env_index` (envs are sorted by group)."""
        e = int(env_index)
        if not 0 <= e < self.n:
            raise GripError(f"env index {e} out of range [0, {self.n})")
        g = max(i for i, o in enumerate(self.offsets[:-1]) if o <= e)
        return self.parts[g], e - self.offsets[g]

    def render_camera(self, env_index=0, *args, **kw):
        """RobotEnv.render's camera view of one env of the set: forwarded to the env's own batch (Batch.render_camera)."""
        part, local = self.part_of(env_index)
        return part.render_camera(local, *args, **kw)

    def reset(self, mask=None):
        if mask is not None:
            mask = mask.to(device=self.device, dtype=self.torch.uint8).contiguous()
            if mask.numel() != self.n:
                raise GripError(f"mask must have {self.n} entries")
        for g, p in enumerate(self.parts):
            p.reset(None if mask is None else mask[self.offsets[g]:self.offsets[g + 1]])
        return self.out

    def step(self, actions):
        a = actions.to(device=self.device, dtype=self.torch.float32).contiguous()
        if a.shape != (self.n, self.action_dim):
            raise GripError(f"actions must be [{self.n},{self.action_dim}], got {tuple(a.shape)}")
        _chk(lib().grip_batchset_step(self.ptr, C.c_void_p(a.data_ptr()), self._stream()))
        return self.out

    def observe(self, obs=None):
        if obs is None:
            obs = self.torch.empty((self.n, self.obs_channels, 64, 64), dtype=self.torch.uint8, device=self.device)
        if obs.dtype != self.torch.uint8 or not obs.is_contiguous() or obs.shape[0] != self.n:
            raise GripError(f"obs must be contiguous uint8 [{self.n}, C, 64, 64]")
        _chk(lib().grip_batchset_observe(self.ptr, C.c_void_p(obs.data_ptr()), self._stream()))
        return obs

    # -- asynchronous stepping: every group keeps its own decision list; the caller sees them side by side ------------------
    def advance(self, slot_actions, slice_len, ready_list, ready_count, budget_us=0, lag=1):
        """Batch.advance for every group in one launch. Group g owns rows [g * cap / G, (g + 1) * cap / G) of slot_actions and
        ready_list: row r of its segment is the action of the env the previous call listed there. ready_list receives global
        env ids, -1 behind the listed envs of every segment (holes); ready_count the capacity (validity = sign of the entry)."""
        t = self.torch
        cap = int(ready_list.numel())
        if cap % len(self.parts):
            raise GripError(f"the ready-list capacity ({cap}) must be a multiple of the number of groups ({len(self.parts)})")
        if slot_actions.dtype != t.float32 or not slot_actions.is_contiguous() or tuple(slot_actions.shape) != (cap, self.action_dim):
            raise GripError(f"slot_actions must be contiguous float32 [{cap},{self.action_dim}]")
        if ready_list.dtype != t.int32 or ready_count.dtype != t.int32:
            raise GripError("ready_list / ready_count must be int32")
        _chk(lib().grip_batchset_advance(self.ptr, C.c_void_p(slot_actions.data_ptr()), int(slice_len), int(budget_us), int(lag), cap,
                                         C.c_void_p(ready_list.data_ptr()), C.c_void_p(ready_count.data_ptr()), self._stream()))
        return self.out

    def observe_list(self, ready_list, ready_count, obs, records=None, record_row=None):
        """Row r of obs (and of records from row record_row on) = observation of env ready_list[r]; holes are skipped."""
        cap = int(ready_list.numel())
        if obs is None and records is None:
            raise GripError("observe_list needs obs rows, record rows, or both")
        if obs is not None and (obs.dtype != self.torch.uint8 or not obs.is_contiguous() or obs.shape[0] < cap):
            raise GripError("obs must be contiguous uint8 [capacity, C, 64, 64]")
        rp = None if records is None else C.c_void_p(records.data_ptr())
        rr = None if record_row is None else C.c_void_p(record_row.data_ptr())
        _chk(lib().grip_batchset_observe_list(self.ptr, C.c_void_p(ready_list.data_ptr()), cap, None if obs is None else C.c_void_p(obs.data_ptr()), rp, rr, self._stream()))
        return obs

    def add_intrinsic_reward(self, *a, **k):
        return self.parts[0].add_intrinsic_reward(*a, **k)      # pairs of observation rows only; no group state involved

    def get_state(self):
        parts = [p.get_state() for p in self.parts]
        return tuple(np.concatenate([q[i] for q in parts]) for i in range(4))

    def set_state(self, qpos=None, qvel=None, ctrl=None, warm=None):
        for g, p in enumerate(self.parts):
            lo, hi = self.offsets[g], self.offsets[g + 1]
            p.set_state(*[None if x is None else np.asarray(x)[lo:hi] for x in (qpos, qvel, ctrl, warm)])

    def get_flags(self):
        parts = [p.get_flags() for p in self.parts]
        return tuple(np.concatenate([q[i] for q in parts]) for i in range(3))

    def kernel_time(self, reset=True):
        """(mean duration in ms of the set's macro-step launches, number of launches): timed on the first group's event ring."""
        return self.parts[0].kernel_time(reset)

    def device_time(self, reset=True):
        """the same from the device's clock stamps, every launch. Every group stamps the window of ITS workgroups (first start to last end) inside the set's one launch;
        the groups run side by side, so the LONGEST group's mean window is reported as the launch's (a lower bound of first-start-to-last-end over all groups)."""
        per = [p.device_time(reset) for p in self.parts]
        ms, n = max(per, key=lambda t: t[0])
        return ms, n

    def close(self):
        if getattr(self, "ptr", None) and _lib is not None:
            _lib.grip_batchset_destroy(self.ptr)
            self.ptr = None
        for p in getattr(self, "parts", []):
            p.close()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass
