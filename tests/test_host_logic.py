"""Host-side helpers of round 5 that need no GPU: the source fingerprint that gates the counter summary (engine.source_fingerprint, bench.py), the scoping of the
recorded GEMM choices (sb3/gemm_choices.py), the operand-buffer sizes the engine and the ABI header agree on."""
import hashlib
import os
import re

from mujoco_rl_manipulate_unknown_objects_amd import engine

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_source_fingerprint_is_a_hash_of_names_and_contents():
    fp = engine.source_fingerprint()
    assert re.fullmatch(r"[0-9a-f]{16}", fp) and fp == engine.source_fingerprint()
    # recomputed by hand as its docstring says: csrc/*.hip, csrc/*.h (sorted), then include/grip_sim.h -- name, NUL, contents
    h = hashlib.sha256()
    files = sorted(os.path.join(engine.CSRC, f) for f in os.listdir(engine.CSRC) if f.endswith((".hip", ".h")))
    files.append(os.path.join(ROOT, "include", "grip_sim.h"))
    for f in files:
        h.update(os.path.basename(f).encode()); h.update(b"\0")
        with open(f, "rb") as fh:
            h.update(fh.read())
    assert h.hexdigest()[:16] == fp
    # built libraries and diagnostic variants in csrc/ do not enter it
    assert not any(f.endswith(".so") for f in files)


def test_recorded_gemm_choices_is_inert_without_a_gpu():
    import torch
    from mujoco_rl_manipulate_unknown_objects_amd.sb3 import gemm_choices
    assert os.path.exists(gemm_choices.GEMM_CHOICES)
    if torch.cuda.is_available():
        return                                              # (the GPU suite exercises the real thing: tests/test_gpu_async.py)
    with gemm_choices.recorded_gemm_choices() as on:
        assert on is False
    with gemm_choices.recorded_gemm_choices() as on:        # asked once, remembered
        assert on is False


def test_conv23_operand_buffers_match_the_header():
    """engine.CONV23_B?_ROWS (what conv23_prep allocates) against the sizes include/grip_sim.h documents for grip_conv23_prep"""
    text = open(os.path.join(ROOT, "include", "grip_sim.h")).read()
    m = re.search(r"b2_mat_dev \((\d+) x 64 floats.*?b3_mat_dev \((\d+) x 64 floats", text, re.S)
    assert m and (int(m.group(1)), int(m.group(2))) == (engine.CONV23_B2_ROWS, engine.CONV23_B3_ROWS)
    # fp32 matrix in both layouts + the forward's fragments + the data gradients' fragments (three bf16 terms = 6 bytes per weight, 256 bytes per row)
    assert engine.CONV23_B2_ROWS == 2 * 512 + 2 * (64 * 512 * 6 // 256) and engine.CONV23_B3_ROWS == 2 * 576 + 2 * (64 * 576 * 6 // 256)
