"""Recorded GEMM kernel choices for the captured graphs (the PPO minibatch update, the rollout tick), scoped to the captures.

The update's dozen library GEMMs (fp32, [4096 x 256..1024] operands) and the rollout forward's four run ~4 % faster through the kernels a TunableOp search
picked for them on MI355X -- rocBLAS for about half of the shapes, non-default hipBLASLt solutions for most of the rest. The search's result is a committed
file (assets/tunableop_gfx950.csv, written once by `tools/update_time.py find tune`); nothing is searched at run time and the file is never written.

Round 4 switched TunableOp on for the whole process when a PPO was constructed. It is a process-global switch: every later GEMM of the user's program (SAC,
evaluation models, other libraries) then went through its table lookups too. Now it is on only INSIDE `recorded_gemm_choices()` -- the eager warm-up steps
and the stream capture of a graph, whose replays keep the kernels that were picked -- and the previous state is restored on the way out. A caller's own
TunableOp set-up (already enabled) is left alone. GRIP_TUNABLEOP=0 switches the file off; GRIP_TUNABLEOP_TUNE=<path> (maintenance) searches the shapes the file
lacks into a COPY of it at that path for the life of the process.
"""
import contextlib
import os
import warnings

import torch as th

GEMM_CHOICES = os.path.abspath(os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "assets", "tunableop_gfx950.csv"))
_state = {"checked": False, "usable": False, "grow": False}


def _file_validators(path):
    out = {}
    with open(path) as f:
        for ln in f:
            p = ln.strip().split(",")
            if len(p) >= 3 and p[0] == "Validator":
                out[p[1]] = ",".join(p[2:])
    return out


def _check_once():
    """Is the file there, wanted, and made for the running library stack? Said once (a warning) when it is not."""
    if _state["checked"]:
        return _state["usable"]
    _state["checked"] = True
    if os.environ.get("GRIP_TUNABLEOP") == "0" or not os.path.exists(GEMM_CHOICES) or not th.cuda.is_available():
        return False
    try:
        t = th.cuda.tunable
        grow = os.environ.get("GRIP_TUNABLEOP_TUNE")        # maintenance: search the shapes the file lacks into a COPY of it at this path (then copy it back by hand)
        if grow and not t.is_enabled():
            import shutil
            shutil.copyfile(GEMM_CHOICES, grow)
            t.set_max_tuning_duration(30); t.set_max_tuning_iterations(20)
            t.tuning_enable(True); t.set_filename(grow); t.enable(True)
            _state["grow"] = True
            return False                                     # (already on, process-wide, by request)
        want = _file_validators(GEMM_CHOICES)
        have = {str(k): str(v) for k, v in t.get_validators()}
        diff = {k: (v, have.get(k)) for k, v in want.items() if k in have and have[k] != v}
        if diff:
            warnings.warn("the recorded GEMM choices (assets/tunableop_gfx950.csv) were made for another library stack and TunableOp will ignore them: "
                          + "; ".join(f"{k}: file {a}, running {b}" for k, (a, b) in diff.items()) + " -- the update runs on the libraries' default kernels")
            return False
        _state["usable"] = True
    except Exception as ex:                                  # noqa: BLE001 -- an optimisation only
        warnings.warn(f"recorded GEMM choices not used ({ex})")
    return _state["usable"]


@contextlib.contextmanager
def recorded_gemm_choices():
    """TunableOp on, reading the recorded choices, for the body only (warm-up steps + capture of a graph); the previous state comes back afterwards."""
    if not _check_once():
        yield False
        return
    t = th.cuda.tunable
    if t.is_enabled():                           # a caller's own TunableOp set-up (tools/update_time.py tune) stays as it is
        yield False
        return
    prev_tuning = t.tuning_is_enabled()
    try:
        t.tuning_enable(False)                   # nothing is searched, so nothing is ever written to the file either
        t.set_filename(GEMM_CHOICES); t.enable(True)
    except Exception as ex:                      # noqa: BLE001
        warnings.warn(f"recorded GEMM choices not used ({ex})")
        yield False
        return
    try:
        yield True
    finally:
        t.enable(False)
        t.tuning_enable(prev_tuning)
