"""Extract the first episodes of the reference's own training logs into a small fixture (tests/golden/).

The reference holds no tests, but it does hold data produced by its real MuJoCo + SB3 stack: one Monitor CSV per published
run (models/trained_models/**/log_file.monitor.csv: episode return, length, wall time). The first episodes of a run are
played by a freshly initialised SAC actor (tanh-Gaussian, unit variance: close to random actions), so their lengths and
returns say how often the real simulation ends an episode early (Status.FAIL) and how far random pushing moves the object.
Only the progress-reward runs are taken (the *_im_reward_* runs add the intrinsic term to the return), and only episodes
that end before training step 2000: train_agent.py hands the SAME env to EvalCallback (eval_freq 2000), so from the first
evaluation on the log mixes evaluation episodes in (and shows "episodes" longer than the 400-step time horizon).

Run here (the reference is not on the GPU box):  python tools/make_monitor_fixture.py
"""
import csv
import glob
import json
import os

REF = "/root/reference/models/trained_models"
OUT = os.path.join(os.path.dirname(__file__), "..", "tests", "golden", "reference_monitor_early_episodes.json")
FIRST_EVAL = 2000      # config/train_config.py: --eval_freq default

runs = {}
for f in sorted(glob.glob(os.path.join(REF, "dir*_reward", "*_progress_reward_best_model", "log_file.monitor.csv"))):
    direction = int(f.split(os.sep)[-3].replace("dir", "").replace("_reward", ""))
    obj = f.split(os.sep)[-2].replace("_progress_reward_best_model", "")
    rows, steps = [], 0
    for r in list(csv.reader(open(f)))[2:]:
        steps += int(r[1])
        if steps > FIRST_EVAL:
            break
        rows.append(r)
    runs[f"{obj}_dir{direction}"] = {"source": os.path.relpath(f, "/root/reference"),
                                      "returns": [float(r[0]) for r in rows], "lengths": [int(r[1]) for r in rows]}
json.dump({"what": f"episodes (return, length) that end before training step {FIRST_EVAL} in each progress-reward training log of the reference", "runs": runs},
          open(OUT, "w"), indent=1)
print(OUT, len(runs), "runs")
