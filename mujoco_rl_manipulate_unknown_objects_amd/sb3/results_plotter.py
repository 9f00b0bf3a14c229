"""``load_results`` / ``ts2xy`` with the call shape of stable_baselines3.common.results_plotter, as used by the
reference's SaveOnBestTrainingRewardCallback (models/callbacks.py:5,68) and plotting scripts: read every
``*monitor.csv`` under a folder (format written by sb3/vec_env.py:Monitor -- ``#{json header}``, columns r,l,t)."""
import glob
import json
import os

import numpy as np
import pandas

X_TIMESTEPS, X_EPISODES, X_WALLTIME = "timesteps", "episodes", "walltime_hrs"


def load_results(path):
    files = sorted(glob.glob(os.path.join(path, "*monitor.csv")))
    if not files:
        raise FileNotFoundError(f"no monitor files of the form *monitor.csv found in {path}")
    frames = []
    for f in files:
        with open(f, "rt") as fh:
            first = fh.readline()
            assert first[0] == "#", "monitor file without its json header"
            header = json.loads(first[1:])
            df = pandas.read_csv(fh, index_col=None)
        df["t"] += header["t_start"]
        frames.append(df)
    out = pandas.concat(frames)
    out.sort_values("t", inplace=True)
    out.reset_index(inplace=True)
    out["t"] -= min(json.loads(open(f).readline()[1:])["t_start"] for f in files)
    return out


def ts2xy(data_frame, x_axis):
    if x_axis == X_TIMESTEPS:
        x, y = np.cumsum(data_frame.l.values), data_frame.r.values
    elif x_axis == X_EPISODES:
        x, y = np.arange(len(data_frame)), data_frame.r.values
    elif x_axis == X_WALLTIME:
        x, y = data_frame.t.values / 3600.0, data_frame.r.values
    else:
        raise NotImplementedError(x_axis)
    return x, y
