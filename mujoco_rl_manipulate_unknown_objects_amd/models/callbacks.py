"""Mirror of the reference's models/callbacks.py (progress bar :8-38, SaveOnBestTrainingRewardCallback :41-82) over
this package's SB3-shaped classes, so that train_agent.py:34-43 keeps working after the import switch."""
import os

import numpy as np

from ..sb3.callbacks import BaseCallback
from ..sb3.results_plotter import load_results, ts2xy


class ProgressBarCallback(BaseCallback):
    """models/callbacks.py:8-20"""

    def __init__(self, pbar):
        super().__init__()
        self._pbar = pbar

    def _on_step(self):
        self._pbar.n = self.num_timesteps
        self._pbar.update(0)
        return True


class ProgressBarManager(object):
    """models/callbacks.py:23-38: ``with ProgressBarManager(total) as callback: model.learn(total, callback=callback)``"""

    def __init__(self, total_timesteps):
        self.pbar = None
        self.total_timesteps = total_timesteps

    def __enter__(self):
        from tqdm.auto import tqdm
        self.pbar = tqdm(total=self.total_timesteps)
        return ProgressBarCallback(self.pbar)

    def __exit__(self, exc_type, exc_val, exc_tb):
        self.pbar.n = self.total_timesteps
        self.pbar.update(0)
        self.pbar.close()


class SaveOnBestTrainingRewardCallback(BaseCallback):
    """models/callbacks.py:41-82: every ``check_freq`` calls, mean return of the last 100 monitored episodes; save the
    model to ``<log_dir>/best_model_training`` when it improves."""

    def __init__(self, check_freq, log_dir, verbose=1):
        super().__init__(verbose)
        self.check_freq = check_freq
        self.log_dir = log_dir
        self.save_path = os.path.join(log_dir, "best_model_training")
        self.best_mean_reward = -np.inf

    def _init_callback(self):
        if self.save_path is not None:
            os.makedirs(self.save_path, exist_ok=True)

    def _on_step(self):
        if self.n_calls % self.check_freq == 0:
            try:
                x, y = ts2xy(load_results(self.log_dir), "timesteps")
            except FileNotFoundError:
                return True
            if len(x) > 0:
                mean_reward = np.mean(y[-100:])
                if self.verbose > 0:
                    print(f"Num timesteps: {self.num_timesteps}")
                    print(f"Best mean reward: {self.best_mean_reward:.2f} - Last mean reward per episode: {mean_reward:.2f}")
                if mean_reward > self.best_mean_reward:
                    self.best_mean_reward = mean_reward
                    if self.verbose > 0:
                        print(f"Saving new best model to {self.save_path}.zip")
                    self.model.save(self.save_path)
        return True
