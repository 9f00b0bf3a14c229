"""Callback classes with SB3's shape (stable_baselines3.common.callbacks), enough for the
reference's models/callbacks.py:8-82 and train_agent.py:34-43 to run against this PPO."""
import os

import numpy as np


class BaseCallback:
    def __init__(self, verbose=0):
        self.model = None
        self.training_env = None
        self.n_calls = 0
        self.num_timesteps = 0
        self.verbose = verbose
        self.locals, self.globals = {}, {}

    def init_callback(self, model):
        self.model = model
        self.training_env = model.env
        self._init_callback()

    def _init_callback(self):
        pass

    def on_training_start(self, locals_, globals_):
        self.locals, self.globals = locals_, globals_
        self._on_training_start()

    def _on_training_start(self):
        pass

    def on_rollout_start(self):
        pass

    def on_rollout_end(self):
        pass

    def on_step(self):
        self.n_calls += 1
        self.num_timesteps = self.model.num_timesteps
        return self._on_step()

    def _on_step(self):
        return True

    def on_training_end(self):
        pass


class CallbackList(BaseCallback):
    def __init__(self, callbacks):
        super().__init__()
        self.callbacks = callbacks

    def _init_callback(self):
        for c in self.callbacks:
            c.init_callback(self.model)

    def _on_training_start(self):
        for c in self.callbacks:
            c.on_training_start(self.locals, self.globals)

    def _on_step(self):
        ok = True
        for c in self.callbacks:
            ok = c.on_step() and ok
        return ok

    def on_training_end(self):
        for c in self.callbacks:
            c.on_training_end()


class EvalCallback(BaseCallback):
    """Periodic deterministic evaluation; writes ``evaluations.npz`` (keys timesteps, results,
    ep_lengths) and saves ``best_model`` like SB3's EvalCallback (train_agent.py:34-40)."""

    def __init__(self, eval_env, best_model_save_path=None, log_path=None, eval_freq=2000, n_eval_episodes=3,
                 deterministic=True, render=False, verbose=0):
        super().__init__(verbose)
        self.eval_env, self.best_model_save_path, self.log_path = eval_env, best_model_save_path, log_path
        self.eval_freq, self.n_eval_episodes, self.deterministic = eval_freq, n_eval_episodes, deterministic
        self.best_mean_reward = -np.inf
        self.evaluations_timesteps, self.evaluations_results, self.evaluations_length = [], [], []

    def _on_step(self):
        if self.eval_freq <= 0 or self.n_calls % self.eval_freq:
            return True
        returns, lengths = evaluate_policy(self.model, self.eval_env, self.n_eval_episodes, self.deterministic)
        self.evaluations_timesteps.append(self.num_timesteps); self.evaluations_results.append(returns); self.evaluations_length.append(lengths)
        if self.log_path:
            os.makedirs(self.log_path, exist_ok=True)
            np.savez(os.path.join(self.log_path, "evaluations"), timesteps=self.evaluations_timesteps,
                     results=self.evaluations_results, ep_lengths=self.evaluations_length)
        mean = float(np.mean(returns))
        if mean > self.best_mean_reward:
            self.best_mean_reward = mean
            if self.best_model_save_path:
                self.model.save(os.path.join(self.best_model_save_path, "best_model"))
        return True


def evaluate_policy(model, env, n_eval_episodes=3, deterministic=True, max_steps=100000):
    """Episode returns / lengths of `model` on a one-env numpy VecEnv or raw env."""
    returns, lengths = [], []
    vec = hasattr(env, "num_envs")
    for _ in range(n_eval_episodes):
        obs = env.reset(); total, n = 0.0, 0
        for _ in range(max_steps):
            action, _ = model.predict(obs, deterministic=deterministic)
            obs, r, d, _ = env.step(action)
            total += float(np.asarray(r).reshape(-1)[0]) if vec else float(r); n += 1
            if (bool(np.asarray(d).reshape(-1)[0]) if vec else bool(d)):
                break
        returns.append(total); lengths.append(n)
    return returns, lengths
