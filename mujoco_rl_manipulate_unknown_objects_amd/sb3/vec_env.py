"""VecEnv-shaped wrappers (stand-ins for stable_baselines3.common.vec_env / monitor).

``GpuVecEnv``    the native path: one BatchedRobotEnv = N envs on one GPU, tensors stay in HBM,
                 auto-reset inside the step kernel (DummyVecEnv semantics, train_agent.py:22).
``DummyVecEnv``  the reference's call shape ``DummyVecEnv([lambda: Monitor(env, path)])`` around the
                 one-env numpy RobotEnv (train_agent.py:22,31).
``Monitor``      episode return / length / wall-time CSV in SB3's format (``#{"t_start":..}`` header,
                 columns r,l,t) -- the input of the reference's plotting scripts.
"""
import json
import os
import time

import numpy as np


class GpuVecEnv:
    def __init__(self, batched_env):
        self.env = batched_env
        self.num_envs = batched_env.n_envs
        self.observation_space = batched_env.observation_space
        self.action_space = batched_env.action_space
        self.device = batched_env.device
        if not batched_env.batch.cfg.auto_reset:
            batched_env.batch.set_config(auto_reset=1)
        torch = batched_env.batch.torch
        self._ep_ret = torch.zeros(self.num_envs, device=self.device)
        self._ep_len = torch.zeros(self.num_envs, dtype=torch.int32, device=self.device)
        self.episode_returns, self.episode_lengths = [], []

    def reset(self):
        self._ep_ret.zero_(); self._ep_len.zero_()
        return self.env.reset()

    def step(self, actions):
        obs, rew, done, info = self.env.step(actions)
        self._ep_ret += rew; self._ep_len += 1
        return obs, rew, done, info

    def pop_finished(self, done):
        """Move finished episodes' (return, length) to host lists; one small D2H copy, call sparingly."""
        idx = done.nonzero().flatten()
        if idx.numel():
            self.episode_returns += self._ep_ret[idx].tolist(); self.episode_lengths += self._ep_len[idx].tolist()
            self._ep_ret[idx] = 0; self._ep_len[idx] = 0

    def close(self):
        self.env.close()


class Monitor:
    def __init__(self, env, filename=None):
        self.env = env
        self.observation_space, self.action_space = env.observation_space, env.action_space
        self.t_start = time.time()
        self.rewards, self.episode_returns, self.episode_lengths, self.episode_times = [], [], [], []
        self.file = None
        if filename is not None:
            if not filename.endswith("monitor.csv"):
                filename = filename + ".monitor.csv"
            os.makedirs(os.path.dirname(filename) or ".", exist_ok=True)
            self.file = open(filename, "wt")
            self.file.write("#%s\n" % json.dumps({"t_start": self.t_start, "env_id": None}))
            self.file.write("r,l,t\n"); self.file.flush()

    def reset(self, **kw):
        self.rewards = []
        return self.env.reset(**kw)

    def step(self, action):
        obs, reward, done, info = self.env.step(action)
        self.rewards.append(reward)
        if done:
            r, l, t = round(float(sum(self.rewards)), 6), len(self.rewards), round(time.time() - self.t_start, 6)
            info["episode"] = {"r": r, "l": l, "t": t}
            self.episode_returns.append(r); self.episode_lengths.append(l); self.episode_times.append(t)
            if self.file:
                self.file.write(f"{r},{l},{t}\n"); self.file.flush()
        return obs, reward, done, info

    def close(self):
        if self.file:
            self.file.close()
        self.env.close()

    def __getattr__(self, name):
        return getattr(self.env, name)


class DummyVecEnv:
    def __init__(self, env_fns):
        self.envs = [fn() for fn in env_fns]
        self.num_envs = len(self.envs)
        self.observation_space, self.action_space = self.envs[0].observation_space, self.envs[0].action_space

    def _stack(self, obs_list):
        return {k: np.stack([o[k] for o in obs_list]) for k in obs_list[0]}

    def reset(self):
        return self._stack([e.reset() for e in self.envs])

    def step(self, actions):
        obs, rews, dones, infos = [], [], [], []
        for e, a in zip(self.envs, actions):
            o, r, d, i = e.step(a)
            if d:
                i["terminal_observation"] = o
                o = e.reset()
            obs.append({k: np.array(v) for k, v in o.items()}); rews.append(r); dones.append(d); infos.append(i)
        return self._stack(obs), np.array(rews, dtype=np.float32), np.array(dones), infos

    def close(self):
        for e in self.envs:
            e.close()


class VecVideoRecorder:
    """stable_baselines3.common.vec_env.VecVideoRecorder's call shape (train_agent.py:25-29): wraps a VecEnv, starts a recording when
    `record_video_trigger(step)` fires and stops it `video_length` steps later. Frames come from the wrapped env's
    `render(mode='rgb_array')` (robot_env.py:302-340); the clip is written as an animated GIF (`<name_prefix>-step-<a>-to-step-<b>.gif`:
    PIL is what this image has, ffmpeg is not)."""

    def __init__(self, venv, video_folder, record_video_trigger, video_length=200, name_prefix="rl-video"):
        self.venv, self.video_folder, self.trigger, self.video_length, self.name_prefix = venv, video_folder, record_video_trigger, int(video_length), name_prefix
        self.num_envs = venv.num_envs
        self.observation_space, self.action_space = venv.observation_space, venv.action_space
        os.makedirs(video_folder, exist_ok=True)
        self.step_id, self.frames, self.start, self.recording, self.saved = 0, [], 0, False, []

    def _env0(self):
        e = self.venv
        while hasattr(e, "venv"):
            e = e.venv
        return e.envs[0] if hasattr(e, "envs") else getattr(e, "env", e)

    def _frame(self):
        return np.asarray(self._env0().render(mode="rgb_array"))

    def _maybe_start(self):
        if not self.recording and self.trigger(self.step_id):
            self.recording, self.frames, self.start = True, [self._frame()], self.step_id

    def reset(self):
        obs = self.venv.reset()
        self._maybe_start()
        return obs

    def step(self, actions):
        out = self.venv.step(actions)
        self.step_id += 1
        if self.recording:
            self.frames.append(self._frame())
            if len(self.frames) > self.video_length:
                self.close_video_recorder()
        else:
            self._maybe_start()
        return out

    def close_video_recorder(self):
        if self.recording and self.frames:
            from PIL import Image
            path = os.path.join(self.video_folder, f"{self.name_prefix}-step-{self.start}-to-step-{self.start + self.video_length}.gif")
            imgs = [Image.fromarray(f) for f in self.frames]
            imgs[0].save(path, format="GIF", append_images=imgs[1:], save_all=True, duration=100, loop=0)
            self.saved.append(path)
        self.recording, self.frames = False, []

    def close(self):
        self.close_video_recorder()
        self.venv.close()

    def __getattr__(self, name):
        return getattr(self.venv, name)
