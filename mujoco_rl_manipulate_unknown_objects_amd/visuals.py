"""Evaluation visuals of the reference's eval_agent.py: the trajectory GIF (eval_agent.py:11-25) and the 3-D plot of gripper and
object paths (scripts/plot_3D.py:6, called from eval_agent.py:72-76). Host-side, PIL / matplotlib; not on any hot path."""
import os

import numpy as np


def make_gif(frames_in, path, duration=100):
    """eval_agent.py:11-25: frames (uint8 HxWx3 arrays, e.g. env.render('rgb_array') per step) -> looping GIF at `path`."""
    from PIL import Image
    os.makedirs(os.path.dirname(path) or ".", exist_ok=True)
    frames = [Image.fromarray(np.asarray(f, dtype=np.uint8)) for f in frames_in]
    frames[0].save(path, format="GIF", append_images=frames[1:], save_all=True, duration=duration, loop=0)
    return path


def plot_3D(gripper_positions, object_positions, path=None, title=""):
    """scripts/plot_3D.py:6: gripper and object positions of an evaluation episode as two 3-D curves; saved to `path` (PNG) when given."""
    import matplotlib
    matplotlib.use("Agg")
    import matplotlib.pyplot as plt
    g = np.asarray(gripper_positions, dtype=float).reshape(-1, 3); o = np.asarray(object_positions, dtype=float).reshape(-1, 3)
    fig = plt.figure(figsize=(7, 6)); ax = fig.add_subplot(111, projection="3d")
    ax.plot(g[:, 0], g[:, 1], g[:, 2], label="gripper", color="tab:blue"); ax.plot(o[:, 0], o[:, 1], o[:, 2], label="object", color="tab:orange")
    ax.scatter(*g[0], color="tab:blue", marker="o"); ax.scatter(*o[0], color="tab:orange", marker="o")
    ax.set_xlabel("x [m]"); ax.set_ylabel("y [m]"); ax.set_zlabel("z [m]"); ax.set_title(title); ax.legend()
    if path:
        os.makedirs(os.path.dirname(path) or ".", exist_ok=True)
        fig.savefig(path, dpi=100)
    plt.close(fig)
    return path
