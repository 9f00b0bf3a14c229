"""Host-side mirrors of simulation/utils/utils.py (numpy; API compatibility and tests only --
the rollout kernels carry their own fp32 versions of these formulas)."""
import numpy as np


def project_to_target_direction(pos, target_dir):
    """utils.py:30-31: scalar projection (p . d) / |d|^2; batched over leading axes."""
    pos = np.asarray(pos); target_dir = np.asarray(target_dir)
    return (pos * target_dir).sum(-1) / np.linalg.norm(target_dir) ** 2


def transform_depth(depth):
    """utils.py:11-19, in place like the original: shift by the min, scale by twice the mean of
    the near (<= 1 m) pixels, map to [0, 255]. NaN if no pixel is near (quirk Q7)."""
    depth -= depth.min()
    depth /= 2 * depth[depth <= 1].mean()
    return 255 * np.clip(depth, 0, 1)


def chw_to_hwc(img):
    return img.transpose((1, 2, 0))


def hwc_to_chw(img):
    return img.transpose((2, 0, 1))
