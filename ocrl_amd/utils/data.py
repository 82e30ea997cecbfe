"""Synthetic "random-N5C4S4S2-style" scenes (own generator; the reference's HDF5 dataset and spriteworld are
not available offline).  Semantics from the reference's configs/env/random-N5C4S4S2.yaml:4-11 and
envs/synthetic_envs/base.py:81-151 / randomobjs.py:15-27: black background, 5 sprites, colour in
{blue, green, yellow, red}, shape in {square, triangle, star_4, circle}, scale in {0.15, 0.22} of the image
side, centres uniform in [r+0.08, 1-r-0.08]^2 with pairwise centre distance >= 0.15 (occlusion allowed) and
>= 0.15 from the undrawn agent position (0.5, 0.5); uint8 HWC, later /255 as utils/datasets.py:17."""
import numpy as np
import torch

COLORS = np.array([[0, 0, 255], [0, 255, 0], [255, 255, 0], [255, 0, 0]], dtype=np.uint8)
SCALES = (0.15, 0.22)


def _mask(shape_id, xx, yy, cx, cy, r):
    dx, dy = xx - cx, yy - cy
    if shape_id == 0:      # square
        return (np.abs(dx) <= r) & (np.abs(dy) <= r)
    if shape_id == 1:      # triangle (apex up)
        t = (dy + r) / (2 * r)
        return (t >= 0) & (t <= 1) & (np.abs(dx) <= r * t)
    if shape_id == 2:      # star_4
        return np.sqrt(np.abs(dx)) + np.sqrt(np.abs(dy)) <= np.sqrt(r) * 1.25
    return dx * dx + dy * dy <= r * r   # circle


def random_sprite_scenes(n, size, seed=0, num_objs=5, with_masks=False):
    """-> uint8 [n, size, size, 3] (and float masks [n, num_objs+1, size, size, 1], background last)"""
    rs = np.random.RandomState(seed)
    lin = (np.arange(size) + 0.5) / size
    xx, yy = np.meshgrid(lin, lin)
    out = np.zeros((n, size, size, 3), dtype=np.uint8)
    masks = np.zeros((n, num_objs + 1, size, size, 1), dtype=np.float32) if with_masks else None
    for i in range(n):
        centres = [(0.5, 0.5)]
        vis = np.zeros((size, size), dtype=np.int32)     # 0 = background, k = object k (later objects occlude)
        for k in range(num_objs):
            scale = SCALES[rs.randint(2)]
            r = scale / 2
            for _ in range(1000):
                cx, cy = rs.uniform(r + 0.08, 1 - r - 0.08, size=2)
                if all((cx - px) ** 2 + (cy - py) ** 2 >= 0.15 ** 2 for px, py in centres):
                    break
            centres.append((cx, cy))
            m = _mask(rs.randint(4), xx, yy, cx, cy, r)
            out[i][m] = COLORS[rs.randint(4)]
            vis[m] = k + 1
        if with_masks:
            for k in range(num_objs):
                masks[i, k, :, :, 0] = vis == k + 1
            masks[i, num_objs, :, :, 0] = vis == 0
    return (out, masks) if with_masks else out


def scenes_to_obs(u8):
    """uint8 [n,H,W,3] -> float32 [n,3,H,W] in [0,1] (utils/datasets.py:17)"""
    return torch.from_numpy(u8).permute(0, 3, 1, 2).float().div_(255.0).contiguous()
