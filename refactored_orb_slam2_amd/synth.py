"""Seeded synthetic grey-level imagery for parity tests and the benchmark.

No dataset (KITTI/TUM/EuRoC) exists on either box, so every input is generated here:
a base scene = 3 octaves of bilinearly up-sampled uniform noise (cell 64/16/4 px, amplitude
60/40/25) + 400 random axis-aligned / rotated rectangles (FAST corners at every pyramid level)
+ +-3 grey-level white noise, clamped to [0,255].  A sequence crops the scene with a 2 px/frame
translation so consecutive frames overlap; the right image of a stereo pair is the left one with
a per-row integer disparity growing from 5 px (top) to 60 px (bottom).

Frame f of sequence s uses numpy's PCG64 seeded with 0xB5EED + 1000*s + f.
"""
from __future__ import annotations

import numpy as np

SCENE_W, SCENE_H = 1600, 600
SEED0 = 0xB5EED


def _bilinear_upsample(small: np.ndarray, h: int, w: int) -> np.ndarray:
    sh, sw = small.shape
    ys = np.linspace(0, sh - 1, h)
    xs = np.linspace(0, sw - 1, w)
    y0 = np.floor(ys).astype(np.int64).clip(0, sh - 2)
    x0 = np.floor(xs).astype(np.int64).clip(0, sw - 2)
    fy = (ys - y0)[:, None]
    fx = (xs - x0)[None, :]
    a = small[y0][:, x0]
    b = small[y0][:, x0 + 1]
    c = small[y0 + 1][:, x0]
    d = small[y0 + 1][:, x0 + 1]
    return (a * (1 - fx) + b * fx) * (1 - fy) + (c * (1 - fx) + d * fx) * fy


def make_scene(seed: int, w: int = SCENE_W, h: int = SCENE_H, n_rects: int = 400) -> np.ndarray:
    """Base scene, uint8 (h, w)."""
    rng = np.random.Generator(np.random.PCG64(seed))
    img = np.full((h, w), 110.0)
    for cell, amp in ((64, 60.0), (16, 40.0), (4, 25.0)):
        small = rng.random((h // cell + 2, w // cell + 2)) - 0.5
        img += amp * _bilinear_upsample(small, h, w)
    yy, xx = np.mgrid[0:h, 0:w]
    for _ in range(n_rects):
        cx, cy = rng.uniform(0, w), rng.uniform(0, h)
        hw, hh = rng.uniform(4, 60), rng.uniform(4, 60)
        grey = rng.uniform(10, 245)
        theta = 0.0 if rng.random() < 0.5 else rng.uniform(0, np.pi)
        r = int(np.ceil(np.hypot(hw, hh))) + 1
        x0, x1 = max(0, int(cx) - r), min(w, int(cx) + r + 1)
        y0, y1 = max(0, int(cy) - r), min(h, int(cy) + r + 1)
        if x0 >= x1 or y0 >= y1:
            continue
        dx = xx[y0:y1, x0:x1] - cx
        dy = yy[y0:y1, x0:x1] - cy
        c, s = np.cos(theta), np.sin(theta)
        u = dx * c + dy * s
        v = -dx * s + dy * c
        mask = (np.abs(u) <= hw) & (np.abs(v) <= hh)
        img[y0:y1, x0:x1][mask] = grey
    img += rng.integers(-3, 4, size=(h, w))
    return np.clip(np.rint(img), 0, 255).astype(np.uint8)


def _crop(scene: np.ndarray, w: int, h: int, ox: int, oy: int) -> np.ndarray:
    sh, sw = scene.shape
    ys = (np.arange(h) + oy) % sh
    xs = (np.arange(w) + ox) % sw
    return np.ascontiguousarray(scene[ys][:, xs])


def frame(w: int, h: int, seq: int = 0, f: int = 0, scene: np.ndarray | None = None) -> np.ndarray:
    """Left/mono image of frame f in sequence seq, uint8 (h, w)."""
    if scene is None:
        scene = make_scene(SEED0 + 1000 * seq, max(SCENE_W, w + 400), max(SCENE_H, h + 64))
    rng = np.random.Generator(np.random.PCG64(SEED0 + 1000 * seq + f + 1))
    img = _crop(scene, w, h, 100 + 2 * f, 8).astype(np.int16)
    img += rng.integers(-2, 3, size=img.shape, dtype=np.int16)  # per-frame sensor noise
    return np.clip(img, 0, 255).astype(np.uint8)


def stereo_pair(w: int, h: int, seq: int = 0, f: int = 0, scene: np.ndarray | None = None):
    """(left, right): right(x, y) = scene(x + d(y)) with d from 5 px (top) to 60 px (bottom)."""
    if scene is None:
        scene = make_scene(SEED0 + 1000 * seq, max(SCENE_W, w + 400), max(SCENE_H, h + 64))
    left = frame(w, h, seq, f, scene)
    rng = np.random.Generator(np.random.PCG64(SEED0 + 1000 * seq + f + 500_000))
    sh, sw = scene.shape
    disp = np.rint(5 + 55 * np.arange(h) / max(h - 1, 1)).astype(np.int64)
    ys = (np.arange(h) + 8) % sh
    xs = (np.arange(w)[None, :] + 100 + 2 * f + disp[:, None]) % sw
    right = scene[ys[:, None], xs].astype(np.int16)
    right += rng.integers(-2, 3, size=right.shape, dtype=np.int16)
    return left, np.clip(right, 0, 255).astype(np.uint8)


def sequence(w: int, h: int, n: int, seq: int = 0, stereo: bool = False):
    """List of n frames (or (left,right) pairs) sharing one scene."""
    scene = make_scene(SEED0 + 1000 * seq, max(SCENE_W, w + 400 + 2 * n), max(SCENE_H, h + 64))
    if stereo:
        return [stereo_pair(w, h, seq, f, scene) for f in range(n)]
    return [frame(w, h, seq, f, scene) for f in range(n)]
