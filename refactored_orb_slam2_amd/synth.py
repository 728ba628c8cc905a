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


def camera_pose(seed: int):
    """A seeded camera pose (Rcw, tcw): small rotation about a random axis plus a translation of a few metres."""
    rng = np.random.default_rng(seed)
    axis = rng.normal(size=3); axis /= np.linalg.norm(axis)
    ang = rng.uniform(-0.3, 0.3)
    K = np.array([[0, -axis[2], axis[1]], [axis[2], 0, -axis[0]], [-axis[1], axis[0], 0]])
    R = np.eye(3) + np.sin(ang) * K + (1 - np.cos(ang)) * (K @ K)
    t = rng.uniform(-3, 3, 3)
    return R.astype(np.float32), t.astype(np.float32)


def local_map(keys, desc, frustum, seed: int, n_extra: int = 500, dup_frac: float = 0.2) -> np.ndarray:
    """Synthetic mvpLocalMapPoints for one frame (records in the layout of orbfe_map_point): every keypoint of the frame is
    back-projected to a random depth so that it re-projects within ~1 px of the keypoint at a predicted level that
    contains the keypoint's octave; a fraction gets a second, noisier point (contention for the same keypoint);
    n_extra points are placed anywhere (behind the camera, outside the image, too near/far, oblique normals).  The
    descriptor is the keypoint's with 0..40 flipped bits; ~10 % of the points are flagged skip, ~80 % observed."""
    from ._lib import MAP_POINT_DTYPE
    rng = np.random.default_rng(seed)
    fr = frustum.reshape(-1)[0]
    R = fr["Rcw"].reshape(3, 3).astype(np.float64); t = fr["tcw"].astype(np.float64); Ow = fr["Ow"].astype(np.float64)
    sf = float(fr["scale_factors"][1]); nl = int(fr["n_levels"])
    n0 = len(keys)
    src = np.concatenate([np.arange(n0), rng.choice(n0, int(dup_frac * n0), replace=False)]) if n0 else np.zeros(0, np.int64)
    n = len(src)
    mp = np.zeros(n + n_extra, MAP_POINT_DTYPE)
    if n:
        z = rng.uniform(4.0, 40.0, n)
        u = keys["x"][src].astype(np.float64) + rng.normal(0, 0.6, n)
        v = keys["y"][src].astype(np.float64) + rng.normal(0, 0.6, n)
        Xc = np.stack([(u - fr["cx"]) * z / fr["fx"], (v - fr["cy"]) * z / fr["fy"], z], 1)
        P = (Xc - t) @ R          # R^T (Xc - t), row form
        PO = P - Ow
        dist = np.linalg.norm(PO, axis=1)
        # normal: viewing ray tilted by up to ~70 degrees (beyond 60 the point fails viewCos >= 0.5)
        tilt = rng.uniform(0, 1.22, n) * (rng.random(n) < 0.5)
        perp = np.cross(PO, rng.normal(size=(n, 3))); perp /= np.linalg.norm(perp, axis=1, keepdims=True)
        nrm = np.cos(tilt)[:, None] * PO / dist[:, None] + np.sin(tilt)[:, None] * perp
        e = keys["octave"][src] + rng.uniform(-0.95, 0.95, n)
        e[::17] = np.round(e[::17])            # ratio = sf**k: the ceil() boundary
        maxd = dist * sf ** e
        mp["pos"][:n] = P; mp["normal"][:n] = nrm
        mp["max_distance"][:n] = maxd; mp["min_distance"][:n] = maxd / sf ** (nl - 1)
        d = desc[src].copy()
        flips = rng.integers(0, 41, n)
        for i in range(n):
            b = rng.choice(256, flips[i], replace=False)
            np.bitwise_xor.at(d[i], b >> 3, (1 << (b & 7)).astype(np.uint8))
        mp["desc"][:n] = d
    if n_extra:
        mp["pos"][n:] = rng.uniform(-60, 60, (n_extra, 3)) + Ow
        nrm = rng.normal(size=(n_extra, 3)); mp["normal"][n:] = nrm / np.linalg.norm(nrm, axis=1, keepdims=True)
        md = rng.uniform(5, 120, n_extra)
        mp["max_distance"][n:] = md; mp["min_distance"][n:] = md / sf ** (nl - 1)
        mp["desc"][n:] = rng.integers(0, 256, (n_extra, 32), dtype=np.uint8)
    mp["skip"] = rng.random(len(mp)) < 0.1
    mp["observed"] = rng.random(len(mp)) < 0.8
    return mp[rng.permutation(len(mp))]
