#!/usr/bin/env python3
"""CPU-only statistics behind the FAST quick-test layout decisions (profiles/r06_fast.md): on the pyramid levels of the synthetic
KITTI frames and of the DBoW2 demo images, how often does NO pixel of a 64-lane block (128 pixels) pass tier 1 (antipodal pairs
(0, 8), (4, 12)) / the whole quick test (four pairs), for different block shapes -- the wave-uniform early-out of
fast_cells_kernel -- and how many lane pairs / pixels per cell pass.  Uses the oracle only to build the pyramid levels.
    python tools/fast_block_stats.py"""
import os, sys
import numpy as np
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from refactored_orb_slam2_amd import synth
from tests import oracle_lib as ol

PAIRS1 = [((0, 3), (0, -3)), ((3, 0), (-3, 0))]
PAIRS2 = PAIRS1 + [((2, 2), (-2, -2)), ((2, -2), (-2, 2))]


def quick(img, t, pairs):
    I = img.astype(np.int32); H, W = I.shape
    v = I[3:H - 3, 3:W - 3]
    P = lambda dx, dy: I[3 + dy:H - 3 + dy, 3 + dx:W - 3 + dx]
    lo = hi = None
    for a, b in pairs:
        mn, mx = np.minimum(P(*a), P(*b)), np.maximum(P(*a), P(*b))
        lo = mn if lo is None else np.maximum(lo, mn); hi = mx if hi is None else np.minimum(hi, mx)
    return (lo < v - t) | (hi > v + t)


def levels(imgs):
    ex = ol.OracleExtractor(2000)
    out = []
    for im in imgs:
        ex(im); out += [ex.level_pixels(l).copy() for l in range(8)]
    return out


def block_stats(lv, t):
    row = {}
    for bh, bw in [(4, 31), (8, 16), (2, 62), (1, 124)]:
        tot = e1 = e2 = 0
        for im in lv:
            m1, m2 = quick(im, t, PAIRS1)[13:-13, 13:-13], quick(im, t, PAIRS2)[13:-13, 13:-13]
            Hh, Ww = (m1.shape[0] // bh) * bh, (m1.shape[1] // bw) * bw
            b1 = m1[:Hh, :Ww].reshape(Hh // bh, bh, Ww // bw, bw).any(axis=(1, 3))
            b2 = m2[:Hh, :Ww].reshape(Hh // bh, bh, Ww // bw, bw).any(axis=(1, 3))
            tot += b1.size; e1 += (~b1).sum(); e2 += (~b2).sum()
        row[f"{bh}x{bw}"] = (round(e1 / tot, 3), round(e2 / tot, 3))
    return row


def cell_stats(lv, t):
    pr, p1, p2 = [], [], []
    for im in lv:
        m1, m2 = quick(im, t, PAIRS1)[13:-13, 13:-13], quick(im, t, PAIRS2)[13:-13, 13:-13]
        Hh, Ww = (m1.shape[0] // 32) * 32, (m1.shape[1] // 32) * 32
        a = m1[:Hh, :Ww].reshape(Hh // 32, 32, Ww // 32, 16, 2)
        pr += list(a.any(axis=4).sum(axis=(1, 3)).ravel()); p1 += list(a.sum(axis=(1, 3, 4)).ravel())
        p2 += list(m2[:Hh, :Ww].reshape(Hh // 32, 32, Ww // 32, 32).sum(axis=(1, 3)).ravel())
    pr = np.array(pr)
    return dict(pairs_per_cell=round(float(pr.mean()), 1), p90=float(np.percentile(pr, 90)), tier1_pixels=round(float(np.mean(p1)), 1),
                quick_test_pixels=round(float(np.mean(p2)), 1), rounds_of_64_pairs_per_cell=round(float(np.ceil(pr / 64).mean()), 2))


if __name__ == "__main__":
    syn = levels([p[0] for p in synth.sequence(1241, 376, 2, seq=0, stereo=True)])
    demo = levels(list(np.load(os.path.join(os.path.dirname(__file__), "..", "tests", "golden", "real_demo.npz"))["images"][:2]))
    for name, lv in (("synthetic", syn), ("demo", demo)):
        for t in (20, 7):
            print(name, "t =", t, "| share of 128-pixel blocks with no pass (tier 1, four pairs) by block shape rows x cols:", block_stats(lv, t))
            print(name, "t =", t, "| per 32 x 32 cell:", cell_stats(lv, t))
