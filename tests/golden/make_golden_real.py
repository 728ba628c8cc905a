#!/usr/bin/env python3
"""Real-image fixture: the four 640x480 grey-level PNGs of the DBoW2 demo that ships inside the reference checkout
(Source/ThirdParty/DBoW2/DBoW2-local/demo/images/image{0..3}.png, data files under DBoW2's BSD licence) -- the only real
images the reference holds.  Natural texture exercises what the synthetic generator barely produces:
score plateaus, saturated regions, flat cells that fall back to minThFAST, NMS ties, octree early-outs.

Run in the build container (the GPU box has no /root/reference): python tests/golden/make_golden_real.py
Writes tests/golden/real_demo.npz = the images (inputs) + the oracle's outputs at the TUM configuration (1000 features,
1.2, 8 levels, 20 / 7), every extractor stage cross-checked against tests/np_restatement.py before anything is written,
and matcher vectors between consecutive images (SearchByBoW on a synthetic vocabulary, SearchByProjection(cur,last),
SearchForInitialization, the grouped Hamming brute force).  The reference cannot be built here (no OpenCV), so like the
other fixtures these pin the ORACLE on realistic inputs; oracle/opencv_check.cpp diffs the oracle's primitives against a
real OpenCV wherever one exists.
"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from tests import np_restatement as nr  # noqa: E402
from tests import oracle_lib as ol  # noqa: E402

OUT = os.path.dirname(os.path.abspath(__file__))
SRC = "/root/reference/Source/ThirdParty/DBoW2/DBoW2-local/demo/images"
NFEAT = 1000
PATTERN = np.array([ol.lib().oo_pattern()[i] for i in range(1024)])


def load_images():
    from PIL import Image
    imgs = []
    for i in range(4):
        im = Image.open(os.path.join(SRC, f"image{i}.png"))
        a = np.asarray(im.convert("L"), np.uint8)
        assert a.shape == (480, 640), a.shape
        imgs.append(np.ascontiguousarray(a))
    # a fifth frame with known motion on real texture: image0 moved by (+4, +2) pixels, borders replicated
    imgs.append(np.ascontiguousarray(np.pad(imgs[0], ((2, 0), (4, 0)), mode="edge")[:480, :640]))
    return np.stack(imgs)


def node_groups(voc, desc):
    """DBoW2::FeatureVector of the descriptors on the synthetic vocabulary: {node id at level L - 2: [feature indices]}"""
    return voc.transform(desc, levelsup=2)[1]


def main():
    imgs = load_images()
    data = {"images": imgs, "nfeatures": np.int32(NFEAT)}
    e = ol.OracleExtractor(NFEAT, 1.2, 8, 20, 7)
    sf = e.scale_factors
    K, D = [], []
    stats = []
    for i, img in enumerate(imgs):
        k, d = e(img)
        K.append(k); D.append(d)
        data[f"kp_{i}"] = k
        data[f"desc_{i}"] = d
        for l in range(8):
            x, y, s = e.level_candidates(l)
            lv = e.level_pixels(l)
            nx, ny, ns = nr.fast_candidates(lv)   # independent restatement must agree before anything is written
            assert np.array_equal(x, nx) and np.array_equal(y, ny) and np.array_equal(s, ns), (i, l)
            if l > 0:
                assert np.array_equal(nr.resize_linear(e.level_pixels(l - 1), lv.shape[1], lv.shape[0]), lv)
            assert np.array_equal(nr.gaussian_blur7(lv), e.level_blurred(l))
            sel = nr.distribute_octree(x, y, s, 16, lv.shape[1] - 16, 16, lv.shape[0] - 16, e.features_per_level[l])
            kl = e.level_keypoints(l)
            assert np.array_equal(x[sel] + 16, kl["x"].astype(np.int64)) and np.array_equal(y[sel] + 16, kl["y"].astype(np.int64))
            data[f"cand_{i}_{l}"] = np.stack([x, y, s]).astype(np.int16)
            data[f"lkp_{i}_{l}"] = np.stack([kl["x"], kl["y"], kl["response"]]).astype(np.int16)
        stats.append((len(k), [int(data[f"cand_{i}_{l}"].shape[1]) for l in range(8)]))
        # orientation / descriptor restatements on every 7th keypoint of every level
        off = 0
        for l in range(8):
            kl = e.level_keypoints(l)
            lv, bl = e.level_pixels(l), e.level_blurred(l)
            for j in range(0, len(kl), 7):
                ang = nr.ic_angle(lv, int(kl["x"][j]), int(kl["y"][j]))
                assert ang == k["angle"][off + j], (i, l, j)
                assert np.array_equal(nr.orb_descriptor(bl, int(kl["x"][j]), int(kl["y"][j]), ang, PATTERN), d[off + j]), (i, l, j)
            off += len(kl)
    # ---- matcher vectors between consecutive images
    voc = ol.OracleVocabulary.from_arrays(10, 3, *ol.synthetic_vocabulary(k=10, L=3, seed=5))
    for (a, b) in ((0, 1), (2, 3), (0, 4)):
        ka, da, kb, db = K[a], D[a], K[b], D[b]
        ga, gb = node_groups(voc, da), node_groups(voc, db)
        valid = (np.arange(len(da)) % 11 != 0).astype(np.uint8)
        nm, matchB = ol.search_by_bow(da, ka["angle"], valid, ga, db, kb["angle"], gb, np.float32(0.7), True)
        data[f"bow_{a}{b}_nodesA"] = np.array([n for n, v in sorted(ga.items()) for _ in v], np.int32)
        data[f"bow_{a}{b}_idxA"] = np.array([i for n, v in sorted(ga.items()) for i in v], np.int32)
        data[f"bow_{a}{b}_nodesB"] = np.array([n for n, v in sorted(gb.items()) for _ in v], np.int32)
        data[f"bow_{a}{b}_idxB"] = np.array([i for n, v in sorted(gb.items()) for i in v], np.int32)
        data[f"bow_{a}{b}_valid"] = valid
        data[f"bow_{a}{b}_nm"] = np.int32(nm)
        data[f"bow_{a}{b}_matchB"] = matchB
        # SearchByProjection(cur = b, last = a): queries at a's keypoint positions, window 15 * scale, +-1 level
        q = np.zeros(len(ka), ol.QUERY_DTYPE)
        q["u"] = ka["x"]; q["v"] = ka["y"]; q["u_r"] = -1
        q["radius"] = np.float32(15.0) * sf[ka["octave"]]
        q["min_level"] = ka["octave"] - 1; q["max_level"] = ka["octave"] + 1
        q["valid"] = 1; q["blocks"] = 1; q["angle"] = ka["angle"]; q["desc"] = da
        of = ol.OracleFrame(kb, db, sf, 0, 640, 0, 480)
        pnm, pa, pb = of.search_by_projection_frame(q, True)
        data[f"proj_{a}{b}_nm"] = np.int32(pnm); data[f"proj_{a}{b}_assigned"] = pa; data[f"proj_{a}{b}_blocked"] = pb
        # SearchByProjection(F, MapPoints)-style ratio test on the same windows, levels [l-1, l]
        q2 = q.copy(); q2["max_level"] = ka["octave"]
        qnm, qa, qb = of.search_by_projection_points(q2, np.float32(0.8))
        data[f"points_{a}{b}_nm"] = np.int32(qnm); data[f"points_{a}{b}_assigned"] = qa
        # SearchForInitialization(F1 = a, F2 = b, window 100)
        prev = np.stack([ka["x"], ka["y"]], axis=1).astype(np.float32)
        inm, m12, p2 = ol.search_for_initialization(ka, da, of, prev, 100, np.float32(0.9), True)
        data[f"init_{a}{b}_nm"] = np.int32(inm); data[f"init_{a}{b}_m12"] = m12; data[f"init_{a}{b}_prev"] = p2
        # grouped brute force (the inner loops of SearchByBoW)
        grpA = np.zeros(len(da), np.int32); grpB = np.zeros(len(db), np.int32)
        for n, v in ga.items():
            grpA[v] = n
        for n, v in gb.items():
            grpB[v] = n
        bi, bd, sd = ol.hamming_bf(da, db, grpA, grpB)
        data[f"bf_{a}{b}"] = np.stack([bi, bd, sd]).astype(np.int32)
        print(f"pair {a}{b}: SearchByBoW {nm}, SearchByProjection(frame) {pnm}, (points) {qnm}, init {inm}")
    np.savez_compressed(os.path.join(OUT, "real_demo.npz"), **data)
    for i, (n, c) in enumerate(stats):
        print(f"image{i}: {n} keypoints, candidates per level {c}")


if __name__ == "__main__":
    main()
