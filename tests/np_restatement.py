"""Independent numpy restatement of the front-end primitives (TEST INFRASTRUCTURE).

Structurally different from oracle/orb_oracle.c on purpose: vectorised whole-level formulations
instead of literal per-cell / per-node loops.  Used to cross-check the C oracle (the reference has no
golden vectors: SURVEY.md §4, §8(c)) and to validate the data-parallel formulations the HIP kernels use:

  * FAST: one score map per level at minThFAST, NMS restricted to the cell's tested region, two-threshold
    cell rule (SURVEY.md §8(a) row A3 "derived single-pass form").
  * DistributeOctTree: leaves kept as an array in std::list order; one iteration = quadrant counts per
    key, prefix sums, array rebuild (no pointers; ties ordered by list position = creation order).

Reference citations: L/ = Source/Libraries/ORB_SLAM2/.
"""
from __future__ import annotations

import numpy as np

RING = [(0, 3), (1, 3), (2, 2), (3, 1), (3, 0), (3, -1), (2, -2), (1, -3), (0, -3), (-1, -3), (-2, -2),
        (-3, -1), (-3, 0), (-3, 1), (-2, 2), (-1, 3)]


def cv_round(v):
    return np.rint(v).astype(np.int64)  # numpy rint = round half to even


# ------------------------------------------------------------------------------------------ resize (P2)
def _axis_tables(s, d, clamp_f):
    scale = 1.0 / (float(d) / float(s))
    i = np.arange(d, dtype=np.float64)
    f = ((i + 0.5) * scale - 0.5).astype(np.float32)
    si = np.floor(f).astype(np.int64)
    f = (f - si.astype(np.float32)).astype(np.float32)
    if clamp_f:
        neg = si < 0
        f[neg] = 0; si[neg] = 0
        hi = si >= s - 1
        f[hi] = 0; si[hi] = s - 1
    c0 = np.clip(cv_round((np.float32(1) - f) * np.float32(2048)), -32768, 32767)
    c1 = np.clip(cv_round(f * np.float32(2048)), -32768, 32767)
    return si, c0, c1


def resize_linear(src, dw, dh):
    sh, sw = src.shape
    xo, a0, a1 = _axis_tables(sw, dw, True)
    yo, b0, b1 = _axis_tables(sh, dh, False)
    S = src.astype(np.int64)
    x1 = np.minimum(xo + 1, sw - 1)
    H = S[:, xo] * a0[None, :] + S[:, x1] * a1[None, :]  # (sh, dw)
    y0 = np.clip(yo, 0, sh - 1)
    y1 = np.clip(yo + 1, 0, sh - 1)
    T0 = H[y0] >> 4
    T1 = H[y1] >> 4
    out = (((b0[:, None] * T0) >> 16) + ((b1[:, None] * T1) >> 16) + 2) >> 2
    return out.astype(np.uint8)


# -------------------------------------------------------------------------------------------- blur (P3)
GAUSS_TAPS = np.array([18, 34, 48, 56, 48, 34, 18], dtype=np.int64)


def gaussian_blur7(src):
    p = np.pad(src.astype(np.int64), 3, mode="reflect")  # numpy 'reflect' == BORDER_REFLECT_101
    h, w = src.shape
    hor = sum(GAUSS_TAPS[k] * p[:, k:k + w] for k in range(7))
    ver = sum(GAUSS_TAPS[k] * hor[k:k + h, :] for k in range(7))
    return ((ver + 32768) >> 16).astype(np.uint8)


# -------------------------------------------------------------------------------------------- FAST (P4)
def fast_score_map(img):
    """S(p) = largest t such that p is a FAST-9/16 corner at threshold t (=-1.. if never); computed for
    every pixel that has a full ring.  Border pixels get -1."""
    h, w = img.shape
    I = img.astype(np.int64)
    S = np.full((h, w), -1, dtype=np.int64)
    c = I[3:h - 3, 3:w - 3]
    d = np.stack([c - I[3 + dy:h - 3 + dy, 3 + dx:w - 3 + dx] for dx, dy in RING])  # v - ring
    dd = np.concatenate([d, d[:8]], axis=0)
    # min / max over every 9-arc
    amin = np.full(c.shape, -10 ** 6)
    bmin = np.full(c.shape, -10 ** 6)
    for k in range(16):
        arc = dd[k:k + 9]
        amin = np.maximum(amin, arc.min(axis=0))        # dark ring: all v-p > t
        bmin = np.maximum(bmin, (-arc).min(axis=0))     # bright ring: all p-v > t
    S[3:h - 3, 3:w - 3] = np.maximum(amin, bmin) - 1
    return S


def level_cells(w, h):
    """Cell geometry of ComputeKeyPointsOctTree (L/src/ORBextractor.cc:740-771).
    Returns list of (i, j, iniX, iniY, maxX, maxY) for cells that are not skipped, plus (wCell, hCell)."""
    minB = 16
    maxBX, maxBY = w - 16, h - 16
    width, height = np.float32(maxBX - minB), np.float32(maxBY - minB)
    nCols, nRows = int(width / np.float32(30)), int(height / np.float32(30))
    if nCols < 1 or nRows < 1:
        return [], (0, 0)
    wCell = int(np.ceil(width / np.float32(nCols)))
    hCell = int(np.ceil(height / np.float32(nRows)))
    cells = []
    for i in range(nRows):
        iniY = minB + i * hCell
        maxY = iniY + hCell + 6
        if iniY >= maxBY - 3:
            continue
        maxY = min(maxY, maxBY)
        for j in range(nCols):
            iniX = minB + j * wCell
            maxX = iniX + wCell + 6
            if iniX >= maxBX - 6:
                continue
            maxX = min(maxX, maxBX)
            cells.append((i, j, iniX, iniY, maxX, maxY))
    return cells, (wCell, hCell)


def fast_candidates(img, ini_th=20, min_th=7):
    """Single-pass form: score map once, per-cell NMS region, two-threshold rule.
    Returns (x, y, score) relative to (16,16) in the reference's vToDistributeKeys order."""
    h, w = img.shape
    S = fast_score_map(img)
    S = np.where(S >= min_th, S, 0)
    cells, (wCell, hCell) = level_cells(w, h)
    xs, ys, ss = [], [], []
    for (i, j, iniX, iniY, maxX, maxY) in cells:
        x0, x1, y0, y1 = iniX + 3, maxX - 3, iniY + 3, maxY - 3  # tested region [x0,x1) x [y0,y1)
        if x1 <= x0 or y1 <= y0:
            continue
        R = np.zeros((y1 - y0 + 2, x1 - x0 + 2), dtype=np.int64)
        R[1:-1, 1:-1] = S[y0:y1, x0:x1]
        C = R[1:-1, 1:-1]
        keep = C > 0
        for dy in (-1, 0, 1):
            for dx in (-1, 0, 1):
                if dx or dy:
                    keep &= C > R[1 + dy:R.shape[0] - 1 + dy, 1 + dx:R.shape[1] - 1 + dx]
        hi = keep & (C >= ini_th)
        sel = hi if hi.any() else keep
        yy, xx = np.nonzero(sel)  # row-major
        xs.append(xx + x0 - 16); ys.append(yy + y0 - 16); ss.append(C[yy, xx])
    if not xs:
        z = np.zeros(0, np.int64)
        return z, z, z
    return np.concatenate(xs), np.concatenate(ys), np.concatenate(ss)


# ------------------------------------------------------------------------------------------ octree (A4)
def distribute_octree(x, y, score, minX, maxX, minY, maxY, N):
    """Array formulation of DistributeOctTree (L/src/ORBextractor.cc:529-731).  Leaves are kept in an
    array in list order (index 0 = list head).  Returns selected candidate indices in list order."""
    x = np.asarray(x, np.int64); y = np.asarray(y, np.int64); score = np.asarray(score, np.int64)
    n = len(x)
    nIni = int(np.floor(np.float32(maxX - minX) / np.float32(maxY - minY) + np.float32(0.5)))  # std::round, positive
    if nIni < 1:
        return np.zeros(0, np.int64)
    hX = np.float32(maxX - minX) / np.float32(nIni)
    # leaves: bbox arrays + per-key node id (position in the leaf array)
    ini = np.arange(nIni)
    bx0 = (hX * ini.astype(np.float32)).astype(np.int64)
    bx1 = (hX * (ini + 1).astype(np.float32)).astype(np.int64)
    by0 = np.zeros(nIni, np.int64)
    by1 = np.full(nIni, maxY - minY, np.int64)
    node = np.minimum((x.astype(np.float32) / hX).astype(np.int64), nIni - 1) if n else np.zeros(0, np.int64)
    cnt = np.bincount(node, minlength=nIni)
    keepm = cnt > 0  # erase empty ini nodes, order preserved
    remap = np.cumsum(keepm) - 1
    bx0, bx1, by0, by1, cnt = bx0[keepm], bx1[keepm], by0[keepm], by1[keepm], cnt[keepm]
    node = remap[node]

    def quadrants():
        halfX = (bx1 - bx0 + 1) // 2  # ceil(w/2) for w >= 0
        halfY = (by1 - by0 + 1) // 2
        mx, my = bx0 + halfX, by0 + halfY
        q = (x >= mx[node]).astype(np.int64) + 2 * (y >= my[node]).astype(np.int64)  # 0:n1 1:n2 2:n3 3:n4
        c4 = np.zeros((len(bx0), 4), np.int64)
        np.add.at(c4, (node, q), 1)
        return q, c4, mx, my

    def rebuild(split_order, q, c4, mx, my):
        """split_order: leaf positions to split, in processing (= child creation) order."""
        nonlocal bx0, bx1, by0, by1, cnt, node
        L = len(bx0)
        is_split = np.zeros(L, bool); is_split[split_order] = True
        # children in creation order
        cb = []
        child_of = {}
        for p in split_order:
            for qq in range(4):
                if c4[p, qq] > 0:
                    x0 = bx0[p] if qq in (0, 2) else mx[p]
                    x1 = mx[p] if qq in (0, 2) else bx1[p]
                    y0 = by0[p] if qq in (0, 1) else my[p]
                    y1 = my[p] if qq in (0, 1) else by1[p]
                    child_of[(p, qq)] = len(cb)
                    cb.append((x0, x1, y0, y1, c4[p, qq]))
        K = len(cb)
        stay = np.nonzero(~is_split)[0]
        newpos_stay = np.full(L, -1, np.int64)
        newpos_stay[stay] = K + np.arange(len(stay))
        nb = np.zeros((K + len(stay), 5), np.int64)
        for j, rec in enumerate(cb):
            nb[K - 1 - j] = rec
        nb[K:, 0] = bx0[stay]; nb[K:, 1] = bx1[stay]; nb[K:, 2] = by0[stay]; nb[K:, 3] = by1[stay]; nb[K:, 4] = cnt[stay]
        newnode = np.empty_like(node)
        for k in range(len(node)):
            p = node[k]
            newnode[k] = K - 1 - child_of[(p, q[k])] if is_split[p] else newpos_stay[p]
        bx0, bx1, by0, by1, cnt = nb[:, 0].copy(), nb[:, 1].copy(), nb[:, 2].copy(), nb[:, 3].copy(), nb[:, 4].copy()
        node = newnode

    finish = False
    while not finish:
        prev = len(bx0)
        q, c4, mx, my = quadrants()
        expandable = np.nonzero(cnt > 1)[0]
        rebuild(list(expandable), q, c4, mx, my)  # phase 1: list order
        size = len(bx0)
        nToExpand = int((cnt > 1).sum())
        if size >= N or size == prev:
            finish = True
        elif size + 3 * nToExpand > N:
            while not finish:
                prev = len(bx0)
                q, c4, mx, my = quadrants()
                E = np.nonzero(cnt > 1)[0]
                # sort (size asc, "pointer" asc) then walk from the back: size desc, later-created first.
                # later created == smaller list position.
                order = sorted(E, key=lambda p: (-cnt[p], p))
                chosen = []
                sz = prev
                for p in order:
                    chosen.append(p)
                    sz += int((c4[p] > 0).sum()) - 1
                    if sz >= N:
                        break
                rebuild(chosen, q, c4, mx, my)
                size = len(bx0)
                if size >= N or size == prev:
                    finish = True
    # best key per leaf: max response, first (lowest candidate index) wins ties
    out = np.full(len(bx0), -1, np.int64)
    best = np.full(len(bx0), -1, np.int64)
    for k in range(n):
        p = node[k]
        if score[k] > best[p]:
            best[p] = score[k]; out[p] = k
    return out


# --------------------------------------------------------------------------------- orientation (A6, P5)
def umax_table():
    hp = 15
    vmax = int(np.floor(hp * np.sqrt(2.0) / 2 + 1))
    vmin = int(np.ceil(hp * np.sqrt(2.0) / 2))
    um = [0] * (hp + 1)
    for v in range(vmax + 1):
        um[v] = int(np.rint(np.sqrt(float(hp * hp - v * v))))
    v0 = 0
    for v in range(hp, vmin - 1, -1):
        while um[v0] == um[v0 + 1]:
            v0 += 1
        um[v] = v0
        v0 += 1
    return um


def fast_atan2(y, x):
    f = np.float32
    scale = f(180.0 / np.pi)
    p1 = f(0.9997878412794807) * scale
    p3 = f(-0.3258083974640975) * scale
    p5 = f(0.1555786518463281) * scale
    p7 = f(-0.04432655554792128) * scale
    y = f(y); x = f(x)
    ax, ay = abs(x), abs(y)
    eps = f(2.220446049250313e-16)
    if ax >= ay:
        c = ay / (ax + eps); c2 = c * c
        a = (((p7 * c2 + p5) * c2 + p3) * c2 + p1) * c
    else:
        c = ax / (ay + eps); c2 = c * c
        a = f(90.0) - (((p7 * c2 + p5) * c2 + p3) * c2 + p1) * c
    if x < 0:
        a = f(180.0) - a
    if y < 0:
        a = f(360.0) - a
    return f(a)


def ic_angle(img, x, y):
    um = umax_table()
    I = img.astype(np.int64)
    m10 = m01 = 0
    for v in range(-15, 16):
        d = um[abs(v)]
        row = I[y + v, x - d:x + d + 1]
        u = np.arange(-d, d + 1)
        m10 += int((u * row).sum())
        m01 += v * int(row.sum())
    return fast_atan2(m01, m10)


# ------------------------------------------------------------------------------------- descriptor (A8)
def orb_descriptor(blurred, x, y, angle_deg, pattern):
    f = np.float32
    ang = f(angle_deg) * f(np.pi / f(180.0))
    a = f(np.cos(np.float64(ang)))  # correctly rounded float cos/sin (== glibc cosf on all tested inputs)
    b = f(np.sin(np.float64(ang)))
    pat = np.asarray(pattern, np.int64).reshape(512, 2)
    px = pat[:, 0].astype(np.float32); py = pat[:, 1].astype(np.float32)
    yy = cv_round((px * b).astype(np.float32) + (py * a).astype(np.float32))
    xx = cv_round((px * a).astype(np.float32) - (py * b).astype(np.float32))
    vals = blurred[y + yy, x + xx].astype(np.int64)
    bits = (vals[0::2] < vals[1::2]).astype(np.uint8)
    return np.packbits(bits.reshape(32, 8), axis=1, bitorder="little").reshape(32)


def hamming(a, b):
    return int(np.unpackbits(np.bitwise_xor(np.asarray(a, np.uint8), np.asarray(b, np.uint8))).sum())


# ------------------------------------------------------------------------------ whole extraction (A1-A9)
def extractor_tables(nfeatures, scale_factor, nlevels):
    """ORBextractor::ORBextractor (L/src/ORBextractor.cc:408-441): mvScaleFactor (float x the double member scaleFactor,
    L/include/ORBextractor.h:91), mvInvScaleFactor, mnFeaturesPerLevel (float arithmetic, cvRound)."""
    f = np.float32
    sf = [f(1.0)]
    for i in range(1, nlevels):
        sf.append(f(np.float64(sf[i - 1]) * np.float64(f(scale_factor))))
    inv = [f(1.0) / v for v in sf]
    factor = f(1.0) / f(scale_factor)
    nd = f(nfeatures) * (f(1) - factor) / (f(1) - f(np.float64(factor) ** np.float64(nlevels)))
    per, tot = [], 0
    for _ in range(nlevels - 1):
        per.append(int(np.rint(np.float64(nd))))
        tot += per[-1]
        nd = nd * factor
    per.append(max(nfeatures - tot, 0))
    return sf, inv, per


def ref_extract(img, nfeatures, scale_factor, nlevels, ini_th, min_th, pattern):
    """ORBextractor::operator() (L/src/ORBextractor.cc:983-1038) composed from the primitives above: ComputePyramid (:1040-1066, each
    level resized from the previous one), ComputeKeyPointsOctTree (:733-813), computeOrientation, blur + descriptors per level,
    coordinates scaled by mvScaleFactor[level].  Returns (list of (x, y, size, angle, response, octave), descriptors (n, 32))."""
    f = np.float32
    sf, inv, per = extractor_tables(nfeatures, scale_factor, nlevels)
    h0, w0 = img.shape
    pyr = []
    for l in range(nlevels):
        w = int(np.rint(np.float64(f(w0) * inv[l]))); h = int(np.rint(np.float64(f(h0) * inv[l])))
        pyr.append(img.copy() if l == 0 else resize_linear(pyr[l - 1], w, h))
    kps, descs = [], []
    for l in range(nlevels):
        im = pyr[l]
        h, w = im.shape
        x, y, s = fast_candidates(im, ini_th, min_th)
        if len(x) == 0:
            continue
        sel = distribute_octree(x, y, s, 16, w - 16, 16, h - 16, per[l])
        if len(sel) == 0:
            continue
        size = int(f(31) * sf[l])
        blurred = gaussian_blur7(im)
        for k in sel:
            px, py = int(x[k]) + 16, int(y[k]) + 16
            ang = ic_angle(im, px, py)
            descs.append(orb_descriptor(blurred, px, py, ang, pattern))
            fx, fy = f(px), f(py)
            if l != 0:
                fx, fy = fx * sf[l], fy * sf[l]
            kps.append((fx, fy, f(size), ang, f(s[k]), l))
    return kps, (np.stack(descs) if descs else np.zeros((0, 32), np.uint8))


# ============================================================================================ matchers (second reading)
# An INDEPENDENT restatement of the order-dependent matcher loops, written from the reference sources alone
# (L/src/ORBmatcher.cc:45-128, 161-273, 388-492, 1247-1383, 1506-1556 and L/src/Frame.cc:250-263, 341-410) as plain Python
# loops over lists -- structurally unlike oracle/orb_oracle.c (per-cell Python lists instead of a CSR grid, dict FeatureVectors,
# no query records inside the loops).  tests/test_oracle_cpu.py asserts that the C oracle agrees with it on real-image pairs and
# on a ties-heavy synthetic case; a misread quirk would have to be made twice, in two languages, to go unnoticed.
import math

FRAME_GRID_ROWS, FRAME_GRID_COLS = 48, 64      # L/include/Frame.h:36-37
TH_HIGH, TH_LOW, HISTO_LENGTH = 100, 50, 30    # L/src/ORBmatcher.cc:36-38
_f32 = np.float32
_POPCOUNT = np.array([bin(i).count("1") for i in range(256)], np.int32)


def c_round(x):
    """C round(): halves away from zero."""
    x = float(x)
    return int(math.floor(abs(x) + 0.5)) * (1 if x >= 0 else -1)


def descriptor_distance(a, b):
    """ORBmatcher::DescriptorDistance (:1542-1556): bit count of the XOR of 8 x 32 bits."""
    return int(_POPCOUNT[np.bitwise_xor(a, b)].sum())


class RefFrame:
    """The members of Frame the matcher reads, with AssignFeaturesToGrid (Frame.cc:250-263), PosInGrid (:399-410) and
    GetFeaturesInArea (:341-397)."""

    def __init__(self, keys, desc, min_x, max_x, min_y, max_y, u_right=None):
        self.N = len(keys)
        self.x = [_f32(v) for v in keys["x"]]; self.y = [_f32(v) for v in keys["y"]]
        self.octave = [int(v) for v in keys["octave"]]; self.angle = [_f32(v) for v in keys["angle"]]
        self.desc = np.asarray(desc, np.uint8)
        self.u_right = None if u_right is None else [_f32(v) for v in u_right]
        self.mnMinX, self.mnMaxX, self.mnMinY, self.mnMaxY = _f32(min_x), _f32(max_x), _f32(min_y), _f32(max_y)
        self.mfGridElementWidthInv = _f32(FRAME_GRID_COLS) / (self.mnMaxX - self.mnMinX)     # Frame.cc:107-110
        self.mfGridElementHeightInv = _f32(FRAME_GRID_ROWS) / (self.mnMaxY - self.mnMinY)
        self.mGrid = [[[] for _ in range(FRAME_GRID_ROWS)] for _ in range(FRAME_GRID_COLS)]
        for i in range(self.N):
            posX = c_round((self.x[i] - self.mnMinX) * self.mfGridElementWidthInv)
            posY = c_round((self.y[i] - self.mnMinY) * self.mfGridElementHeightInv)
            if posX < 0 or posX >= FRAME_GRID_COLS or posY < 0 or posY >= FRAME_GRID_ROWS:
                continue
            self.mGrid[posX][posY].append(i)

    def GetFeaturesInArea(self, x, y, r, minLevel=-1, maxLevel=-1):
        x, y, r = _f32(x), _f32(y), _f32(r)
        vIndices = []
        nMinCellX = max(0, int(math.floor((x - self.mnMinX - r) * self.mfGridElementWidthInv)))
        if nMinCellX >= FRAME_GRID_COLS:
            return vIndices
        nMaxCellX = min(FRAME_GRID_COLS - 1, int(math.ceil((x - self.mnMinX + r) * self.mfGridElementWidthInv)))
        if nMaxCellX < 0:
            return vIndices
        nMinCellY = max(0, int(math.floor((y - self.mnMinY - r) * self.mfGridElementHeightInv)))
        if nMinCellY >= FRAME_GRID_ROWS:
            return vIndices
        nMaxCellY = min(FRAME_GRID_ROWS - 1, int(math.ceil((y - self.mnMinY + r) * self.mfGridElementHeightInv)))
        if nMaxCellY < 0:
            return vIndices
        bCheckLevels = (minLevel > 0) or (maxLevel >= 0)
        for ix in range(nMinCellX, nMaxCellX + 1):
            for iy in range(nMinCellY, nMaxCellY + 1):
                for j in self.mGrid[ix][iy]:
                    if bCheckLevels:
                        if self.octave[j] < minLevel:
                            continue
                        if maxLevel >= 0 and self.octave[j] > maxLevel:
                            continue
                    distx = self.x[j] - x
                    disty = self.y[j] - y
                    if abs(distx) < r and abs(disty) < r:
                        vIndices.append(j)
        return vIndices


def compute_three_maxima(sizes):
    """ORBmatcher::ComputeThreeMaxima (:1506-1538) on the bin sizes."""
    max1 = max2 = max3 = 0
    ind1 = ind2 = ind3 = -1
    for i, s in enumerate(sizes):
        if s > max1:
            max3, max2, max1 = max2, max1, s
            ind3, ind2, ind1 = ind2, ind1, i
        elif s > max2:
            max3, max2 = max2, s
            ind3, ind2 = ind2, i
        elif s > max3:
            max3 = s
            ind3 = i
    if _f32(max2) < _f32(0.1) * _f32(max1):
        ind2 = ind3 = -1
    elif _f32(max3) < _f32(0.1) * _f32(max1):
        ind3 = -1
    return ind1, ind2, ind3


def _rot_bin(angle_a, angle_b):
    rot = _f32(angle_a) - _f32(angle_b)
    if rot < 0.0:
        rot = rot + _f32(360.0)
    b = c_round(rot * (_f32(1.0) / _f32(HISTO_LENGTH)))
    return 0 if b == HISTO_LENGTH else b


def ref_search_by_projection_points(F, points, nnratio, has_observed_point):
    """SearchByProjection(Frame&, vector<MapPoint*>&, th) (:45-128).  points: list of dicts with the MapPoint fields the loop
    reads after the caller's part -- skip (= !mbTrackInView || isBad()), proj_x, proj_y, proj_xr, level, radius
    (= r * mvScaleFactors[level]), observed (Observations() > 0), desc.  has_observed_point[idx]: F.mvpMapPoints[idx] exists
    with Observations() > 0 before the call.  Returns (nmatches, point index per keypoint or -1, and per keypoint whether
    F.mvpMapPoints[idx] holds a point with Observations() > 0 after the call)."""
    mvpMapPoints = [None] * F.N                      # index of the point written by this call
    pre = list(has_observed_point)
    nmatches = 0
    for iMP, p in enumerate(points):
        if p["skip"]:
            continue
        lvl = p["level"]
        vIndices = F.GetFeaturesInArea(p["proj_x"], p["proj_y"], p["radius"], lvl - 1, lvl)
        if not vIndices:
            continue
        bestDist, bestLevel, bestDist2, bestLevel2, bestIdx = 256, -1, 256, -1, -1
        for idx in vIndices:
            cur = mvpMapPoints[idx]
            if (cur is not None and points[cur]["observed"]) or (cur is None and pre[idx]):
                continue
            if F.u_right is not None and F.u_right[idx] > 0:
                er = abs(_f32(p["proj_xr"]) - F.u_right[idx])
                if er > _f32(p["radius"]):
                    continue
            dist = descriptor_distance(p["desc"], F.desc[idx])
            if dist < bestDist:
                bestDist2, bestDist = bestDist, dist
                bestLevel2, bestLevel = bestLevel, F.octave[idx]
                bestIdx = idx
            elif dist < bestDist2:
                bestLevel2 = F.octave[idx]
                bestDist2 = dist
        if bestDist <= TH_HIGH:
            if bestLevel == bestLevel2 and _f32(bestDist) > _f32(nnratio) * _f32(bestDist2):
                continue
            mvpMapPoints[bestIdx] = iMP
            pre[bestIdx] = False
            nmatches += 1
    after = [bool(points[c]["observed"]) if c is not None else bool(pre[i]) for i, c in enumerate(mvpMapPoints)]
    return nmatches, [(-1 if v is None else v) for v in mvpMapPoints], after


def ref_search_by_projection_frame(F, last, check_orientation, has_observed_point):
    """SearchByProjection(Frame& cur, const Frame& last, th, bMono) from the window search on (:1289-1383).  last: list of
    dicts per LastFrame keypoint -- skip (no point / outlier / behind the camera / outside the image), u, v, ur
    (= u - mbf * invzc), radius, min_level, max_level (the bForward / bBackward / default choice made by the caller),
    observed, angle (LastFrame.mvKeysUn[i].angle), desc.  Returns (nmatches, last-frame index per keypoint or -1, and per
    keypoint whether CurrentFrame.mvpMapPoints[idx] holds a point with Observations() > 0 after the call -- a slot the rotation
    check cleared is NULL again)."""
    mvpMapPoints = [None] * F.N
    pre = list(has_observed_point)
    rotHist = [[] for _ in range(HISTO_LENGTH)]
    nmatches = 0
    for i, p in enumerate(last):
        if p["skip"]:
            continue
        vIndices2 = F.GetFeaturesInArea(p["u"], p["v"], p["radius"], p["min_level"], p["max_level"])
        if not vIndices2:
            continue
        bestDist, bestIdx2 = 256, -1
        for i2 in vIndices2:
            cur = mvpMapPoints[i2]
            if (cur is not None and cur >= 0 and last[cur]["observed"]) or (cur is None and pre[i2]):
                continue
            if F.u_right is not None and F.u_right[i2] > 0:
                er = abs(_f32(p["ur"]) - F.u_right[i2])
                if er > _f32(p["radius"]):
                    continue
            dist = descriptor_distance(p["desc"], F.desc[i2])
            if dist < bestDist:
                bestDist, bestIdx2 = dist, i2
        if bestDist <= TH_HIGH:
            mvpMapPoints[bestIdx2] = i
            pre[bestIdx2] = False
            nmatches += 1
            if check_orientation:
                rotHist[_rot_bin(p["angle"], F.angle[bestIdx2])].append(bestIdx2)
    if check_orientation:
        ind = compute_three_maxima([len(h) for h in rotHist])
        for b in range(HISTO_LENGTH):
            if b in ind:
                continue
            for idx in rotHist[b]:
                mvpMapPoints[idx] = -1          # static_cast<MapPoint*>(NULL)
                nmatches -= 1
    after = [(c >= 0 and bool(last[c]["observed"])) if c is not None else bool(pre[i]) for i, c in enumerate(mvpMapPoints)]
    return nmatches, [(-1 if v is None else v) for v in mvpMapPoints], after


def ref_search_by_bow(descKF, angleKF, validKF, featvecKF, descF, angleF, featvecF, nnratio, check_orientation):
    """SearchByBoW(KeyFrame*, Frame&, vpMapPointMatches) (:161-273).  featvec*: {node id: [feature indices]} (DBoW2::FeatureVector
    is a std::map: ascending ids); validKF[i]: the keyframe's map point i exists and is not bad.  Returns (nmatches,
    keyframe feature index per frame feature or -1)."""
    matches = [-1] * len(descF)
    rotHist = [[] for _ in range(HISTO_LENGTH)]
    nmatches = 0
    idsKF, idsF = sorted(featvecKF), sorted(featvecF)
    a = b = 0
    while a < len(idsKF) and b < len(idsF):
        if idsKF[a] == idsF[b]:
            for realIdxKF in featvecKF[idsKF[a]]:
                if not validKF[realIdxKF]:
                    continue
                bestDist1, bestIdxF, bestDist2 = 256, -1, 256
                for realIdxF in featvecF[idsF[b]]:
                    if matches[realIdxF] >= 0:
                        continue
                    dist = descriptor_distance(descKF[realIdxKF], descF[realIdxF])
                    if dist < bestDist1:
                        bestDist2, bestDist1, bestIdxF = bestDist1, dist, realIdxF
                    elif dist < bestDist2:
                        bestDist2 = dist
                if bestDist1 <= TH_LOW and _f32(bestDist1) < _f32(nnratio) * _f32(bestDist2):
                    matches[bestIdxF] = realIdxKF
                    if check_orientation:
                        rotHist[_rot_bin(angleKF[realIdxKF], angleF[bestIdxF])].append(bestIdxF)
                    nmatches += 1
            a += 1
            b += 1
        elif idsKF[a] < idsF[b]:
            while a < len(idsKF) and idsKF[a] < idsF[b]:      # lower_bound
                a += 1
        else:
            while b < len(idsF) and idsF[b] < idsKF[a]:
                b += 1
    if check_orientation:
        ind = compute_three_maxima([len(h) for h in rotHist])
        for i in range(HISTO_LENGTH):
            if i in ind:
                continue
            for j in rotHist[i]:
                matches[j] = -1
                nmatches -= 1
    return nmatches, matches


def _featvec_walk(fvA, fvB):
    """The lock-step walk over two DBoW2::FeatureVector maps (ascending node ids, lower_bound on a mismatch): yields the lists of
    feature indices of every node both hold."""
    idsA, idsB = sorted(fvA), sorted(fvB)
    a = b = 0
    while a < len(idsA) and b < len(idsB):
        if idsA[a] == idsB[b]:
            yield fvA[idsA[a]], fvB[idsB[b]]
            a += 1
            b += 1
        elif idsA[a] < idsB[b]:
            while a < len(idsA) and idsA[a] < idsB[b]:
                a += 1
        else:
            while b < len(idsB) and idsB[b] < idsA[a]:
                b += 1


def ref_search_by_bow_kf(desc1, angle1, valid1, featvec1, desc2, angle2, valid2, featvec2, nnratio, check_orientation):
    """SearchByBoW(KeyFrame* pKF1, KeyFrame* pKF2, vpMatches12) (:494-612).  valid*[i]: the keyframe's map point i exists and is not
    bad.  Returns (nmatches, index in keyframe 2 per feature of keyframe 1 or -1)."""
    vpMatches12 = [-1] * len(desc1)
    vbMatched2 = [False] * len(desc2)
    rotHist = [[] for _ in range(HISTO_LENGTH)]
    nmatches = 0
    for node1, node2 in _featvec_walk(featvec1, featvec2):
        for idx1 in node1:
            if not valid1[idx1]:
                continue
            bestDist1, bestIdx2, bestDist2 = 256, -1, 256
            for idx2 in node2:
                if vbMatched2[idx2] or not valid2[idx2]:
                    continue
                dist = descriptor_distance(desc1[idx1], desc2[idx2])
                if dist < bestDist1:
                    bestDist2, bestDist1, bestIdx2 = bestDist1, dist, idx2
                elif dist < bestDist2:
                    bestDist2 = dist
            if bestDist1 < TH_LOW:
                if _f32(bestDist1) < _f32(nnratio) * _f32(bestDist2):
                    vpMatches12[idx1] = bestIdx2
                    vbMatched2[bestIdx2] = True
                    if check_orientation:
                        rotHist[_rot_bin(angle1[idx1], angle2[bestIdx2])].append(idx1)
                    nmatches += 1
    if check_orientation:
        ind = compute_three_maxima([len(h) for h in rotHist])
        for i in range(HISTO_LENGTH):
            if i in ind:
                continue
            for j in rotHist[i]:
                vpMatches12[j] = -1
                nmatches -= 1
    return nmatches, vpMatches12


def check_dist_epipolar_line(x1, y1, x2, y2, octave2, F12, level_sigma2):
    """ORBmatcher::CheckDistEpipolarLine (:137-159): float products and sums left to right, the last comparison in double (3.84
    is a double constant)."""
    F = [[_f32(F12[r][c]) for c in range(3)] for r in range(3)]
    x1, y1, x2, y2 = _f32(x1), _f32(y1), _f32(x2), _f32(y2)
    a = x1 * F[0][0] + y1 * F[1][0] + F[2][0]
    b = x1 * F[0][1] + y1 * F[1][1] + F[2][1]
    c = x1 * F[0][2] + y1 * F[1][2] + F[2][2]
    num = a * x2 + b * y2 + c
    den = a * a + b * b
    if den == 0:
        return False
    dsqr = num * num / den
    return float(dsqr) < 3.84 * float(_f32(level_sigma2[octave2]))


def ref_search_for_triangulation(keys1, desc1, u_right1, has_mp1, featvec1, keys2, desc2, u_right2, has_mp2, featvec2, F12, ex, ey,
                                 scale_factors, level_sigma2, only_stereo, check_orientation):
    """SearchForTriangulation(pKF1, pKF2, F12, vMatchedPairs, bOnlyStereo) (:614-764) after the epipole (ex, ey) has been formed.
    u_right*: mvuRight or None (monocular: every value -1).  vbMatched2 is read but never set by the reference -- kept so.  Returns
    (nmatches, vMatches12)."""
    n1, n2 = len(keys1), len(keys2)
    ur1 = [_f32(-1)] * n1 if u_right1 is None else [_f32(v) for v in u_right1]
    ur2 = [_f32(-1)] * n2 if u_right2 is None else [_f32(v) for v in u_right2]
    ex, ey = _f32(ex), _f32(ey)
    vbMatched2 = [False] * n2
    vMatches12 = [-1] * n1
    rotHist = [[] for _ in range(HISTO_LENGTH)]
    nmatches = 0
    for node1, node2 in _featvec_walk(featvec1, featvec2):
        for idx1 in node1:
            if has_mp1[idx1]:
                continue
            bStereo1 = ur1[idx1] >= 0
            if only_stereo and not bStereo1:
                continue
            bestDist, bestIdx2 = TH_LOW, -1
            for idx2 in node2:
                if vbMatched2[idx2] or has_mp2[idx2]:
                    continue
                bStereo2 = ur2[idx2] >= 0
                if only_stereo and not bStereo2:
                    continue
                dist = descriptor_distance(desc1[idx1], desc2[idx2])
                if dist > TH_LOW or dist > bestDist:
                    continue
                x2, y2, o2 = _f32(keys2["x"][idx2]), _f32(keys2["y"][idx2]), int(keys2["octave"][idx2])
                if not bStereo1 and not bStereo2:
                    distex = ex - x2
                    distey = ey - y2
                    if distex * distex + distey * distey < _f32(100) * _f32(scale_factors[o2]):
                        continue
                if check_dist_epipolar_line(keys1["x"][idx1], keys1["y"][idx1], x2, y2, o2, F12, level_sigma2):
                    bestIdx2, bestDist = idx2, dist
            if bestIdx2 >= 0:
                vMatches12[idx1] = bestIdx2
                nmatches += 1
                if check_orientation:
                    rotHist[_rot_bin(keys1["angle"][idx1], keys2["angle"][bestIdx2])].append(idx1)
    if check_orientation:
        ind = compute_three_maxima([len(h) for h in rotHist])
        for i in range(HISTO_LENGTH):
            if i in ind:
                continue
            for j in rotHist[i]:
                vMatches12[j] = -1
                nmatches -= 1
    return nmatches, vMatches12


def ref_search_for_initialization(keys1, desc1, F2, prev_matched, window, nnratio, check_orientation):
    """SearchForInitialization(F1, F2, vbPrevMatched, vnMatches12, windowSize) (:388-492).  Returns (nmatches, vnMatches12,
    updated vbPrevMatched)."""
    INT_MAX = 2 ** 31 - 1
    n1 = len(keys1)
    prev = [(_f32(p[0]), _f32(p[1])) for p in prev_matched]
    vnMatches12 = [-1] * n1
    rotHist = [[] for _ in range(HISTO_LENGTH)]
    vMatchedDistance = [INT_MAX] * F2.N
    vnMatches21 = [-1] * F2.N
    nmatches = 0
    for i1 in range(n1):
        level1 = int(keys1["octave"][i1])
        if level1 > 0:
            continue
        vIndices2 = F2.GetFeaturesInArea(prev[i1][0], prev[i1][1], window, level1, level1)
        if not vIndices2:
            continue
        bestDist, bestDist2, bestIdx2 = INT_MAX, INT_MAX, -1
        for i2 in vIndices2:
            dist = descriptor_distance(desc1[i1], F2.desc[i2])
            if vMatchedDistance[i2] <= dist:
                continue
            if dist < bestDist:
                bestDist2, bestDist, bestIdx2 = bestDist, dist, i2
            elif dist < bestDist2:
                bestDist2 = dist
        if bestDist <= TH_LOW and _f32(bestDist) < _f32(bestDist2) * _f32(nnratio):
            if vnMatches21[bestIdx2] >= 0:
                vnMatches12[vnMatches21[bestIdx2]] = -1
                nmatches -= 1
            vnMatches12[i1] = bestIdx2
            vnMatches21[bestIdx2] = i1
            vMatchedDistance[bestIdx2] = bestDist
            nmatches += 1
            if check_orientation:
                rotHist[_rot_bin(keys1["angle"][i1], F2.angle[bestIdx2])].append(i1)
    if check_orientation:
        ind = compute_three_maxima([len(h) for h in rotHist])
        for i in range(HISTO_LENGTH):
            if i in ind:
                continue
            for idx1 in rotHist[i]:
                if vnMatches12[idx1] >= 0:
                    vnMatches12[idx1] = -1
                    nmatches -= 1
    for i1 in range(n1):
        if vnMatches12[i1] >= 0:
            prev[i1] = (F2.x[vnMatches12[i1]], F2.y[vnMatches12[i1]])
    return nmatches, vnMatches12, prev


def ref_compute_stereo_matches(keysL, descL, keysR, descR, pyrL, pyrR, scale_factors, inv_scale_factors, mbf, mb):
    """Frame::ComputeStereoMatches (L/src/Frame.cc:477-646), second reading (per-row Python lists, numpy windows).  pyrL / pyrR:
    lists of the level images of the two extractors' mvImagePyramid.  Returns (mvuRight, mvDepth) as float32 arrays.  Where the
    reference would index out of range (rows outside the image, windows leaving a level) the keypoint gets no match."""
    N = len(keysL)
    mvuRight = np.full(N, -1.0, np.float32); mvDepth = np.full(N, -1.0, np.float32)
    thOrbDist = (TH_HIGH + TH_LOW) // 2
    nRows = pyrL[0].shape[0]
    vRowIndices = [[] for _ in range(nRows)]
    for iR in range(len(keysR)):
        kpY = _f32(keysR["y"][iR])
        r = _f32(2.0) * _f32(scale_factors[int(keysR["octave"][iR])])
        maxr = int(math.ceil(kpY + r)); minr = int(math.floor(kpY - r))
        for yi in range(minr, maxr + 1):
            if 0 <= yi < nRows:
                vRowIndices[yi].append(iR)
    minZ = _f32(mb); minD = _f32(0); maxD = _f32(mbf) / minZ
    vDistIdx = []
    descR = np.asarray(descR, np.uint8)
    for iL in range(N):
        levelL = int(keysL["octave"][iL]); vL = _f32(keysL["y"][iL]); uL = _f32(keysL["x"][iL])
        row = int(vL)
        if row < 0 or row >= nRows:
            continue
        vCandidates = vRowIndices[row]
        if not vCandidates:
            continue
        minU = uL - maxD; maxU = uL - minD
        if maxU < 0:
            continue
        bestDist = TH_HIGH; bestIdxR = 0
        for iR in vCandidates:
            oR = int(keysR["octave"][iR])
            if oR < levelL - 1 or oR > levelL + 1:
                continue
            uR = _f32(keysR["x"][iR])
            if uR >= minU and uR <= maxU:
                dist = descriptor_distance(descL[iL], descR[iR])
                if dist < bestDist:
                    bestDist = dist; bestIdxR = iR
        if not (bestDist < thOrbDist):
            continue
        uR0 = _f32(keysR["x"][bestIdxR])
        scaleFactor = _f32(inv_scale_factors[levelL])
        scaleduL = _f32(c_round(uL * scaleFactor)); scaledvL = _f32(c_round(vL * scaleFactor)); scaleduR0 = _f32(c_round(uR0 * scaleFactor))
        w = 5; L = 5
        imL, imR = pyrL[levelL], pyrR[levelL]
        y0, x0 = int(scaledvL - w), int(scaleduL - w)
        if y0 < 0 or y0 + 2 * w + 1 > imL.shape[0] or x0 < 0 or x0 + 2 * w + 1 > imL.shape[1]:
            continue      # cv::Mat::rowRange / colRange would throw
        IL = imL[y0:y0 + 2 * w + 1, x0:x0 + 2 * w + 1].astype(np.float32)
        IL = IL - IL[w, w]
        iniu = scaleduR0 + _f32(L) - _f32(w); endu = scaleduR0 + _f32(L) + _f32(w) + _f32(1)
        if iniu < 0 or endu >= imR.shape[1]:
            continue
        xr = int(scaleduR0)
        if xr - L - w < 0 or xr + L + w + 1 > imR.shape[1] or y0 + 2 * w + 1 > imR.shape[0]:
            continue
        bestD = 2 ** 31 - 1; bestincR = 0
        vDists = [0.0] * (2 * L + 1)
        for incR in range(-L, L + 1):
            IR = imR[y0:y0 + 2 * w + 1, xr + incR - w:xr + incR + w + 1].astype(np.float32)
            IR = IR - IR[w, w]
            dist = _f32(np.abs(IL - IR).sum(dtype=np.float64))     # NORM_L1 of integer-valued floats: exact
            if dist < bestD:
                bestD = int(dist); bestincR = incR
            vDists[L + incR] = dist
        if bestincR == -L or bestincR == L:
            continue
        dist1, dist2, dist3 = _f32(vDists[L + bestincR - 1]), _f32(vDists[L + bestincR]), _f32(vDists[L + bestincR + 1])
        den = _f32(2.0) * (dist1 + dist3 - _f32(2.0) * dist2)
        with np.errstate(divide="ignore", invalid="ignore"):
            deltaR = (dist1 - dist3) / den
        if deltaR < -1 or deltaR > 1:
            continue
        bestuR = _f32(scale_factors[levelL]) * (scaleduR0 + _f32(bestincR) + deltaR)
        disparity = uL - bestuR
        if disparity >= minD and disparity < maxD:
            if disparity <= 0:
                disparity = _f32(0.01)
                bestuR = _f32(float(uL) - 0.01)
            mvDepth[iL] = _f32(mbf) / disparity
            mvuRight[iL] = bestuR
            vDistIdx.append((bestD, iL))
    if vDistIdx:
        vDistIdx.sort()
        median = _f32(vDistIdx[len(vDistIdx) // 2][0])
        thDist = _f32(1.5) * _f32(1.4) * median
        for i in range(len(vDistIdx) - 1, -1, -1):
            if vDistIdx[i][0] < thDist:
                break
            mvuRight[vDistIdx[i][1]] = -1; mvDepth[vDistIdx[i][1]] = -1
    return mvuRight, mvDepth


# ------------------------------------------------------------------------------------------------------------------------------
# Second reading of the keyframe-rate projection searches (Fuse x2, SearchBySim3, SearchByProjection(KF, Scw) and
# SearchByProjection(Frame, KF)).  cv::Mat arithmetic as DESIGN.md's primitive list has it (the same assumptions as the C
# oracle -- this reading checks the loops, gates and orders, not OpenCV's kernels): A * x + b on 3x3 / 3x1 CV_32F = float dot
# products, the sum with b in double, rounded once; cv::norm and Mat::dot accumulate in double.  logf comes from the C library
# (MapPoint::PredictScale calls std::log on a float).
import ctypes as _ct
import ctypes.util as _ctu

_libm = _ct.CDLL(_ctu.find_library("m") or "libm.so.6")
_libm.logf.restype = _ct.c_float
_libm.logf.argtypes = [_ct.c_float]


def _gemm3(R, x, t):
    out = []
    for r in range(3):
        d = _f32(R[3 * r]) * _f32(x[0]) + _f32(R[3 * r + 1]) * _f32(x[1]) + _f32(R[3 * r + 2]) * _f32(x[2])
        out.append(_f32(float(d) * 1.0 + float(_f32(t[r])) * 1.0))
    return out


def _norm3(v):
    s = 0.0
    for e in v:
        s += float(e) * float(e)
    return _f32(math.sqrt(s))


def _dot3(a, b):
    s = 0.0
    for x, y in zip(a, b):
        s += float(x) * float(y)
    return s


def predict_scale(max_distance, current_dist, log_scale_factor, n_levels):
    """MapPoint::PredictScale (MapPoint.cc:393-423): ceil(logf(ratio) / mfLogScaleFactor) in float, clamped."""
    ratio = _f32(max_distance) / _f32(current_dist)
    n = int(math.ceil(_f32(_libm.logf(float(ratio))) / _f32(log_scale_factor)))
    return 0 if n < 0 else (n_levels - 1 if n >= n_levels else n)


def _kf_in_image(cam, u, v):
    """KeyFrame::IsInImage (KeyFrame.cc:569-571)."""
    return u >= cam["min_x"] and u < cam["max_x"] and v >= cam["min_y"] and v < cam["max_y"]


def _pinhole(cam, pc, double_invz):
    """u, v of a camera-frame point.  `1 / z` is a float division in Fuse and SearchByProjection(KF, Scw) (:797, :317), `1.0 / z` a
    double one in Fuse(Sim3) and SearchBySim3 (:945, :1083)."""
    invz = _f32(1.0 / float(pc[2])) if double_invz else _f32(1) / pc[2]
    x = pc[0] * invz
    y = pc[1] * invz
    return cam["fx"] * x + cam["cx"], cam["fy"] * y + cam["cy"], invz


def _viewing_gates(cam, p):
    """dist3D inside [0.8 min, 1.2 max] and PO . Pn >= 0.5 dist3D; returns dist3D or None."""
    PO = [_f32(p["pos"][k]) - _f32(cam["Ow"][k]) for k in range(3)]
    dist3D = _norm3(PO)
    if dist3D < _f32(0.8) * _f32(p["min_distance"]) or dist3D > _f32(1.2) * _f32(p["max_distance"]):
        return None
    if _dot3(PO, [_f32(v) for v in p["normal"]]) < 0.5 * float(dist3D):
        return None
    return dist3D


def _cam_f32(cam):
    c = {k: (_f32(v) if np.ndim(v) == 0 and k != "n_levels" else v) for k, v in cam.items()}
    c["n_levels"] = int(cam["n_levels"])
    return c


def ref_fuse(KF, inv_level_sigma2, cam, points):
    """Fuse(KeyFrame*, vpMapPoints, th) (:766-912) up to the map update: per point the keypoint the reference acts on (bestIdx when
    bestDist <= TH_LOW is the caller's test; here (bestIdx, bestDist) or (-1, 256)).  KF: RefFrame of the keyframe (the grid is the
    Frame's); points[i]: pos, normal, min_distance, max_distance, skip (NULL / isBad / IsInKeyFrame), desc."""
    cam = _cam_f32(cam)
    out = []
    for p in points:
        res = (-1, 256)
        out.append(res)
        if p["skip"]:
            continue
        pc = _gemm3(cam["R"], p["pos"], cam["t"])
        if pc[2] < _f32(0.0):
            continue
        u, v, invz = _pinhole(cam, pc, False)
        if not _kf_in_image(cam, u, v):
            continue
        ur = u - cam["mbf"] * invz
        dist3D = _viewing_gates(cam, p)
        if dist3D is None:
            continue
        lvl = predict_scale(p["max_distance"], dist3D, cam["log_scale_factor"], cam["n_levels"])
        radius = cam["th"] * _f32(cam["scale_factors"][lvl])
        vIndices = KF.GetFeaturesInArea(u, v, radius)
        if not vIndices:
            continue
        bestDist, bestIdx = 256, -1
        for idx in vIndices:
            kpLevel = KF.octave[idx]
            if kpLevel < lvl - 1 or kpLevel > lvl:
                continue
            ex = u - KF.x[idx]
            ey = v - KF.y[idx]
            if KF.u_right is not None and KF.u_right[idx] >= 0:
                er = ur - KF.u_right[idx]
                e2 = ex * ex + ey * ey + er * er
                if float(e2 * _f32(inv_level_sigma2[kpLevel])) > 7.8:
                    continue
            else:
                e2 = ex * ex + ey * ey
                if float(e2 * _f32(inv_level_sigma2[kpLevel])) > 5.99:
                    continue
            dist = descriptor_distance(p["desc"], KF.desc[idx])
            if dist < bestDist:
                bestDist, bestIdx = dist, idx
        out[-1] = (bestIdx, bestDist if bestIdx >= 0 else 256)
    return out


def ref_fuse_sim3(KF, cam, points):
    """Fuse(KeyFrame*, Scw, vpPoints, th, vpReplacePoint) (:914-1041) after Scw has been decomposed (cam: R, t, Ow), up to the map
    update.  skip = isBad() or already a point of the keyframe."""
    cam = _cam_f32(cam)
    out = []
    for p in points:
        out.append((-1, 256))
        if p["skip"]:
            continue
        pc = _gemm3(cam["R"], p["pos"], cam["t"])
        if pc[2] < _f32(0.0):
            continue
        u, v, _ = _pinhole(cam, pc, True)
        if not _kf_in_image(cam, u, v):
            continue
        dist3D = _viewing_gates(cam, p)
        if dist3D is None:
            continue
        lvl = predict_scale(p["max_distance"], dist3D, cam["log_scale_factor"], cam["n_levels"])
        vIndices = KF.GetFeaturesInArea(u, v, cam["th"] * _f32(cam["scale_factors"][lvl]))
        if not vIndices:
            continue
        bestDist, bestIdx = 2 ** 31 - 1, -1
        for idx in vIndices:
            if KF.octave[idx] < lvl - 1 or KF.octave[idx] > lvl:
                continue
            dist = descriptor_distance(p["desc"], KF.desc[idx])
            if dist < bestDist:
                bestDist, bestIdx = dist, idx
        out[-1] = (bestIdx, bestDist if bestIdx >= 0 else 256)
    return out


def ref_search_by_sim3_dir(KF, cam, points):
    """One direction of SearchBySim3 (:1063-1146; :1149-1222 with the roles swapped): cam R/t = the source keyframe's pose, R2/t2 =
    sR21 / t21; no viewing-angle gate, the distance is that of the point in the target camera."""
    cam = _cam_f32(cam)
    out = []
    for p in points:
        out.append((-1, 256))
        if p["skip"]:
            continue
        pc1 = _gemm3(cam["R"], p["pos"], cam["t"])
        pc2 = _gemm3(cam["R2"], pc1, cam["t2"])
        if float(pc2[2]) < 0.0:
            continue
        u, v, _ = _pinhole(cam, pc2, True)
        if not _kf_in_image(cam, u, v):
            continue
        dist3D = _norm3(pc2)
        if dist3D < _f32(0.8) * _f32(p["min_distance"]) or dist3D > _f32(1.2) * _f32(p["max_distance"]):
            continue
        lvl = predict_scale(p["max_distance"], dist3D, cam["log_scale_factor"], cam["n_levels"])
        vIndices = KF.GetFeaturesInArea(u, v, cam["th"] * _f32(cam["scale_factors"][lvl]))
        if not vIndices:
            continue
        bestDist, bestIdx = 2 ** 31 - 1, -1
        for idx in vIndices:
            if KF.octave[idx] < lvl - 1 or KF.octave[idx] > lvl:
                continue
            dist = descriptor_distance(p["desc"], KF.desc[idx])
            if dist < bestDist:
                bestDist, bestIdx = dist, idx
        out[-1] = (bestIdx, bestDist if bestIdx >= 0 else 256)
    return out


def ref_search_by_projection_kf_scw(KF, cam, points, matched, th_low=TH_LOW):
    """SearchByProjection(KeyFrame*, Scw, vpPoints, vpMatched, th) (:275-386) after the decomposition of Scw.  matched[idx]:
    vpMatched[idx] != NULL on entry.  Returns (nmatches, keypoint index written per point or -1, updated matched)."""
    cam = _cam_f32(cam)
    vpMatched = [bool(m) for m in matched]
    written = []
    nmatches = 0
    for p in points:
        written.append(-1)
        if p["skip"]:
            continue
        pc = _gemm3(cam["R"], p["pos"], cam["t"])
        if float(pc[2]) < 0.0:
            continue
        u, v, _ = _pinhole(cam, pc, False)
        if not _kf_in_image(cam, u, v):
            continue
        dist = _viewing_gates(cam, p)
        if dist is None:
            continue
        lvl = predict_scale(p["max_distance"], dist, cam["log_scale_factor"], cam["n_levels"])
        vIndices = KF.GetFeaturesInArea(u, v, cam["th"] * _f32(cam["scale_factors"][lvl]))
        if not vIndices:
            continue
        bestDist, bestIdx = 256, -1
        for idx in vIndices:
            if vpMatched[idx]:
                continue
            if KF.octave[idx] < lvl - 1 or KF.octave[idx] > lvl:
                continue
            d = descriptor_distance(p["desc"], KF.desc[idx])
            if d < bestDist:
                bestDist, bestIdx = d, idx
        if bestDist <= th_low:
            vpMatched[bestIdx] = True
            written[-1] = bestIdx
            nmatches += 1
    return nmatches, written, vpMatched


def ref_search_by_projection_frame_kf(F, cam, points, has_point, orb_dist, check_orientation):
    """SearchByProjection(Frame& CurrentFrame, KeyFrame*, sAlreadyFound, th, ORBdist) (:1385-1504).  points[i] = the keyframe's map
    point i: skip (NULL / isBad / in sAlreadyFound), pos, min/max_distance, desc, angle (pKF->mvKeysUn[i].angle).  has_point[i2]:
    CurrentFrame.mvpMapPoints[i2] != NULL on entry.  Returns (nmatches, point index per frame keypoint or -1, updated has_point)."""
    cam = _cam_f32(cam)
    mvp = [bool(m) for m in has_point]
    who = [-1] * F.N
    rotHist = [[] for _ in range(HISTO_LENGTH)]
    nmatches = 0
    for i, p in enumerate(points):
        if p["skip"]:
            continue
        xc3 = _gemm3(cam["R"], p["pos"], cam["t"])
        xc, yc = xc3[0], xc3[1]
        invzc = _f32(1.0 / float(xc3[2]))
        u = cam["fx"] * xc * invzc + cam["cx"]
        v = cam["fy"] * yc * invzc + cam["cy"]
        if u < cam["min_x"] or u > cam["max_x"]:
            continue
        if v < cam["min_y"] or v > cam["max_y"]:
            continue
        PO = [_f32(p["pos"][k]) - _f32(cam["Ow"][k]) for k in range(3)]
        dist3D = _norm3(PO)
        if dist3D < _f32(0.8) * _f32(p["min_distance"]) or dist3D > _f32(1.2) * _f32(p["max_distance"]):
            continue
        lvl = predict_scale(p["max_distance"], dist3D, cam["log_scale_factor"], cam["n_levels"])
        radius = cam["th"] * _f32(cam["scale_factors"][lvl])
        vIndices2 = F.GetFeaturesInArea(u, v, radius, lvl - 1, lvl + 1)
        if not vIndices2:
            continue
        bestDist, bestIdx2 = 256, -1
        for i2 in vIndices2:
            if mvp[i2]:
                continue
            dist = descriptor_distance(p["desc"], F.desc[i2])
            if dist < bestDist:
                bestDist, bestIdx2 = dist, i2
        if bestDist <= orb_dist:
            mvp[bestIdx2] = True
            who[bestIdx2] = i
            nmatches += 1
            if check_orientation:
                rotHist[_rot_bin(p["angle"], F.angle[bestIdx2])].append(bestIdx2)
    if check_orientation:
        ind = compute_three_maxima([len(h) for h in rotHist])
        for b in range(HISTO_LENGTH):
            if b in ind:
                continue
            for j in rotHist[b]:
                mvp[j] = False
                who[j] = -1
                nmatches -= 1
    return nmatches, who, mvp


# ------------------------------------------------------------------------------------------------------------------------------
# Second reading of Frame::ComputeBoW's work: DBoW2 TemplatedVocabulary::transform (ThirdParty/DBoW2/DBoW2-local/include/DBoW2/
# TemplatedVocabulary.h:1125-1192 and :1216-1257) on a tree as ORBVocabulary::loadFromTextFile builds it (ORBVocabulary.cc:79-123:
# node ids in file order, children appended in that order, word ids numbering the nodes flagged leaf in that order).
def ref_vocab_transform(L, parent, leaf_flag, node_desc, node_weight, features, levelsup, scoring=0, weighting=0):
    """Returns (BowVector as {word id: value}, FeatureVector as {node id: [feature indices]}).  scoring: 0 L1, 1 L2, 2 chi-square,
    3 KL, 4 Bhattacharyya, 5 dot product (ScoringObject.h:73-88); weighting: 0 TF_IDF, 1 TF, 2 IDF, 3 BINARY."""
    n = len(parent)
    children = [[] for _ in range(n)]
    word_id = [-1] * n
    n_words = 0
    for nid in range(1, n):
        children[int(parent[nid])].append(nid)
        if leaf_flag[nid]:
            word_id[nid] = n_words
            n_words += 1
    bow, fv = {}, {}
    if n <= 1:                                       # empty(): no words
        return bow, fv
    for i_feature, f in enumerate(features):
        nid_level = L - levelsup
        nid = 0   # root when nid_level <= 0 (:1225); ALSO when the descent ends above nid_level: the reference then reads an
                  # uninitialised NodeId (:1149, :1249 never reached) -- undefined there, defined as the root here and in the oracle
        final_id, current_level = 0, 0
        while True:
            current_level += 1
            nodes = children[final_id]
            final_id = nodes[0]
            best_d = float(descriptor_distance(f, node_desc[final_id]))
            for c in nodes[1:]:
                d = float(descriptor_distance(f, node_desc[c]))
                if d < best_d:
                    best_d, final_id = d, c
            if current_level == nid_level:
                nid = final_id
            if not children[final_id]:               # Node::isLeaf() = children.empty()
                break
        w = float(node_weight[final_id])
        if w > 0:
            wid = word_id[final_id]
            if weighting in (0, 1):
                bow[wid] = bow.get(wid, 0.0) + w     # addWeight
            elif wid not in bow:
                bow[wid] = w                         # addIfNotExist
            fv.setdefault(nid, []).append(i_feature)
    must = scoring != 5
    if weighting in (0, 1) and bow and not must:
        nd = float(len(bow))
        for k in bow:
            bow[k] /= nd
    if must:
        keys = sorted(bow)
        norm = 0.0
        if scoring == 1:
            for k in keys:
                norm += bow[k] * bow[k]
            norm = math.sqrt(norm)
        else:
            for k in keys:
                norm += abs(bow[k])
        if norm > 0.0:
            for k in keys:
                bow[k] /= norm
    return bow, fv
