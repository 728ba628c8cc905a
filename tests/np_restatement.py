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
