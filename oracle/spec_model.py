"""spec_model.py — a second, independently structured restatement of the path, for checking the oracle.

TEST INFRASTRUCTURE ONLY (tests/test_spec_model.py). oracle/dryv_oracle.c follows the reference's Rust sources
line by line (column-major planes, per-sample neighbour derivation, its own control flow). This file is written
the other way round: from the clauses of ITU-T H.264 (8.3.1-8.3.4, 8.5.6-8.5.13) as plain Python on row-major
numpy planes, macroblock by macroblock, and then patched with the reference's five departures from the standard
(SURVEY.md 8a': Q1-Q5), each behind one `if QUIRKS:` so the patches are visible. Agreement of the two on random
inputs says the oracle transcribes tables, scans, butterflies and the 22 prediction modes consistently; it does not
pin either of them to the reference (PARITY UNPINNED, see the oracle's header).

Slow by design (pure Python per pixel): use on a few hundred macroblocks.
"""
import numpy as np

QUIRKS = True

ZZ4 = [(0, 0), (0, 1), (1, 0), (2, 0), (1, 1), (0, 2), (0, 3), (1, 2), (2, 1), (3, 0), (3, 1), (2, 2), (1, 3), (2, 3),
       (3, 2), (3, 3)]  # (row, col) of 4x4 frame-scan position k (table 8-13 / figure 8-8)


def _zz8():
    # 8x8 frame zig-zag (figure 8-9): walk the anti-diagonals, alternating direction
    out = []
    for s in range(15):
        cells = [(i, s - i) for i in range(8) if 0 <= s - i < 8]  # (row, col), row ascending
        out += cells if s % 2 else cells[::-1]
    return out


ZZ8 = _zz8()
V4 = [(10, 16, 13), (11, 18, 14), (13, 20, 16), (14, 23, 18), (16, 25, 20), (18, 29, 23)]          # table 8-? normAdjust4x4
V8 = [(20, 18, 32, 19, 25, 24), (22, 19, 35, 21, 28, 26), (26, 23, 42, 24, 33, 31), (28, 25, 45, 26, 35, 33),
      (32, 28, 51, 30, 40, 38), (36, 32, 58, 34, 46, 43)]                                          # normAdjust8x8
QPC = [29, 30, 31, 32, 32, 33, 34, 34, 35, 35, 36, 36, 37, 37, 37, 38, 38, 38, 39, 39, 39, 39]       # table 8-15, qPI 30..51


def level_scale4(list16):
    w = [[0] * 4 for _ in range(4)]
    for k, (i, j) in enumerate(ZZ4):
        w[i][j] = int(list16[k])
    ls = np.zeros((6, 4, 4), dtype=np.int64)
    for m in range(6):
        for i in range(4):
            for j in range(4):
                v = V4[m][0] if (i % 2 == 0 and j % 2 == 0) else V4[m][1] if (i % 2 == 1 and j % 2 == 1) else V4[m][2]
                ls[m, i, j] = w[i][j] * v
    return ls


def level_scale8(list64):
    w = [[0] * 8 for _ in range(8)]
    for k, (i, j) in enumerate(ZZ8):
        w[i][j] = int(list64[k])
    ls = np.zeros((6, 8, 8), dtype=np.int64)
    for m in range(6):
        for i in range(8):
            for j in range(8):
                if i % 4 == 0 and j % 4 == 0:
                    c = 0
                elif i % 2 == 1 and j % 2 == 1:
                    c = 1
                elif i % 4 == 2 and j % 4 == 2:
                    c = 2
                elif (i % 4 == 0 and j % 2 == 1) or (i % 2 == 1 and j % 4 == 0):
                    c = 3
                elif (i % 4 == 0 and j % 4 == 2) or (i % 4 == 2 and j % 4 == 0):
                    c = 4
                else:
                    c = 5
                ls[m, i, j] = w[i][j] * V8[m][c]
    return ls


def qp_chroma(qpy, offset):
    qpi = min(max(qpy + offset, 0), 51)
    return qpi if qpi < 30 else QPC[qpi - 30]


def _bf4(d):
    e0, e1, e2, e3 = d[0] + d[2], d[0] - d[2], (d[1] >> 1) - d[3], d[1] + (d[3] >> 1)
    return [e0 + e3, e1 + e2, e1 - e2, e0 - e3]


def residual4x4(c, ls, qp, dc_given):
    """8.5.12: c[i][j] -> r[i][j] (i = row). dc_given: c[0][0] is an already scaled DC."""
    d = [[0] * 4 for _ in range(4)]
    for i in range(4):
        for j in range(4):
            if dc_given and i == 0 and j == 0:
                d[i][j] = c[0][0]
            elif qp >= 24:
                d[i][j] = (c[i][j] * int(ls[qp % 6, i, j])) * (1 << (qp // 6 - 4))
            else:
                d[i][j] = (c[i][j] * int(ls[qp % 6, i, j]) + (1 << (3 - qp // 6))) >> (4 - qp // 6)
    f = [_bf4(row) for row in d]
    r = [[0] * 4 for _ in range(4)]
    for j in range(4):
        col = _bf4([f[i][j] for i in range(4)])
        for i in range(4):
            r[i][j] = (col[i] + 32) >> 6
    return r


def _bf8(d):
    e = [d[0] + d[4], -d[3] + d[5] - d[7] - (d[7] >> 1), d[0] - d[4], d[1] + d[7] - d[3] - (d[3] >> 1),
         (d[2] >> 1) - d[6], -d[1] + d[7] + d[5] + (d[5] >> 1), d[2] + (d[6] >> 1), d[3] + d[5] + d[1] + (d[1] >> 1)]
    f = [e[0] + e[6], e[1] + (e[7] >> 2), e[2] + e[4], e[3] + (e[5] >> 2), e[2] - e[4], (e[3] >> 2) - e[5], e[0] - e[6],
         e[7] - (e[1] >> 2)]
    return [f[0] + f[7], f[2] + f[5], f[4] + f[3], f[6] + f[1], f[6] - f[1], f[4] - f[3], f[2] - f[5], f[0] - f[7]]


def residual8x8(list64, ls8, qp):
    c = [[0] * 8 for _ in range(8)]
    for k, (i, j) in enumerate(ZZ8):
        c[i][j] = int(list64[k])
    d = [[0] * 8 for _ in range(8)]
    for i in range(8):
        for j in range(8):
            p = c[i][j] * int(ls8[qp % 6, i, j])
            d[i][j] = p * (1 << (qp // 6 - 6)) if qp >= 36 else (p + (1 << (5 - qp // 6))) >> (6 - qp // 6)
    g = [_bf8(row) for row in d]
    r = [[0] * 8 for _ in range(8)]
    for j in range(8):
        col = _bf8([g[i][j] for i in range(8)])
        for i in range(8):
            r[i][j] = (col[i] + 32) >> 6
    return r


def _clip(v):
    return 0 if v < 0 else 255 if v > 255 else v


BLK_XY = [((b & 1) * 4 + ((b >> 2) & 1) * 8, ((b >> 1) & 1) * 4 + ((b >> 3) & 1) * 8) for b in range(16)]  # fig. 6-10


def blk_of(bx, by):
    return (by >> 1) * 8 + (bx >> 1) * 4 + (by & 1) * 2 + (bx & 1)


# reference samples a mode needs: (top, left, corner)
NEEDS = {0: (1, 0, 0), 1: (0, 1, 0), 2: (0, 0, 0), 3: (1, 0, 0), 4: (1, 1, 1), 5: (1, 1, 1), 6: (1, 1, 1), 7: (1, 0, 0),
         8: (0, 1, 0)}


def pred_nxn(mode, n, p, top_av, left_av, corner_av):
    """8.3.1.2.x (n = 4) / 8.3.2.2.x (n = 8, p already filtered). p: dict (x, y) -> sample. Returns pred[y][x]."""
    out = [[0] * n for _ in range(n)]
    need = NEEDS[mode]
    if (need[0] and not top_av) or (need[1] and not left_av) or (need[2] and not corner_av):
        assert QUIRKS  # a conforming stream never does this; the reference leaves its zero-initialised block (Q4)
        return out
    T = lambda x: p[(x, -1)]
    L = lambda y: p[(-1, y)]
    last = 2 * n - 1
    for y in range(n):
        for x in range(n):
            if mode == 0:
                v = T(x)
            elif mode == 1:
                v = L(y)
            elif mode == 2:
                sh = 2 if n == 4 else 3
                st = sum(T(i) for i in range(n)) if top_av else 0
                sl = sum(L(i) for i in range(n)) if left_av else 0
                if top_av and left_av:
                    v = (st + sl + n) >> (sh + 1)
                elif left_av:
                    v = (sl + n // 2) >> sh
                elif top_av:
                    v = (st + n // 2) >> sh
                else:
                    v = 128
            elif mode == 3:
                if x == n - 1 and y == n - 1:
                    v = (T(last - 1) + 3 * T(last) + 2) >> 2
                else:
                    v = (T(x + y) + 2 * T(x + y + 1) + T(x + y + 2) + 2) >> 2
            elif mode == 4:
                if x > y:
                    v = (T(x - y - 2) + 2 * T(x - y - 1) + T(x - y) + 2) >> 2 if x - y - 2 >= 0 else \
                        (p[(-1, -1)] + 2 * T(0) + T(1) + 2) >> 2
                elif x < y:
                    v = (L(y - x - 2) + 2 * L(y - x - 1) + L(y - x) + 2) >> 2 if y - x - 2 >= 0 else \
                        (p[(-1, -1)] + 2 * L(0) + L(1) + 2) >> 2
                else:
                    v = (T(0) + 2 * p[(-1, -1)] + L(0) + 2) >> 2
            elif mode == 5:
                z, k = 2 * x - y, x - (y >> 1)
                P = lambda i: p[(i, -1)]  # i = -1 is the corner
                if z >= 0 and z % 2 == 0:
                    v = (P(k - 1) + P(k) + 1) >> 1
                elif z >= 0:
                    v = (P(k - 2) + 2 * P(k - 1) + P(k) + 2) >> 2
                elif z == -1:
                    v = (L(0) + 2 * p[(-1, -1)] + T(0) + 2) >> 2
                else:
                    Q = lambda i: p[(-1, i)]  # i = -1 is the corner
                    v = (Q(y - 2 * x - 1) + 2 * Q(y - 2 * x - 2) + Q(y - 2 * x - 3) + 2) >> 2
            elif mode == 6:
                z, k = 2 * y - x, y - (x >> 1)
                Q = lambda i: p[(-1, i)]
                if z >= 0 and z % 2 == 0:
                    v = (Q(k - 1) + Q(k) + 1) >> 1
                elif z >= 0:
                    v = (Q(k - 2) + 2 * Q(k - 1) + Q(k) + 2) >> 2
                elif z == -1:
                    v = (L(0) + 2 * p[(-1, -1)] + T(0) + 2) >> 2
                else:
                    P = lambda i: p[(i, -1)]
                    v = (P(x - 2 * y - 1) + 2 * P(x - 2 * y - 2) + P(x - 2 * y - 3) + 2) >> 2
            elif mode == 7:
                k = x + (y >> 1)
                v = (T(k) + T(k + 1) + 1) >> 1 if y % 2 == 0 else (T(k) + 2 * T(k + 1) + T(k + 2) + 2) >> 2
            else:
                z, k = x + 2 * y, y + (x >> 1)
                zm = 2 * n - 3
                if z < zm and z % 2 == 0:
                    v = (L(k) + L(k + 1) + 1) >> 1
                elif z < zm:
                    v = (L(k) + 2 * L(k + 1) + L(k + 2) + 2) >> 2
                elif z == zm:
                    v = (L(n - 2) + 3 * L(n - 1) + 2) >> 2
                else:
                    v = L(n - 1)
            out[y][x] = v
    return out


def filter8x8(p, top_av, left_av, corner_av, tr_av):
    """8.3.2.2.1 reference sample filtering. p holds the raw samples that exist; returns the filtered dict."""
    q = {}
    if top_av:
        if not tr_av:
            for x in range(8, 16):
                p[(x, -1)] = p[(7, -1)]
        q[(0, -1)] = (p[(-1, -1)] + 2 * p[(0, -1)] + p[(1, -1)] + 2) >> 2 if corner_av else \
            (3 * p[(0, -1)] + p[(1, -1)] + 2) >> 2
        if QUIRKS and not corner_av:
            # Q1: the reference's 3-tap loop starts at x = 0 and uses its "unavailable" marker -1 as p[-1,-1]
            q[(0, -1)] = (-1 + 2 * p[(0, -1)] + p[(1, -1)] + 2) >> 2
        for x in range(1, 15):
            q[(x, -1)] = (p[(x - 1, -1)] + 2 * p[(x, -1)] + p[(x + 1, -1)] + 2) >> 2
        q[(15, -1)] = (p[(14, -1)] + 3 * p[(15, -1)] + 2) >> 2
    if corner_av:
        if top_av and left_av:
            q[(-1, -1)] = (p[(0, -1)] + 2 * p[(-1, -1)] + p[(-1, 0)] + 2) >> 2
        elif top_av:
            q[(-1, -1)] = (3 * p[(-1, -1)] + p[(0, -1)] + 2) >> 2
        elif left_av:
            q[(-1, -1)] = (3 * p[(-1, -1)] + p[(-1, 0)] + 2) >> 2
        else:
            q[(-1, -1)] = p[(-1, -1)]
    if left_av:
        q[(-1, 0)] = (p[(-1, -1)] + 2 * p[(-1, 0)] + p[(-1, 1)] + 2) >> 2 if corner_av else \
            (3 * p[(-1, 0)] + p[(-1, 1)] + 2) >> 2
        for y in range(1, 7):
            q[(-1, y)] = (p[(-1, y - 1)] + 2 * p[(-1, y)] + p[(-1, y + 1)] + 2) >> 2
        q[(-1, 7)] = (p[(-1, 6)] + 3 * p[(-1, 7)] + 2) >> 2
    return q


class Picture:
    def __init__(self, fp):
        self.W, self.H = fp.pic_width_in_mbs, fp.pic_height_in_mbs
        self.Y = np.zeros((16 * self.H, 16 * self.W), dtype=np.int64)
        self.C = [np.zeros((8 * self.H, 8 * self.W), dtype=np.int64) for _ in range(2)]
        self.kind = {}    # (mbx, mby) -> 0/1/2 (3 = unsupported record)
        self.modes = {}   # (mbx, mby) -> 16 modes by luma4x4BlkIdx (I8x8: the block's mode on its four positions)
        sl4 = np.array(fp.scaling_list4x4, dtype=np.int64).reshape(6, 16)
        sl8 = np.array(fp.scaling_list8x8, dtype=np.int64).reshape(6, 64)
        self.ls4 = level_scale4(sl4[0])   # Q3: list 0 (Intra Y) also serves Cb and Cr in the reference
        self.ls8 = level_scale8(sl8[0])
        self.cqo = (fp.chroma_qp_index_offset, fp.second_chroma_qp_index_offset)

    def has(self, mbx, mby):
        return 0 <= mbx < self.W and 0 <= mby < self.H and (mbx, mby) in self.kind

    def neighbour_mode(self, mbx, mby, bx, by):
        """Intra4x4PredMode / Intra8x8PredMode of the block covering 4x4 position (bx, by) of macroblock (mbx, mby), as
        8.3.1.1 / 8.3.2.1 see it: 2 for anything that is not Intra4x4 / Intra8x8."""
        if self.kind[(mbx, mby)] in (0, 1):
            return self.modes[(mbx, mby)][blk_of(bx, by)]
        return 2


def decode_mb(pic, mbx, mby, desc, co):
    kind, qp = int(desc["mb_kind"]), int(desc["qp"])
    if kind > 2 or qp > 51 or int(desc["i16_pred_mode"]) > 3 or int(desc["intra_chroma_pred_mode"]) > 3:
        pic.kind[(mbx, mby)] = 3
        return 1
    pic.kind[(mbx, mby)] = kind
    A, B = pic.has(mbx - 1, mby), pic.has(mbx, mby - 1)
    Cc, D = pic.has(mbx + 1, mby - 1), pic.has(mbx - 1, mby - 1)
    X0, Y0 = 16 * mbx, 16 * mby
    co = [int(v) for v in co]
    prev = int(desc["prev_flags"])
    rem = [(int(desc["rem_modes"][i >> 1]) >> (4 * (i & 1))) & 7 for i in range(16)]
    Yp = pic.Y

    def sample(x, y):  # luma sample at macroblock-relative (x, y); None when outside the decoded area
        gx, gy = X0 + x, Y0 + y
        mbx2, mby2 = gx // 16, gy // 16
        if gx < 0 or gy < 0 or not pic.has(mbx2, mby2):
            return None
        return int(Yp[gy, gx])

    modes = [2] * 16
    if kind == 0:
        for b in range(16):
            ox, oy = BLK_XY[b]
            bx, by = ox // 4, oy // 4
            left_av, top_av = bx > 0 or A, by > 0 or B
            # 8.3.1.1
            if not left_av or not top_av:
                pm = 2
            else:
                ma = modes[blk_of(bx - 1, by)] if bx > 0 else pic.neighbour_mode(mbx - 1, mby, 3, by)
                mb_ = modes[blk_of(bx, by - 1)] if by > 0 else pic.neighbour_mode(mbx, mby - 1, bx, 3)
                pm = min(ma, mb_)
            mode = pm if (prev >> b) & 1 else (rem[b] if rem[b] < pm else rem[b] + 1)
            modes[b] = mode
            pic.modes[(mbx, mby)] = modes
            corner_av = sample(ox - 1, oy - 1) is not None  # inside the macroblock it was decoded earlier
            # top-right: inside the macroblock only blocks decoded earlier count
            if by == 0:
                tr_av = B if bx < 3 else Cc
            else:
                tr_av = bx < 3 and blk_of(bx + 1, by - 1) < b
            p = {}
            for x in range(-1, 8):  # 8.3.1.2: without a top-right block p[4..7,-1] := p[3,-1]
                p[(x, -1)] = sample(ox + 3, oy - 1) if (x >= 4 and not tr_av) else sample(ox + x, oy - 1)
            for y in range(4):
                p[(-1, y)] = sample(ox - 1, oy + y)
            pr = pred_nxn(mode, 4, p, top_av, left_av, corner_av)
            lst = co[16 * b:16 * b + 16]
            c = [[0] * 4 for _ in range(4)]
            for k, (i, j) in enumerate(ZZ4):
                c[i][j] = lst[k]
            r = residual4x4(c, pic.ls4, qp, False)
            for y in range(4):
                for x in range(4):
                    Yp[Y0 + oy + y, X0 + ox + x] = _clip(pr[y][x] + r[y][x])
    elif kind == 1:
        for b8 in range(4):
            ox, oy = 8 * (b8 & 1), 8 * (b8 >> 1)
            bx, by = b8 & 1, b8 >> 1
            left_av, top_av = bx > 0 or A, by > 0 or B
            if not left_av or not top_av:
                pm = 2
            else:
                ma = modes[blk_of(2 * bx - 1, 2 * by)] if bx > 0 else pic.neighbour_mode(mbx - 1, mby, 3, 2 * by)
                mb_ = modes[blk_of(2 * bx, 2 * by - 1)] if by > 0 else pic.neighbour_mode(mbx, mby - 1, 2 * bx, 3)
                pm = min(ma, mb_)
            r8 = rem[b8]
            mode = pm if (prev >> b8) & 1 else (r8 if r8 < pm else r8 + 1)
            for k in range(4):
                modes[4 * b8 + k] = mode
            pic.modes[(mbx, mby)] = modes
            corner_av = [D, B, A, True][b8]
            tr_av = [B, Cc, True, False][b8]
            p = {}
            for x in range(-1, 16):
                v = sample(ox + x, oy - 1)
                if v is not None:
                    p[(x, -1)] = v
            for y in range(8):
                v = sample(ox - 1, oy + y)
                if v is not None:
                    p[(-1, y)] = v
            if not tr_av:  # samples of not-yet-decoded or absent blocks do not count
                for x in range(8, 16):
                    p.pop((x, -1), None)
            q = filter8x8(p, top_av, left_av, corner_av, tr_av)
            pr = pred_nxn(mode, 8, q, top_av, left_av, corner_av)
            r = residual8x8(co[64 * b8:64 * b8 + 64], pic.ls8, qp)
            for y in range(8):
                for x in range(8):
                    Yp[Y0 + oy + y, X0 + ox + x] = _clip(pr[y][x] + r[y][x])
    else:
        pic.modes[(mbx, mby)] = modes
        m16 = int(desc["i16_pred_mode"])
        pr = [[0] * 16 for _ in range(16)]
        T = [sample(x, -1) for x in range(16)] if B else None
        L = [sample(-1, y) for y in range(16)] if A else None
        if m16 == 0:
            if B:
                pr = [[T[x] for x in range(16)] for _ in range(16)]
        elif m16 == 1:
            if A:
                pr = [[L[y]] * 16 for y in range(16)]
        elif m16 == 2:
            if A and B:
                v = (sum(T) + sum(L) + 16) >> 5
            elif A:
                v = (sum(L) + 8) >> 4
            elif B:
                v = (sum(T) + 8) >> 4
            else:
                v = 128
            pr = [[v] * 16 for _ in range(16)]
        elif A and B:  # plane (Q5: the corner is read unchecked; with A and B present D always is)
            corner = sample(-1, -1)
            TT = lambda x: corner if x < 0 else T[x]
            LL = lambda y: corner if y < 0 else L[y]
            Hh = sum((k + 1) * (TT(8 + k) - TT(6 - k)) for k in range(8))
            Vv = sum((k + 1) * (LL(8 + k) - LL(6 - k)) for k in range(8))
            a, b, c = 16 * (L[15] + T[15]), (5 * Hh + 32) >> 6, (5 * Vv + 32) >> 6
            pr = [[_clip((a + b * (x - 7) + c * (y - 7) + 16) >> 5) for x in range(16)] for y in range(16)]
        # 8.5.10: luma DC
        c = [[0] * 4 for _ in range(4)]
        for k, (i, j) in enumerate(ZZ4):
            c[i][j] = co[k]
        Am = [[1, 1, 1, 1], [1, 1, -1, -1], [1, -1, -1, 1], [1, -1, 1, -1]]
        t = [[sum(Am[i][k] * c[k][j] for k in range(4)) for j in range(4)] for i in range(4)]
        f = [[sum(t[i][k] * Am[k][j] for k in range(4)) for j in range(4)] for i in range(4)]
        ls00 = int(pic.ls4[qp % 6, 0, 0])
        dc = [[(f[i][j] * ls00) * (1 << (qp // 6 - 6)) if qp >= 36 else
               (f[i][j] * ls00 + (1 << (5 - qp // 6))) >> (6 - qp // 6) for j in range(4)] for i in range(4)]
        for b in range(16):
            ox, oy = BLK_XY[b]
            lst = [dc[oy // 4][ox // 4]] + co[16 + 15 * b:16 + 15 * b + 15]
            cc = [[0] * 4 for _ in range(4)]
            for k, (i, j) in enumerate(ZZ4):
                cc[i][j] = lst[k]
            r = residual4x4(cc, pic.ls4, qp, True)
            for y in range(4):
                for x in range(4):
                    Yp[Y0 + oy + y, X0 + ox + x] = _clip(pr[oy + y][ox + x] + r[y][x])

    # ---- chroma (8.3.4, 8.5.11) -------------------------------------------------------------
    cm = int(desc["intra_chroma_pred_mode"])
    for pl in range(2):
        P = pic.C[pl]
        cx0, cy0 = 8 * mbx, 8 * mby
        T = [int(P[cy0 - 1, cx0 + x]) for x in range(8)] if B else None
        L = [int(P[cy0 + y, cx0 - 1]) for y in range(8)] if A else None
        pr = [[0] * 8 for _ in range(8)]
        if cm == 0:
            for blk in range(4):
                ox, oy = 4 * (blk & 1), 4 * (blk >> 1)
                t = T[ox:ox + 4] if B else [-1] * 4
                l = L[oy:oy + 4] if A else [-1] * 4
                st, sl = sum(t), sum(l)
                if QUIRKS:
                    # Q2: the reference tests "> 0" on samples where the standard asks whether they are available
                    if (ox == 0 and oy == 0) or (ox > 0 and oy > 0):
                        if A and B:
                            v = (st + sl + 4) >> 3
                        elif A:
                            v = (sl + 2) >> 2
                        elif B and all(s > 0 for s in t):
                            v = (st + 2) >> 2
                        else:
                            v = 128
                    elif ox > 0:
                        v = (st + 2) >> 2 if B else (sl + 2) >> 2 if (A and l[3] > 0) else 128
                    else:
                        v = (sl + 2) >> 2 if (A and l[3] > 0) else (st + 2) >> 2 if (B and t[3] > 0) else 128
                else:
                    if (ox == 0 and oy == 0) or (ox > 0 and oy > 0):
                        v = (st + sl + 4) >> 3 if (A and B) else (sl + 2) >> 2 if A else (st + 2) >> 2 if B else 128
                    elif ox > 0:
                        v = (st + 2) >> 2 if B else (sl + 2) >> 2 if A else 128
                    else:
                        v = (sl + 2) >> 2 if A else (st + 2) >> 2 if B else 128
                for y in range(4):
                    for x in range(4):
                        pr[oy + y][ox + x] = v
        elif cm == 1:
            if A:
                pr = [[L[y]] * 8 for y in range(8)]
        elif cm == 2:
            if B:
                pr = [list(T) for _ in range(8)]
        elif A and B:
            corner = int(P[cy0 - 1, cx0 - 1])
            TT = lambda x: corner if x < 0 else T[x]
            LL = lambda y: corner if y < 0 else L[y]
            Hh = sum((k + 1) * (TT(4 + k) - TT(2 - k)) for k in range(4))
            Vv = sum((k + 1) * (LL(4 + k) - LL(2 - k)) for k in range(4))
            a, b, c = 16 * (L[7] + T[7]), (34 * Hh + 32) >> 6, (34 * Vv + 32) >> 6
            pr = [[_clip((a + b * (x - 3) + c * (y - 3) + 16) >> 5) for x in range(8)] for y in range(8)]
        qpc = qp_chroma(qp, pic.cqo[pl])
        base = 256 + 64 * pl
        cdc = co[base:base + 4]
        f = [[cdc[0] + cdc[1] + cdc[2] + cdc[3], cdc[0] - cdc[1] + cdc[2] - cdc[3]],
             [cdc[0] + cdc[1] - cdc[2] - cdc[3], cdc[0] - cdc[1] - cdc[2] + cdc[3]]]
        ls00 = int(pic.ls4[qpc % 6, 0, 0])
        for blk in range(4):
            ox, oy = 4 * (blk & 1), 4 * (blk >> 1)
            dcv = ((f[blk >> 1][blk & 1] * ls00) * (1 << (qpc // 6))) >> 5
            lst = [dcv] + co[base + 4 + 15 * blk:base + 4 + 15 * blk + 15]
            cc = [[0] * 4 for _ in range(4)]
            for k, (i, j) in enumerate(ZZ4):
                cc[i][j] = lst[k]
            r = residual4x4(cc, pic.ls4, qpc, True)
            for y in range(4):
                for x in range(4):
                    P[cy0 + oy + y, cx0 + ox + x] = _clip(pr[oy + y][ox + x] + r[y][x])
    return 0


def reconstruct(fp, n_frames, mbs, coeffs):
    """Same contract as oracle.reconstruct: (status, uint8 planes Y|Cb|Cr per frame, row-major)."""
    W, H = fp.pic_width_in_mbs, fp.pic_height_in_mbs
    out = []
    status = 0
    coeffs = np.asarray(coeffs).reshape(-1, 384)
    for f in range(n_frames):
        pic = Picture(fp)
        for mby in range(H):
            for mbx in range(W):
                a = (f * H + mby) * W + mbx
                status |= decode_mb(pic, mbx, mby, mbs[a], coeffs[a])
        out += [pic.Y.astype(np.uint8).reshape(-1), pic.C[0].astype(np.uint8).reshape(-1),
                pic.C[1].astype(np.uint8).reshape(-1)]
    return (-2 if status else 0), np.concatenate(out)
