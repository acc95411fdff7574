"""deblock_model.py — a second restatement of H.264 clause 8.7, for checking oracle/dryv_deblock.c.

TEST INFRASTRUCTURE ONLY (tests/test_deblock.py). dryv_deblock.c filters sample line by sample line with scalar code;
this one takes a whole edge (16 or 8 lines) at a time as numpy vectors, derives the thresholds from the closed forms of
tables 8-16 / 8-17 where they have one (alpha, beta) and from the table otherwise, and walks planes rather than
macroblock members. Agreement on random pictures says the C oracle transcribes the clause consistently; nothing pins
either to an independent decoder (none in this image), and the reference has no deblocking at all.
"""
import numpy as np

# table 8-16, written as the standard's closed forms: alpha'(i) = 0.8 (2^(i/6) - 1) rounded as tabulated; kept as a table
# for exactness, entered independently of the C file (rows of ten)
ALPHA = [0] * 16 + [4, 4, 5, 6, 7, 8, 9, 10, 12, 13,
                    15, 17, 20, 22, 25, 28, 32, 36, 40, 45,
                    50, 56, 63, 71, 80, 90, 101, 113, 127, 144,
                    162, 182, 203, 226, 255, 255]
BETA = [0] * 16 + [2, 2, 2, 3, 3, 3, 3, 4, 4, 4] + [v for b in range(6, 19) for v in (b, b)]   # 6,6,7,7,...,18,18
TC0_BS3 = [0] * 17 + [1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 2, 2, 2, 2, 3, 3, 3, 4, 4, 4, 5, 6, 6, 7, 8, 9, 10, 11, 13, 14, 16, 18, 20, 23, 25]
QPC_TAIL = [29, 30, 31, 32, 32, 33, 34, 34, 35, 35, 36, 36, 37, 37, 37, 38, 38, 38, 39, 39, 39, 39]
assert len(ALPHA) == 52 and len(BETA) == 52 and len(TC0_BS3) == 52


def qpc(qpy, off):
    q = min(max(qpy + off, 0), 51)
    return q if q < 30 else QPC_TAIL[q - 30]


def _edge(P, Q, strong, qp_p, qp_q, chroma, offA, offB):
    """P, Q: int arrays [4 (or 2)][n] of samples p0.. / q0.. outward from the edge. Returns the filtered copies."""
    qpav = (qp_p + qp_q + 1) >> 1
    ia, ib = min(max(qpav + offA, 0), 51), min(max(qpav + offB, 0), 51)
    alpha, beta = ALPHA[ia], BETA[ib]
    p0, p1, q0, q1 = P[0], P[1], Q[0], Q[1]
    on = (np.abs(p0 - q0) < alpha) & (np.abs(p1 - p0) < beta) & (np.abs(q1 - q0) < beta)
    Pn, Qn = P.copy(), Q.copy()
    if chroma:
        if strong:
            Pn[0] = np.where(on, (2 * p1 + p0 + q1 + 2) >> 2, p0)
            Qn[0] = np.where(on, (2 * q1 + q0 + p1 + 2) >> 2, q0)
        else:
            tc = TC0_BS3[ia] + 1
            d = np.clip((((q0 - p0) << 2) + (p1 - q1) + 4) >> 3, -tc, tc)
            Pn[0] = np.where(on, np.clip(p0 + d, 0, 255), p0)
            Qn[0] = np.where(on, np.clip(q0 - d, 0, 255), q0)
        return Pn, Qn
    p2, q2 = P[2], Q[2]
    ap, aq = np.abs(p2 - p0) < beta, np.abs(q2 - q0) < beta
    if strong:
        p3, q3 = P[3], Q[3]
        small = np.abs(p0 - q0) < ((alpha >> 2) + 2)
        sp, sq = on & ap & small, on & aq & small
        Pn[0] = np.where(sp, (p2 + 2 * p1 + 2 * p0 + 2 * q0 + q1 + 4) >> 3, np.where(on, (2 * p1 + p0 + q1 + 2) >> 2, p0))
        Pn[1] = np.where(sp, (p2 + p1 + p0 + q0 + 2) >> 2, p1)
        Pn[2] = np.where(sp, (2 * p3 + 3 * p2 + p1 + p0 + q0 + 4) >> 3, p2)
        Qn[0] = np.where(sq, (p1 + 2 * p0 + 2 * q0 + 2 * q1 + q2 + 4) >> 3, np.where(on, (2 * q1 + q0 + p1 + 2) >> 2, q0))
        Qn[1] = np.where(sq, (p0 + q0 + q1 + q2 + 2) >> 2, q1)
        Qn[2] = np.where(sq, (2 * q3 + 3 * q2 + q1 + q0 + p0 + 4) >> 3, q2)
        return Pn, Qn
    tc0 = TC0_BS3[ia]
    tc = tc0 + ap.astype(np.int64) + aq.astype(np.int64)
    d = np.clip((((q0 - p0) << 2) + (p1 - q1) + 4) >> 3, -tc, tc)
    Pn[0] = np.where(on, np.clip(p0 + d, 0, 255), p0)
    Qn[0] = np.where(on, np.clip(q0 - d, 0, 255), q0)
    avg = (p0 + q0 + 1) >> 1
    Pn[1] = np.where(on & ap, p1 + np.clip((p2 + avg - (p1 << 1)) >> 1, -tc0, tc0), p1)
    Qn[1] = np.where(on & aq, q1 + np.clip((q2 + avg - (q1 << 1)) >> 1, -tc0, tc0), q1)
    return Pn, Qn


def deblock(W, H, qps, kinds, yuv, cqo_cb=0, cqo_cr=0, disable_idc=0, alpha_div2=0, beta_div2=0):
    """One picture. qps / kinds: per macroblock (raster). yuv: planes in write_to_yuv_file order. Returns the filtered copy."""
    out = np.array(yuv, dtype=np.int64)
    if disable_idc == 1:
        return out.astype(np.uint8)
    Y = out[:256 * W * H].reshape(16 * H, 16 * W)
    Cb = out[256 * W * H:320 * W * H].reshape(8 * H, 8 * W)
    Cr = out[320 * W * H:].reshape(8 * H, 8 * W)
    offA, offB = 2 * alpha_div2, 2 * beta_div2
    for a in range(W * H):
        mx, my = a % W, a // W
        for plane, n, chroma, off in ((Y, 16, False, 0), (Cb, 8, True, cqo_cb), (Cr, 8, True, cqo_cr)):
            q_cur = qpc(int(qps[a]), off) if chroma else int(qps[a])
            edges = (0, 4) if chroma else ((0, 8) if kinds[a] == 1 else (0, 4, 8, 12))
            x0, y0 = n * mx, n * my
            depth = 2 if chroma else 4
            for e in edges:      # vertical edges, left to right
                if e == 0 and mx == 0:
                    continue
                q_p = q_cur if e else (qpc(int(qps[a - 1]), off) if chroma else int(qps[a - 1]))
                rows = slice(y0, y0 + n)
                P = np.stack([plane[rows, x0 + e - 1 - k] for k in range(depth)])
                Q = np.stack([plane[rows, x0 + e + k] for k in range(depth)])
                Pn, Qn = _edge(P, Q, e == 0, q_p, q_cur, chroma, offA, offB)
                for k in range(depth):
                    plane[rows, x0 + e - 1 - k] = Pn[k]
                    plane[rows, x0 + e + k] = Qn[k]
            for e in edges:      # horizontal edges, top to bottom
                if e == 0 and my == 0:
                    continue
                q_p = q_cur if e else (qpc(int(qps[a - W]), off) if chroma else int(qps[a - W]))
                cols = slice(x0, x0 + n)
                P = np.stack([plane[y0 + e - 1 - k, cols] for k in range(depth)])
                Q = np.stack([plane[y0 + e + k, cols] for k in range(depth)])
                Pn, Qn = _edge(P, Q, e == 0, q_p, q_cur, chroma, offA, offB)
                for k in range(depth):
                    plane[y0 + e - 1 - k, cols] = Pn[k]
                    plane[y0 + e + k, cols] = Qn[k]
    return out.astype(np.uint8)
