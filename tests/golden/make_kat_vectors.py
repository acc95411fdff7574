#!/usr/bin/env python3
"""Writes tests/golden/kat_vectors.json: hand-derived known-answer vectors for dryv's
reconstruction path.

The reference ships no tests or fixtures and cannot be built in this environment, so these
vectors are NOT outputs of the reference: each expected value below is worked out with plain
arithmetic from the reference lines cited next to it (paths under /root/reference/src/video/frame).
Neither the oracle nor the GPU code is imported here.
"""
import json
import os

FLAT_LS = 16  # flat scaling list entry (slice/header.rs:330)
V4_00 = [10, 11, 13, 14, 16, 18]   # normAdjust4x4(m, 0, 0), transform.rs:24-31 column 0
V8_00 = [20, 22, 26, 28, 32, 36]   # normAdjust8x8(m, 0, 0), transform.rs:50-57 column 0

vectors = []


def r_from_dc_only(d00):
    # A lone d[0][0] passes both butterfly stages unchanged to every h[i][j]
    # (transform.rs:159-181 / pred8x8.rs:85-141), then r = (h + 32) >> 6 (transform.rs:183-187).
    return (d00 + 32) >> 6


# K1: all-zero coefficients, Intra16x16 DC + chroma DC, no neighbours -> 128 everywhere
# (pred16x16.rs:358 `1 << (bit_depth_y - 1)`, trans_chroma.rs:225; zero coeffs give r = 0).
vectors.append(dict(name="K1_zero_i16_dc", W=1, H=1,
                    mbs=[dict(kind=2, qp=26, i16=2, chroma=0, prev=0, rem=[0] * 16, coeffs={})],
                    expect=dict(Y=128, Cb=128, Cr=128)))

# K2: Intra4x4, blk0 list [1,0,...], qp 28: LevelScale = 16*16 = 256 (m = 28 % 6 = 4, V4[4][0] = 16),
# d = 256 << (28/6 - 4) = 256 (transform.rs:147-148), r = (256+32)>>6 = 4; DC prediction without
# neighbours = 128 (pred4x4.rs:160) -> blk0 = 132. Every later block has zero residual and a DC
# prediction over available neighbours that are all 132 -> whole macroblock 132
# (pred4x4.rs:116-161: (4*132+2)>>2 = 132, (8*132+4)>>3 = 132). prev_flag = 1 everywhere ->
# predicted mode = DC because a neighbour is missing or is DC (pred4x4.rs:386-417).
d = (1 * FLAT_LS * V4_00[28 % 6]) << (28 // 6 - 4)
assert r_from_dc_only(d) == 4
vectors.append(dict(name="K2_i4x4_dc1_qp28", W=1, H=1,
                    mbs=[dict(kind=0, qp=28, i16=0, chroma=0, prev=0xFFFF, rem=[0] * 16, coeffs={"0": 1})],
                    expect=dict(Y=132, Cb=128, Cr=128)))

# K3: same at qp 20: LS = 16*13 = 208 (m = 2), d = (208 + (1 << (3 - 3))) >> (4 - 3) = 104
# (transform.rs:150-152), r = (104+32)>>6 = 2 -> 130.
d = (1 * FLAT_LS * V4_00[20 % 6] + (1 << (3 - 20 // 6))) >> (4 - 20 // 6)
assert d == 104 and r_from_dc_only(d) == 2
vectors.append(dict(name="K3_i4x4_dc1_qp20", W=1, H=1,
                    mbs=[dict(kind=0, qp=20, i16=0, chroma=0, prev=0xFFFF, rem=[0] * 16, coeffs={"0": 1})],
                    expect=dict(Y=130, Cb=128, Cr=128)))

# K4: Intra16x16, DC list [1,0,...], qp 28: the Hadamard of a lone c[0][0] is f == 1 everywhere
# (pred16x16.rs:446-463); dcY = (1*256 + (1 << (5-4))) >> (6-4) = 64 (:472-478); each 4x4 block gets
# d00 = 64 unscaled (transform.rs:145-146) -> r = (64+32)>>6 = 1 -> 129.
dcy = (1 * FLAT_LS * V4_00[28 % 6] + (1 << (5 - 28 // 6))) >> (6 - 28 // 6)
assert dcy == 64 and r_from_dc_only(dcy) == 1
vectors.append(dict(name="K4_i16_dc1_qp28", W=1, H=1,
                    mbs=[dict(kind=2, qp=28, i16=2, chroma=0, prev=0, rem=[0] * 16, coeffs={"0": 1})],
                    expect=dict(Y=129, Cb=128, Cr=128)))

# K5: chroma Cb DC [1,0,0,0], QPY 28 -> QPC = 28 (transform.rs:208-209), f == 1 (trans_chroma.rs:390-409),
# dcC = ((1*256) << (28/6)) >> 5 = 128 (:413), r = (128+32)>>6 = 2 -> Cb 130, Cr 128.
dcc = ((1 * FLAT_LS * V4_00[28 % 6]) << (28 // 6)) >> 5
assert dcc == 128 and r_from_dc_only(dcc) == 2
vectors.append(dict(name="K5_chroma_dc1_qp28", W=1, H=1,
                    mbs=[dict(kind=2, qp=28, i16=2, chroma=0, prev=0, rem=[0] * 16, coeffs={"256": 1})],
                    expect=dict(Y=128, Cb=130, Cr=128)))

# K6: Intra8x8 blk0 list [1,0,...]: qp 36 -> LS8 = 16*20 = 320 (m = 0), d = 320 << 0 (pred8x8.rs:73-74),
# r = (320+32)>>6 = 5 -> blk0 = v = 133; qp 30 -> d = (320 + (1 << 0)) >> 1 = 160 (:76-77), r = 3 -> v = 131.
# The other three blocks have zero residual and prev_flag = 1 -> DC (a neighbour MB is missing):
#  blk1 (8,0): only the left column (all v) exists; filtered it stays v (pred8x8.rs:266-286) -> DC = v.
#  blk2 (0,8): only the top row (all v) exists and p[-1,-1] is unavailable, so quirk Q1 applies:
#       p'[0,-1] = (-1 + 2v + v + 2) >> 2 (:245-247), p'[1..15,-1] = v -> DC = (p'[0] + 7v + 4) >> 3 (:407-416).
#  blk3 (8,8): top row = blk1's bottom row (v, top-right substituted by v), corner = v,
#       left column = blk2's right column (all w = blk2's value):
#       p'[-1,0] = (v + 2w + w + 2) >> 2, p'[-1,1..7] = w -> DC = (8v + p'[-1,0] + 7w + 8) >> 4 (:345-362).
for qp in (36, 30):
    if qp >= 36:
        d = (1 * FLAT_LS * V8_00[qp % 6]) << (qp // 6 - 6)
    else:
        d = (1 * FLAT_LS * V8_00[qp % 6] + (1 << (5 - qp // 6))) >> (6 - qp // 6)
    v = 128 + r_from_dc_only(d)
    assert v == {36: 133, 30: 131}[qp]
    q1 = (-1 + 2 * v + v + 2) >> 2
    w = (q1 + 7 * v + 4) >> 3
    l0 = (v + 2 * w + w + 2) >> 2
    b3 = (8 * v + l0 + 7 * w + 8) >> 4
    assert (w, b3) == {36: (129, 131), 30: (127, 129)}[qp]
    vectors.append(dict(name="K6_i8x8_dc1_qp%d" % qp, W=1, H=1,
                        mbs=[dict(kind=1, qp=qp, i16=0, chroma=0, prev=0xF, rem=[0] * 16, coeffs={"0": 1})],
                        expect=dict(Yq=[v, v, w, b3], Cb=128, Cr=128)))

# K7 (quirk Q1): Intra8x8 blk0 of a column-0, row>0 macroblock whose top row is all t = 100.
# p[-1,-1] is unavailable (-1). The loop at pred8x8.rs:245-247 starts at x = 0 and recomputes
# p'[0,-1] = (p[-1,-1] + 2p[0,-1] + p[1,-1] + 2) >> 2 = (-1 + 200 + 100 + 2) >> 2 = 75
# (the spec's rule, computed just before at :242, would give (3*100+100+2)>>2 = 100).
# Vertical prediction (mode 0) copies p'[x,-1] down the block: column 0 = 75, columns 1..7 = 100
# ((100+200+100+2)>>2 = 100; x = 7 uses the substituted/real top-right sample 100).
# Mode 0 is signalled with prev_flag = 0, rem = 0: predicted mode is DC (2) because macroblock A is
# unavailable (pred8x8.rs:723-734), and rem < pred -> mode = rem (:756-759).
assert (-1 + 2 * 100 + 100 + 2) >> 2 == 75
vectors.append(dict(name="K7_q1_i8x8_filter_col0", kind="single_mb", W=1, H=2, mbaddr=1,
                    neighbours=dict(Y=100, Cb=128, Cr=128, kind=2),
                    mb=dict(kind=1, qp=26, i16=0, chroma=0, prev=0xE, rem=[0] * 16, coeffs={}),
                    expect_region=dict(plane="Y", x0=0, y0=16, w=8, h=8,
                                       rows=[[75, 100, 100, 100, 100, 100, 100, 100]] * 8)))

# K8 (quirk Q2): column-0, row>0 macroblock, chroma DC prediction (mode 0), chroma block (0,0):
# top samples [0,50,50,50], left unavailable. trans_chroma.rs:209-216 asks for all four top samples
# `> 0` -> the zero sample fails the test and the block falls through to 128 (:225)
# (the spec's availability test would give (150+2)>>2 = 38).
# Block (4,0) next to it has top samples [50,50,50,50] -> (200+2)>>2 = 50 (:228-238).
vectors.append(dict(name="K8_q2_chroma_dc_zero_sample", kind="single_mb", W=1, H=2, mbaddr=1,
                    neighbours=dict(Y=128, Cb=128, Cr=128, kind=2,
                                    Cb_bottom_row=[0, 50, 50, 50, 50, 50, 50, 50]),
                    mb=dict(kind=2, qp=26, i16=2, chroma=0, prev=0, rem=[0] * 16, coeffs={}),
                    expect_region=dict(plane="Cb", x0=0, y0=8, w=8, h=4,
                                       rows=[[128, 128, 128, 128, 50, 50, 50, 50]] * 4)))

# math.rs:109-125
vectors.append(dict(name="math", kind="math",
                    clamp=[[5, 0, 255, 5], [-3, 0, 255, 0], [300, 0, 255, 255], [-13, -12, 51, -12]],
                    # (a % (d / b)) * b  /  (a / (d / b)) * c
                    inverse_raster_scan=[[5, 16, 16, 1920, 0, 80], [125, 16, 16, 1920, 0, 80],
                                         [125, 16, 16, 1920, 1, 16], [3, 4, 4, 8, 0, 4], [3, 4, 4, 8, 1, 4],
                                         [2, 8, 8, 16, 1, 8]]))

# table 8-15 via get_qpc (transform.rs:194-216): qPI < 30 -> qPI else QPCS[qPI-30]
vectors.append(dict(name="qpc_table", kind="qpc",
                    cases=[[0, 0, 0], [29, 0, 29], [30, 0, 29], [34, 0, 32], [39, 0, 35], [51, 0, 39],
                           [40, 12, 39], [10, -12, 0], [45, -3, 37]]))

out = os.path.join(os.path.dirname(os.path.abspath(__file__)), "kat_vectors.json")
with open(out, "w") as f:
    json.dump(dict(source="hand-derived from the cited reference lines; not reference output",
                   vectors=vectors), f, indent=1)
print("wrote", out, len(vectors), "vectors")
