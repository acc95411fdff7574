"""The synthetic batch generator: deterministic, spec-legal modes, stated mix."""
import numpy as np

import oracle
from dryv_amd import abi, synth


def test_deterministic_and_frame_addressable():
    fp = abi.make_frame_params(6, 4)
    cfg = synth.config(i4x4=0.5, i8x8=0.3)
    a_m, a_c = synth.generate(fp, cfg, 7, 0, 3)
    b_m, b_c = synth.generate(fp, cfg, 7, 0, 3)
    assert a_m.tobytes() == b_m.tobytes() and a_c.tobytes() == b_c.tobytes()
    # frame k of a batch == a batch that starts at frame k (what per-rank sharding relies on)
    c_m, c_c = synth.generate(fp, cfg, 7, 2, 1)
    n = 24
    assert c_m.tobytes() == a_m[2 * n:].tobytes() and c_c.tobytes() == a_c[2 * n:].tobytes()
    d_m, _ = synth.generate(fp, cfg, 8, 0, 3)
    assert d_m.tobytes() != a_m.tobytes()


def test_mix_and_ranges_c2():
    fp, mbs, co, n = synth.workload("C2_1080p_intra_4x4", n_frames=1)
    assert n == 1 and mbs.size == 8160 and co.shape == (8160, 384)
    kinds = np.bincount(mbs["mb_kind"], minlength=3) / mbs.size
    assert abs(kinds[0] - 0.7) < 0.03 and kinds[1] == 0 and abs(kinds[2] - 0.3) < 0.03
    assert mbs["qp"].min() >= 20 and mbs["qp"].max() <= 40
    assert np.abs(co.astype(np.int32)).max() <= 2047
    assert set(np.unique(mbs["intra_chroma_pred_mode"])) == {0, 1, 2, 3}


def test_every_mode_occurs_c3():
    fp, mbs, co, n = synth.workload("C3_4k_intra_8x8", n_frames=1)
    st, _, modes = oracle.reconstruct(fp, 1, mbs, co, want_modes=True)
    assert st == 0
    m4 = np.bincount(modes[mbs["mb_kind"] == 0][:, :16].ravel(), minlength=9)
    m8 = np.bincount(modes[mbs["mb_kind"] == 1][:, 16:].ravel(), minlength=9)
    assert (m4 / m4.sum()).min() > 0.01 and (m8 / m8.sum()).min() > 0.01
    i16 = np.bincount(mbs["i16_pred_mode"][mbs["mb_kind"] == 2], minlength=4)
    assert (i16 / i16.sum()).min() > 0.01
    ch = np.bincount(mbs["intra_chroma_pred_mode"], minlength=4)
    assert (ch / ch.sum()).min() > 0.01


def test_legal_modes_never_hit_missing_samples():
    """With legal_modes_only the picture's first row/column only uses modes whose samples exist:
    the derived modes must respect the availability rules of pred4x4.rs:92-359."""
    fp = abi.make_frame_params(5, 3, transform_8x8=True)
    mbs, co = synth.generate(fp, synth.config(i4x4=0.5, i8x8=0.4), 11, 0, 4)
    st, _, modes = oracle.reconstruct(fp, 4, mbs, co, want_modes=True)
    assert st == 0
    need_top = {0, 3, 7, 4, 5, 6}
    need_left = {1, 8, 4, 5, 6}
    W, H = 5, 3
    for i, (mb, md) in enumerate(zip(mbs, modes)):
        a = i % (W * H)
        mx, my = a % W, a // W
        if mb["mb_kind"] == 0:
            for b in range(16):
                bx = ((b >> 1) & 2) | (b & 1)
                by = ((b >> 2) & 2) | ((b >> 1) & 1)
                if my == 0 and by == 0:
                    assert md[b] not in need_top
                if mx == 0 and bx == 0:
                    assert md[b] not in need_left
        elif mb["mb_kind"] == 1:
            for b in range(4):
                if my == 0 and b < 2:
                    assert md[16 + b] not in need_top
                if mx == 0 and (b & 1) == 0:
                    assert md[16 + b] not in need_left
