"""Parity of the HIP reconstruction path against the CPU oracle, through the C ABI.

All tests here need a real MI355X (`-m gpu`). Bar: bit-exact (the path is integer-only).
Every call goes through libdryv_recon.so's extern "C" entry points (dryv_amd.frame.ReconContext);
the oracle is only the checker.
"""
import hashlib
import json
import os

import numpy as np
import pytest

import oracle
from dryv_amd import abi, synth, Frame, ReconError
from util import first_mismatch, make_coeffs, make_mb, split_planes, packed16_bound_batches, packed16_8x8_bound_batches

pytestmark = pytest.mark.gpu

with open(os.path.join(os.path.dirname(__file__), "golden", "kat_vectors.json")) as f:
    KAT = json.load(f)["vectors"]


def assert_parity(ctx, fp, n_frames, mbs, coeffs, expect_status=0):
    st, want = oracle.reconstruct(fp, n_frames, mbs, coeffs)
    assert st == expect_status
    got = ctx.reconstruct(fp, n_frames, mbs, coeffs, allow_unsupported=expect_status != 0)
    assert ctx.last_status == expect_status
    W, H = fp.pic_width_in_mbs, fp.pic_height_in_mbs
    assert np.array_equal(got, want), first_mismatch(got, want, W, H)
    return got


# ---- known-answer vectors straight through the GPU -------------------------------------------
FRAME_KATS = [v for v in KAT if "mbs" in v]


@pytest.mark.parametrize("v", FRAME_KATS, ids=[v["name"] for v in FRAME_KATS])
def test_kat_on_gpu(recon_ctx, v):
    from test_oracle_kat import build_frame_kat, check_luma
    fp, mbs, co = build_frame_kat(v)
    yuv = recon_ctx.reconstruct(fp, 1, mbs, co)
    Y, Cb, Cr = split_planes(yuv, v["W"], v["H"])
    check_luma(Y, v["expect"])
    assert np.all(Cb == v["expect"]["Cb"]) and np.all(Cr == v["expect"]["Cr"])


def test_kat_q1_quirk_on_gpu(recon_ctx):
    """K7 as a real two-macroblock picture: top MB reconstructs to 100 everywhere (Intra16x16 DC 128 plus
    a flat residual of -28: DC level -45 at qp 24 -> dcY = (-45*160 + 2) >> 2 = -1800, r = (-1800+32)>>6 = -28),
    bottom MB is Intra8x8 with blk0 vertical -> column 0 shows the reference's 75, not the spec's 100."""
    fp = abi.make_frame_params(1, 2, transform_8x8=True)
    mbs = np.array([make_mb(kind=2, qp=24, i16=2), make_mb(kind=1, qp=26, prev=0xE)], dtype=abi.MB_DESC_DTYPE)
    co = np.stack([make_coeffs({0: -45}), make_coeffs()])
    yuv = assert_parity(recon_ctx, fp, 1, mbs, co)
    Y, _, _ = split_planes(yuv, 1, 2)
    assert np.all(Y[:16] == 100)
    assert np.all(Y[16:24, 0] == 75) and np.all(Y[16:24, 1:8] == 100)


# ---- randomized parity at sizes the oracle finishes in seconds --------------------------------
CASES = [
    # name, W, H, frames, synth kwargs, frame-param kwargs
    ("i16_only", 7, 5, 3, dict(i4x4=0.0, i8x8=0.0), {}),
    ("i4x4_only", 7, 5, 3, dict(i4x4=1.0, i8x8=0.0), {}),
    ("i8x8_only", 7, 5, 3, dict(i4x4=0.0, i8x8=1.0), dict(transform_8x8=True)),
    ("c2_mix_small", 12, 9, 4, dict(i4x4=0.7, i8x8=0.0), {}),
    ("c3_mix_small", 12, 9, 4, dict(i4x4=0.35, i8x8=0.40), dict(transform_8x8=True)),
    ("single_mb", 1, 1, 5, dict(i4x4=0.4, i8x8=0.3), dict(transform_8x8=True)),
    ("single_row", 9, 1, 3, dict(i4x4=0.4, i8x8=0.3), dict(transform_8x8=True)),
    ("single_col", 1, 9, 3, dict(i4x4=0.4, i8x8=0.3), dict(transform_8x8=True)),
    ("two_cols", 2, 17, 2, dict(i4x4=0.4, i8x8=0.3), dict(transform_8x8=True)),
    ("all_qp", 10, 8, 3, dict(i4x4=0.4, i8x8=0.3, qp=(0, 51)), dict(transform_8x8=True)),
    ("dense_big_levels", 8, 6, 3, dict(i4x4=0.4, i8x8=0.3, coded=1.0, p0=0.9, decay4=0.97, decay8=0.99,
                                       max_level=2047, qp=(0, 51)), dict(transform_8x8=True)),
    ("illegal_modes_q4", 9, 7, 4, dict(i4x4=0.4, i8x8=0.3, legal_modes_only=False), dict(transform_8x8=True)),
    ("chroma_qp_offsets", 9, 7, 3, dict(i4x4=0.4, i8x8=0.3, qp=(0, 51)), dict(transform_8x8=True, cqo_cb=-7, cqo_cr=11)),
    ("zero_residual", 9, 7, 2, dict(i4x4=0.4, i8x8=0.3, coded=0.0), dict(transform_8x8=True)),
    ("dark_q2_zeros", 9, 7, 4, dict(i4x4=0.3, i8x8=0.2, coded=1.0, p0=0.6, max_level=300, qp=(30, 51)),
     dict(transform_8x8=True)),
]


@pytest.mark.parametrize("case", CASES, ids=[c[0] for c in CASES])
def test_random_parity(recon_ctx, case):
    name, W, H, frames, skw, fkw = case
    fp = abi.make_frame_params(W, H, **fkw)
    mbs, co = synth.generate(fp, synth.config(**skw), 100 + CASES.index(case), 0, frames)
    assert_parity(recon_ctx, fp, frames, mbs, co)


def test_fuzz_geometries_and_configs(recon_ctx):
    """60 seeded random (geometry, mix, qp range, chroma offsets, scaling lists, legality) combinations:
    ragged sizes (1..21 x 1..13 macroblocks: rows that are not a multiple of the 4-row band, widths below the
    top-right reach, single rows/columns), 1..3 frames."""
    rng = np.random.default_rng(20261004)
    for k in range(60):
        W, H, frames = int(rng.integers(1, 22)), int(rng.integers(1, 14)), int(rng.integers(1, 4))
        i4 = float(rng.choice([0.0, 0.3, 0.7, 1.0]))
        i8 = float(rng.choice([0.0, 0.3])) if i4 < 1.0 else 0.0
        i8 = min(i8, 1.0 - i4)
        lo = int(rng.integers(0, 40))
        flat = rng.random() < 0.6
        fkw = dict(transform_8x8=i8 > 0, cqo_cb=int(rng.integers(-12, 13)), cqo_cr=int(rng.integers(-12, 13)))
        if not flat:
            fkw.update(scaling4x4=rng.integers(4, 48, size=(6, 16)), scaling8x8=rng.integers(4, 48, size=(6, 64)))
        skw = dict(i4x4=i4, i8x8=i8, qp=(lo, int(rng.integers(lo, 52))), coded=float(rng.choice([0.2, 0.6, 1.0])),
                   max_level=int(rng.choice([15, 300, 2047])) if flat else 200,
                   legal_modes_only=bool(rng.random() < 0.7), prev_flag=float(rng.choice([0.1, 0.5, 0.9])))
        fp = abi.make_frame_params(W, H, **fkw)
        mbs, co = synth.generate(fp, synth.config(**skw), 1000 + k, k, frames)
        try:
            assert_parity(recon_ctx, fp, frames, mbs, co)
        except AssertionError as e:
            raise AssertionError("fuzz case %d (W=%d H=%d frames=%d %r %r): %s" % (k, W, H, frames, skw, {
                kk: vv for kk, vv in fkw.items() if not hasattr(vv, "shape")}, e))


def test_full_int16_coefficient_range(recon_ctx):
    """The FFI carries int16 coefficients; the kernel computes in int32 where the reference uses 64-bit isize.
    With flat lists and qp <= 24 every intermediate of |c| <= 32767 fits 32 bits, so the
    results must still be identical -- including residuals far outside [-255, 255], which the Intra4x4 path
    carries clamped."""
    rng = np.random.default_rng(9)
    fp = abi.make_frame_params(9, 7, transform_8x8=True)
    frames = 3
    mbs, co = synth.generate(fp, synth.config(i4x4=0.4, i8x8=0.3, coded=1.0, p0=0.9, decay4=0.97, decay8=0.99,
                                              qp=(0, 24)), 61, 0, frames)
    scale = rng.choice([1, 40, 700, 6000], size=(co.shape[0], 1))
    co = np.clip(co.astype(np.int64) * scale, -32768, 32767).astype(np.int16)
    assert np.abs(co.astype(np.int32)).max() == 32768 or np.abs(co.astype(np.int32)).max() >= 32767
    assert_parity(recon_ctx, fp, frames, mbs, co)


def test_nonflat_scaling_lists(recon_ctx):
    """Non-flat matrices: chroma re-uses the luma (list 0) LevelScale tables — quirk Q3."""
    rng = np.random.default_rng(5)
    s4 = rng.integers(8, 40, size=(6, 16))
    s8 = rng.integers(8, 40, size=(6, 64))
    fp = abi.make_frame_params(8, 6, transform_8x8=True, scaling4x4=s4, scaling8x8=s8)
    mbs, co = synth.generate(fp, synth.config(i4x4=0.4, i8x8=0.3, max_level=200), 55, 0, 3)
    assert_parity(recon_ctx, fp, 3, mbs, co)


def test_q2_chroma_zero_neighbours_occur(recon_ctx):
    """The 'dark' case must actually exercise quirk Q2 (a reconstructed chroma neighbour equal to 0)."""
    case = next(c for c in CASES if c[0] == "dark_q2_zeros")
    _, W, H, frames, skw, fkw = case
    fp = abi.make_frame_params(W, H, **fkw)
    mbs, co = synth.generate(fp, synth.config(**skw), 100 + CASES.index(case), 0, frames)
    yuv = recon_ctx.reconstruct(fp, frames, mbs, co)
    zeros = 0
    for f in range(frames):
        _, Cb, Cr = split_planes(yuv, W, H, f)
        zeros += int((Cb[:, 7::8] == 0).sum() + (Cr[:, 7::8] == 0).sum() + (Cb[7::8] == 0).sum())
    assert zeros > 0


def test_unsupported_record_status_and_zero_fill(recon_ctx):
    fp = abi.make_frame_params(4, 3)
    mbs, co = synth.generate(fp, synth.config(i4x4=0.5), 77, 0, 2)
    mbs = mbs.copy()
    mbs["mb_kind"][5] = 3      # I_PCM / inter: todo!() in the reference (frame/mod.rs:86,88)
    mbs["qp"][17] = 77
    assert_parity(recon_ctx, fp, 2, mbs, co, expect_status=abi.DRYV_E_UNSUPPORTED)
    # and the status does not stick to the next batch
    assert_parity(recon_ctx, fp, 2, *synth.generate(fp, synth.config(i4x4=0.5), 78, 0, 2))


def test_error_behaviour(recon_ctx):
    fp = abi.make_frame_params(2, 2)
    mbs, co = synth.generate(fp, synth.config(), 1, 0, 1)
    bad = abi.make_frame_params(2, 2)
    bad.chroma_array_type = 3   # todo!() at trans_chroma.rs:19-20
    with pytest.raises(ReconError) as e:
        recon_ctx.submit(bad, 1, mbs, co)
    assert e.value.status == abi.DRYV_E_UNSUPPORTED
    bad = abi.make_frame_params(2, 2)
    bad.bit_depth_y = 10
    with pytest.raises(ReconError) as e:
        recon_ctx.submit(bad, 1, mbs, co)
    assert e.value.status == abi.DRYV_E_UNSUPPORTED
    zero = abi.make_frame_params(0, 2)
    with pytest.raises(ReconError) as e:
        recon_ctx.submit(zero, 1, mbs, co)
    assert e.value.status == abi.DRYV_E_INVALID
    lib = abi.load_library()
    assert lib.dryv_recon_wait(recon_ctx._h, co.ctypes.data, 10) == abi.DRYV_E_STATE  # wait without submit
    assert_parity(recon_ctx, fp, 1, mbs, co)  # the context still works


def test_frame_mirror_matches_batch_api(recon_ctx, tmp_path):
    """Frame.new / decode (per macroblock) / write_to_yuv_file — the reference's call sequence."""
    fp = abi.make_frame_params(5, 4, transform_8x8=True)
    mbs, co = synth.generate(fp, synth.config(i4x4=0.4, i8x8=0.3), 9, 0, 1)
    frame = Frame.new(fp, recon_ctx)
    for mb, c in zip(mbs, co):
        frame.decode(mb, c)
    p = tmp_path / "yuv_frame"
    frame.write_to_yuv_file(str(p))
    st, want = oracle.reconstruct(fp, 1, mbs, co)
    data = np.fromfile(str(p), dtype=np.uint8)
    assert data.size == 5 * 4 * 384 and np.array_equal(data, want)
    with pytest.raises(ReconError) as e:
        Frame.new(fp, recon_ctx).decode(make_mb(kind=25), make_coeffs())
    assert e.value.status == abi.DRYV_E_UNSUPPORTED


def test_device_resident_path(recon_ctx):
    """submit_device: all buffers already in HBM (what bench.py times)."""
    import torch
    fp = abi.make_frame_params(10, 7, transform_8x8=True)
    mbs, co = synth.generate(fp, synth.config(i4x4=0.4, i8x8=0.3), 21, 0, 6)
    d_m = torch.from_numpy(mbs.view(np.uint8).reshape(-1)).cuda()
    d_c = torch.from_numpy(co).cuda()
    d_o = torch.zeros(6 * 70 * 384, dtype=torch.uint8, device="cuda")
    torch.cuda.synchronize()
    recon_ctx.submit_device(fp, 6, d_m.data_ptr(), d_c.data_ptr(), d_o.data_ptr())
    recon_ctx.sync()
    assert recon_ctx.last_kernel_ms() > 0
    st, want = oracle.reconstruct(fp, 6, mbs, co)
    assert np.array_equal(d_o.cpu().numpy(), want)


def test_handoff_records_across_launches_and_layouts(recon_ctx):
    """Bands hand their bottom lines over in tagged records of the workspace (tag = the launch's generation; the records are
    zeroed per workspace layout, never per launch). One context, launches alternating between geometries whose workspaces
    overlap differently (the records of one sit where the mode records or the records of another were), the same batch
    twice in a row (every tag of the previous launch is still there, one generation old), pictures of a single band (no
    hand-off at all) in between: every picture bit-exact."""
    seq = [(13, 9, 3, False), (5, 21, 2, True), (13, 9, 3, False), (13, 9, 3, False), (40, 4, 1, False), (7, 17, 1, True),
           (13, 9, 2, False), (5, 21, 2, True), (64, 5, 2, False), (13, 9, 3, False)]
    for k, (W, H, frames, t8) in enumerate(seq):
        fp = abi.make_frame_params(W, H, transform_8x8=t8)
        mbs, co = synth.generate(fp, synth.config(i4x4=0.5, i8x8=0.3 if t8 else 0.0), 4200 + (k % 4), 0, frames)
        assert_parity(recon_ctx, fp, frames, mbs, co)


@pytest.mark.parametrize("t8", [False, True], ids=["no8x8", "8x8"])
def test_mode_records_are_never_served_stale(recon_ctx, t8):
    """The one inter-workgroup hand-off that is a flag and not a data-tagged granule: a band's mode records (sc1 stores, drained,
    then the band's progress word; the band below polls the word and reads the records with sc1 loads, no acquire fence --
    DESIGN.md section 4.2 says which form of MI355X_MICROARCH.md that is and where it leaves the measured envelope: several
    workgroups per CU). Directed at exactly that: ONE context, TWO batches of identical geometry -- so every record of one
    launch sits at the address of the other's -- all Intra4x4 (Intra4x4 / Intra8x8 with the 8x8 transform) with different
    seeds, so that the bottom-row modes of every band differ between the two; tiny pictures of several bands, so that a
    launch's records are a few KB and hot in whatever cache served the launch before; sixteen launches alternating between
    the two, queued back to back on the stream. A mode record served from the previous launch derives the row below from the
    wrong neighbours: a wrong picture."""
    import torch
    W, H, frames = 6, 14, 3    # 4 bands (the last of 2 rows), 84 macroblocks: 2.6 KB of mode records per picture
    fp = abi.make_frame_params(W, H, transform_8x8=t8)
    batches = []
    for seed in (9100, 9101):
        mbs, co = synth.generate(fp, synth.config(i4x4=0.5 if t8 else 1.0, i8x8=0.5 if t8 else 0.0), seed, 0, frames)
        st, want = oracle.reconstruct(fp, frames, mbs, co)
        assert st == 0
        batches.append((torch.from_numpy(mbs.view(np.uint8).reshape(-1)).cuda(), torch.from_numpy(co).cuda(), want))
    per = frames * W * H * 384
    outs = [torch.zeros(per, dtype=torch.uint8, device="cuda") for _ in range(16)]
    torch.cuda.synchronize()
    for rep in range(3):
        for o in outs:
            o.zero_()
        torch.cuda.synchronize()
        for k, o in enumerate(outs):
            d_m, d_c, _ = batches[k & 1]
            recon_ctx.submit_device_queued(fp, frames, d_m.data_ptr(), d_c.data_ptr(), o.data_ptr())
        recon_ctx.sync()
        for k, o in enumerate(outs):
            assert np.array_equal(o.cpu().numpy(), batches[k & 1][2]), "launch %d of round %d" % (k, rep)


def test_queued_device_submits(recon_ctx):
    """dryv_recon_submit_device_queued: several batches behind each other on the stream, one sync (what bench.py times).
    Three different batches of different sizes into buffers of their own; then a queue in which the middle batch has
    coefficients beyond the 32-bit path's bound, so that the whole queue is re-run with the wide build at sync. Every
    picture must equal the oracle's, the launches are timed one by one, and nothing else may be submitted meanwhile."""
    import torch
    fp = abi.make_frame_params(11, 9)
    cases = []
    for k, (frames, big) in enumerate([(5, False), (4, True), (2, False)]):   # (the first batch sizes the queue's workspace)
        cfg = synth.config(i4x4=0.6, qp=(51, 51), coded=1.0) if big else synth.config(i4x4=0.6)
        mbs, co = synth.generate(fp, cfg, 300 + k, 0, frames)
        if big:
            co = np.where(np.arange(co.size).reshape(co.shape) % 2 == 0, 32767, -32768).astype(np.int16)
        cases.append((frames, mbs, co))
    for use in ([0, 2, 0], [0, 1, 2], [0, 2, 0, 1]):   # without / with the batch that needs the wide re-run (in the middle, last)
        ev0, nb0 = recon_ctx.wide_rerun_stats()
        bufs = []
        for k in use:
            frames, mbs, co = cases[k]
            d_m = torch.from_numpy(mbs.view(np.uint8).reshape(-1)).cuda()
            d_c = torch.from_numpy(co).cuda()
            d_o = torch.zeros(frames * 99 * 384, dtype=torch.uint8, device="cuda")
            bufs.append((k, d_m, d_c, d_o))
        torch.cuda.synchronize()
        for k, d_m, d_c, d_o in bufs:
            recon_ctx.submit_device_queued(fp, cases[k][0], d_m.data_ptr(), d_c.data_ptr(), d_o.data_ptr())
        with pytest.raises(ReconError) as e:   # a batch of another kind cannot cut in
            recon_ctx.submit_device(fp, cases[0][0], bufs[0][1].data_ptr(), bufs[0][2].data_ptr(), bufs[0][3].data_ptr())
        assert e.value.status == abi.DRYV_E_STATE
        recon_ctx.sync()
        avg, lo, hi = recon_ctx.kernel_ms_stats(len(bufs))
        assert 0 < lo <= avg <= hi
        # the queue is re-run from the first flagged batch on (status word 4 carries its position), not from its head
        ev1, nb1 = recon_ctx.wide_rerun_stats()
        assert (ev1 - ev0, nb1 - nb0) == ((1, len(use) - use.index(1)) if 1 in use else (0, 0))
        for k, d_m, d_c, d_o in bufs:
            frames, mbs, co = cases[k]
            st, want = oracle.reconstruct(fp, frames, mbs, co)
            assert st == 0 and np.array_equal(d_o.cpu().numpy(), want), "queued batch %d" % k


def test_queue_keeps_status_of_batches_not_rerun(recon_ctx):
    """A queue [batch with an unsupported record, ordinary batch, batch beyond the 32-bit path]: the wide re-run at sync
    starts at the third batch and clears the device's status words first -- what the first batch reported must survive
    it (DRYV_E_UNSUPPORTED, the record reconstructed as zero), and only one batch is run again."""
    import torch
    fp = abi.make_frame_params(11, 9)
    cases = []
    for k, kind in enumerate(["bad", "plain", "big"]):
        cfg = synth.config(i4x4=0.6, qp=(51, 51), coded=1.0) if kind == "big" else synth.config(i4x4=0.6)
        mbs, co = synth.generate(fp, cfg, 400 + k, 0, 3)
        if kind == "big":
            co = np.where(np.arange(co.size).reshape(co.shape) % 2 == 0, 32767, -32768).astype(np.int16)
        if kind == "bad":
            mbs = mbs.copy()
            mbs.view(np.uint8).reshape(-1, 16)[40, 0] = 7   # mb_kind 7: not a macroblock kind of this path
        cases.append((mbs, co))
    ev0, nb0 = recon_ctx.wide_rerun_stats()
    bufs = []
    for mbs, co in cases:
        bufs.append((torch.from_numpy(mbs.view(np.uint8).reshape(-1)).cuda(), torch.from_numpy(co).cuda(),
                     torch.zeros(3 * 99 * 384, dtype=torch.uint8, device="cuda")))
    torch.cuda.synchronize()
    for d_m, d_c, d_o in bufs:
        recon_ctx.submit_device_queued(fp, 3, d_m.data_ptr(), d_c.data_ptr(), d_o.data_ptr())
    with pytest.raises(ReconError) as e:
        recon_ctx.sync()
    assert e.value.status == abi.DRYV_E_UNSUPPORTED
    ev1, nb1 = recon_ctx.wide_rerun_stats()
    assert (ev1 - ev0, nb1 - nb0) == (1, 1)
    for (mbs, co), (d_m, d_c, d_o) in zip(cases[1:], bufs[1:]):   # the batches behind it are whole
        st, want = oracle.reconstruct(fp, 3, mbs, co)
        assert st == 0 and np.array_equal(d_o.cpu().numpy(), want)


@pytest.mark.parametrize("lanes", [2, 3, 4])
def test_queue_lanes_same_pictures_and_status(lanes):
    """dryv_recon_set_queue_lanes: queued batches rotate over several streams with half-size grids, two launches resident side
    by side. Seven batches of three sizes and two picture sizes' worth of workspace per lane, twice over the same context (the
    lanes' workspaces and generations carry over); then a queue with an unsupported record on one lane and a batch beyond the
    32-bit path on another: every picture equals the oracle's, the status is the OR over the lanes, the wide re-run starts at
    the flagged batch; then back to one lane."""
    import torch
    from dryv_amd.frame import ReconContext
    fp = abi.make_frame_params(13, 10)
    per = 130
    with ReconContext(0) as ctx:
        ctx.set_queue_lanes(lanes)
        for rnd in range(2):
            bufs = []
            for k in range(7):
                frames = (5, 3, 2)[k % 3]   # (the first batch sizes the queue's workspaces: every lane's)
                mbs, co = synth.generate(fp, synth.config(i4x4=0.5), 900 + 10 * rnd + k, 0, frames)
                bufs.append((frames, mbs, co, torch.from_numpy(mbs.view(np.uint8).reshape(-1)).cuda(), torch.from_numpy(co).cuda(),
                             torch.zeros(frames * per * 384, dtype=torch.uint8, device="cuda")))
            torch.cuda.synchronize()
            for frames, mbs, co, d_m, d_c, d_o in bufs:
                ctx.submit_device_queued(fp, frames, d_m.data_ptr(), d_c.data_ptr(), d_o.data_ptr())
            ctx.sync()
            avg, lo, hi = ctx.kernel_ms_stats(len(bufs))
            assert 0 < lo <= avg <= hi
            for k, (frames, mbs, co, d_m, d_c, d_o) in enumerate(bufs):
                st, want = oracle.reconstruct(fp, frames, mbs, co)
                assert st == 0 and np.array_equal(d_o.cpu().numpy(), want), "round %d batch %d" % (rnd, k)
        # status over the lanes; the wide re-run
        cases = []
        for k, kind in enumerate(["plain", "bad", "plain", "big", "plain"]):
            cfg = synth.config(i4x4=0.6, qp=(51, 51), coded=1.0) if kind == "big" else synth.config(i4x4=0.6)
            mbs, co = synth.generate(fp, cfg, 950 + k, 0, 3)
            if kind == "big":
                co = np.where(np.arange(co.size).reshape(co.shape) % 2 == 0, 32767, -32768).astype(np.int16)
            if kind == "bad":
                mbs = mbs.copy()
                mbs.view(np.uint8).reshape(-1, 16)[40, 0] = 7
            cases.append((kind, mbs, co, torch.from_numpy(mbs.view(np.uint8).reshape(-1)).cuda(), torch.from_numpy(co).cuda(),
                          torch.zeros(3 * per * 384, dtype=torch.uint8, device="cuda")))
        torch.cuda.synchronize()
        ev0, nb0 = ctx.wide_rerun_stats()
        for kind, mbs, co, d_m, d_c, d_o in cases:
            ctx.submit_device_queued(fp, 3, d_m.data_ptr(), d_c.data_ptr(), d_o.data_ptr())
        with pytest.raises(ReconError) as e:
            ctx.sync()
        assert e.value.status == abi.DRYV_E_UNSUPPORTED
        ev1, nb1 = ctx.wide_rerun_stats()
        assert (ev1 - ev0, nb1 - nb0) == (1, 2)   # the flagged batch and the one behind it
        for kind, mbs, co, d_m, d_c, d_o in cases:
            if kind == "bad":
                continue
            st, want = oracle.reconstruct(fp, 3, mbs, co)
            assert st == 0 and np.array_equal(d_o.cpu().numpy(), want), kind
        # one lane again
        ctx.set_queue_lanes(1)
        kind, mbs, co, d_m, d_c, d_o = cases[0]
        d_o.zero_()
        ctx.submit_device_queued(fp, 3, d_m.data_ptr(), d_c.data_ptr(), d_o.data_ptr())
        ctx.sync()
        st, want = oracle.reconstruct(fp, 3, mbs, co)
        assert np.array_equal(d_o.cpu().numpy(), want)
        with pytest.raises(ReconError) as e:
            ctx.set_queue_lanes(5)
        assert e.value.status == abi.DRYV_E_INVALID


def test_queue_cannot_outgrow_its_workspace():
    """A queued batch that needs a larger workspace than the queue is running on is refused (DRYV_E_STATE: sync first), not
    launched: a context of its own, so that the workspace is known to be the first batch's; every buffer is full size."""
    import torch
    from dryv_amd.frame import ReconContext
    small, large = abi.make_frame_params(6, 5), abi.make_frame_params(40, 30)
    with ReconContext(0) as ctx:
        bufs = []
        for fp, frames in ((small, 2), (large, 8)):
            mbs, co = synth.generate(fp, synth.config(), 77, 0, frames)
            per = fp.pic_width_in_mbs * fp.pic_height_in_mbs
            bufs.append((fp, frames, mbs, co, torch.from_numpy(mbs.view(np.uint8).reshape(-1)).cuda(), torch.from_numpy(co).cuda(),
                         torch.zeros(frames * per * 384, dtype=torch.uint8, device="cuda")))
        torch.cuda.synchronize()
        fp, frames, mbs, co, d_m, d_c, d_o = bufs[0]
        ctx.submit_device_queued(fp, frames, d_m.data_ptr(), d_c.data_ptr(), d_o.data_ptr())
        fpl, fl, mbl, col, d_ml, d_cl, d_ol = bufs[1]
        with pytest.raises(ReconError) as e:
            ctx.submit_device_queued(fpl, fl, d_ml.data_ptr(), d_cl.data_ptr(), d_ol.data_ptr())
        assert e.value.status == abi.DRYV_E_STATE
        ctx.sync()
        st, want = oracle.reconstruct(fp, frames, mbs, co)
        assert np.array_equal(d_o.cpu().numpy(), want)
        ctx.submit_device_queued(fpl, fl, d_ml.data_ptr(), d_cl.data_ptr(), d_ol.data_ptr())   # after the sync it may grow
        ctx.sync()
        st, want = oracle.reconstruct(fpl, fl, mbl, col)
        assert np.array_equal(d_ol.cpu().numpy(), want)


# ---- BASELINE.json full sizes: golden digest of one frame + size-independent properties --------
def _digest(a):
    return hashlib.sha256(a.tobytes()).hexdigest()


@pytest.mark.parametrize("wl", ["C2_1080p_intra_4x4", "C3_4k_intra_8x8"])
def test_full_size_frames_match_oracle(recon_ctx, wl):
    fp, mbs, co, n = synth.workload(wl, n_frames=2, first_frame=3)
    assert_parity(recon_ctx, fp, n, mbs, co)


def test_full_batch_properties_c2(recon_ctx):
    """300-frame 1080p batch (config 2): frames are independent, so
    (a) the batch result equals per-frame results (checked on a sample against the oracle),
    (b) re-running is idempotent, (c) permuting frames permutes the output."""
    fp, mbs, co, n = synth.workload("C2_1080p_intra_4x4", n_frames=300)
    W, H = 120, 68
    per = W * H
    out1 = recon_ctx.reconstruct(fp, n, mbs, co)
    out2 = recon_ctx.reconstruct(fp, n, mbs, co)
    assert _digest(out1) == _digest(out2)
    fb = 384 * per
    for f in (0, 149, 299):
        st, want = oracle.reconstruct(fp, 1, mbs[f * per:(f + 1) * per], co[f * per:(f + 1) * per])
        got = out1[f * fb:(f + 1) * fb]
        assert np.array_equal(got, want), first_mismatch(got, want, W, H)
    perm = np.arange(n)[::-1]
    idx = (perm[:, None] * per + np.arange(per)[None, :]).ravel()
    out3 = recon_ctx.reconstruct(fp, n, mbs[idx], co[idx])
    assert np.array_equal(out3.reshape(n, fb), out1.reshape(n, fb)[perm])


def test_oversubscribed_grid_completes(tmp_path):
    """More workgroups than the device keeps resident (DRYV_RECON_GRID=1100 x 2 bands > 1024 resident x 2): every
    band is claimed from the one queue, so late-starting workgroups can never hold a band others wait for.
    Runs in a child process under a timeout: a scheduling deadlock must fail the test, not hang the suite."""
    import subprocess
    import sys
    code = (
        "import numpy as np, sys\n"
        "sys.path.insert(0, %r)\n"
        "import oracle\n"
        "from dryv_amd import synth, ReconContext\n"
        "fp, mbs, co, n = synth.workload('C2_1080p_intra_4x4', n_frames=300)\n"
        "got = ReconContext(0).reconstruct(fp, n, mbs, co)\n"
        "per = 8160\n"
        "st, want = oracle.reconstruct(fp, 4, mbs[-4 * per:], co[-4 * per:])\n"
        "assert st == 0 and np.array_equal(got[-4 * per * 384:], want)\n"
        "print('ok')\n" % os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    for grid in ("1100", "3000", "7"):
        env = dict(os.environ, DRYV_RECON_GRID=grid)
        r = subprocess.run([sys.executable, "-c", code], env=env, capture_output=True, text=True, timeout=180)
        assert r.returncode == 0 and "ok" in r.stdout, (grid, r.stdout[-500:], r.stderr[-2000:])


def test_two_contexts_in_flight(recon_ctx):
    """Two contexts (own stream, staging buffers and workspace each) with submits in flight at the same time."""
    from dryv_amd import ReconContext
    other = ReconContext(0)
    try:
        fpa = abi.make_frame_params(40, 23, transform_8x8=True)
        fpb = abi.make_frame_params(33, 18)
        ma, ca = synth.generate(fpa, synth.config(i4x4=0.4, i8x8=0.3), 901, 0, 6)
        mb_, cb = synth.generate(fpb, synth.config(i4x4=0.7, i8x8=0.0), 902, 0, 9)
        for _ in range(3):
            recon_ctx.submit(fpa, 6, ma, ca)
            other.submit(fpb, 9, mb_, cb)
            ga = recon_ctx.wait()
            gb = other.wait()
        assert np.array_equal(ga, oracle.reconstruct(fpa, 6, ma, ca)[1])
        assert np.array_equal(gb, oracle.reconstruct(fpb, 9, mb_, cb)[1])
    finally:
        other.close()


@pytest.mark.parametrize("qp_range", [(0, 24), (25, 40), (41, 51)])
def test_full_int16_range_every_qp(recon_ctx, qp_range):
    """The FFI carries int16 coefficients and scaling weights up to 255; the reference computes in 64-bit isize. The fast
    builds of the kernel compute in int32 (or packed 16 bits) and flag a macroblock beyond the bound under which that is
    provably the same (per block and qp); the library then re-runs the batch with the kernel's 64-bit build before
    reporting: bit-exact at every qp, with and without the 8x8 transform."""
    rng = np.random.default_rng(9 + qp_range[0])
    s4, s8 = rng.integers(1, 256, size=(6, 16)), rng.integers(1, 256, size=(6, 64))
    for lists, cfg in ((dict(), dict(i4x4=0.6, i8x8=0.0)), (dict(scaling4x4=s4), dict(i4x4=0.6, i8x8=0.0)),
                       (dict(transform_8x8=True), dict(i4x4=0.3, i8x8=0.5)),
                       (dict(transform_8x8=True, scaling4x4=s4, scaling8x8=s8), dict(i4x4=0.3, i8x8=0.5))):
        fp = abi.make_frame_params(9, 7, **lists)
        mbs, co = synth.generate(fp, synth.config(coded=1.0, p0=0.9, decay4=0.97, decay8=0.99, qp=qp_range, **cfg), 61, 0, 3)
        scale = rng.choice([1, 40, 700, 6000], size=(co.shape[0], 1))
        co = np.clip(co.astype(np.int64) * scale, -32768, 32767).astype(np.int16)
        assert_parity(recon_ctx, fp, 3, mbs, co)


def test_fast_and_wide_builds_agree():
    """The two builds of the band kernel -- the fast one (int32 / packed 16-bit residual arithmetic; flags what it cannot do)
    and the WIDE one (64-bit fallback: what the library re-runs a flagged batch with) -- on the same ordinary batches, with
    and without the 8x8 transform. The WIDE build is forced through DRYV_RECON_FORCE_WIDE (a test hook) in a child
    process; the digests over all pictures must be equal."""
    import subprocess
    import sys
    code = (
        "import numpy as np, sys, hashlib\n"
        "sys.path.insert(0, %r)\n"
        "from dryv_amd import abi, synth, ReconContext\n"
        "ctx = ReconContext(0)\n"
        "rng = np.random.default_rng(5)\n"
        "h = hashlib.sha256()\n"
        "for k in range(12):\n"
        "    W, H, frames = int(rng.integers(1, 40)), int(rng.integers(1, 14)), int(rng.integers(1, 6))\n"
        "    t8 = k %% 3 != 0\n"
        "    fp = abi.make_frame_params(W, H, transform_8x8=t8)\n"
        "    mbs, co = synth.generate(fp, synth.config(i4x4=0.4, i8x8=0.35 if t8 else 0.0, legal_modes_only=bool(k & 1)), 500 + k, k, frames)\n"
        "    h.update(ctx.reconstruct(fp, frames, mbs, co).tobytes())\n"
        "print('digest', h.hexdigest())\n" % os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    out = {}
    for wide in ("0", "1"):
        env = dict(os.environ, DRYV_RECON_FORCE_WIDE=wide)
        r = subprocess.run([sys.executable, "-c", code], env=env, capture_output=True, text=True, timeout=300)
        assert r.returncode == 0 and "digest" in r.stdout, (wide, r.stdout[-500:], r.stderr[-3000:])
        out[wide] = r.stdout.split("digest")[1].strip()
    assert out["0"] == out["1"]


def test_full_batch_properties_c3(recon_ctx):
    """BASELINE.json configs[2] at full size: 100 frames of 240x135 macroblocks, 8x8 transform enabled. Too large for
    the oracle as a whole: idempotence, frame permutation, and three sampled frames against the oracle. The batch must
    contain every one of the 9 Intra4x4, 9 Intra8x8, 4 Intra16x16 and 4 chroma modes in at least 1 % of the respective
    macroblocks / blocks (configs[4]'s requirement), counted on the oracle's derived modes of a sampled frame."""
    fp, mbs, co, n = synth.workload("C3_4k_intra_8x8", n_frames=100)
    W, H = 240, 135
    per = W * H
    fb = 384 * per
    out1 = recon_ctx.reconstruct(fp, n, mbs, co)
    out2 = recon_ctx.reconstruct(fp, n, mbs, co)
    assert _digest(out1) == _digest(out2)
    for f in (0, 57, 99):
        st, want, modes = oracle.reconstruct(fp, 1, mbs[f * per:(f + 1) * per], co[f * per:(f + 1) * per], want_modes=True)
        got = out1[f * fb:(f + 1) * fb]
        assert st == 0 and np.array_equal(got, want), first_mismatch(got, want, W, H)
    m = mbs[57 * per:58 * per]
    kind = m["mb_kind"]
    for k, nm, cols in ((0, 9, slice(0, 16)), (1, 9, slice(16, 20))):
        derived = modes[kind == k][:, cols].ravel()
        share = np.bincount(derived.astype(np.int64), minlength=nm) / derived.size
        assert share.min() >= 0.01, (k, share)
    i16 = m["i16_pred_mode"][kind == 2]
    assert (np.bincount(i16, minlength=4) / i16.size).min() >= 0.01
    cm = m["intra_chroma_pred_mode"]
    assert (np.bincount(cm, minlength=4) / cm.size).min() >= 0.01
    perm = np.arange(n)[::-1]
    idx = (perm[:, None] * per + np.arange(per)[None, :]).ravel()
    out3 = recon_ctx.reconstruct(fp, n, mbs[idx], co[idx])
    assert np.array_equal(out3.reshape(n, fb), out1.reshape(n, fb)[perm])


def test_pipelined_host_path_chunk_boundaries(recon_ctx):
    """dryv_recon_submit_host: chunks of frames on three streams (copy-in, reconstruction, copy-out), page-locked and
    pageable buffers, chunk sizes that do and do not divide the batch -- always the oracle's planes."""
    fp = abi.make_frame_params(23, 11)
    frames = 13
    mbs, co = synth.generate(fp, synth.config(i4x4=0.6, i8x8=0.0), 77, 0, frames)
    st, want = oracle.reconstruct(fp, frames, mbs, co)
    assert st == 0
    pm = recon_ctx.alloc_host(mbs.shape, mbs.dtype)
    pc = recon_ctx.alloc_host(co.shape, co.dtype)
    po = recon_ctx.alloc_host(want.shape, np.uint8)
    pm[...] = mbs
    pc[...] = co
    try:
        for chunk in ("1", "4", "5", "13", "64"):
            os.environ["DRYV_RECON_CHUNK_FRAMES"] = chunk
            po[...] = 0
            recon_ctx.submit_host(fp, frames, pm, pc, po)
            recon_ctx.sync()
            assert np.array_equal(po, want), (chunk, first_mismatch(po, want, 23, 11))
            out = np.zeros_like(want)                     # pageable buffers
            recon_ctx.submit_host(fp, frames, mbs, co, out)
            recon_ctx.sync()
            assert np.array_equal(out, want), chunk
        # a batch with blocks beyond int32 (wide re-run) and an 8x8 stream through the same path
        os.environ["DRYV_RECON_CHUNK_FRAMES"] = "3"
        big = np.clip(co.astype(np.int64) * 6000, -32768, 32767).astype(np.int16)
        mb51 = mbs.copy()
        mb51["qp"] = 51
        st, want_big = oracle.reconstruct(fp, frames, mb51, big)
        out = np.zeros_like(want)
        recon_ctx.submit_host(fp, frames, mb51, big, out)
        recon_ctx.sync()
        assert st == 0 and np.array_equal(out, want_big)
        fp8 = abi.make_frame_params(9, 6, transform_8x8=True)
        m8, c8 = synth.generate(fp8, synth.config(i4x4=0.3, i8x8=0.4), 78, 0, 7)
        st, want8 = oracle.reconstruct(fp8, 7, m8, c8)
        out8 = np.zeros_like(want8)
        recon_ctx.submit_host(fp8, 7, m8, c8, out8)
        recon_ctx.sync()
        assert st == 0 and np.array_equal(out8, want8)
    finally:
        os.environ.pop("DRYV_RECON_CHUNK_FRAMES", None)
        for a in (pm, pc, po):
            recon_ctx.free_host(a)


def _expected_packed(yuv, W, H, frames, fmt, crop):
    """The output stage restated in numpy: crop the oracle's full planes, then I420 or NV12 byte order."""
    l, r, t, b = crop
    out = []
    per = W * H * 384
    for f in range(frames):
        Y, Cb, Cr = split_planes(yuv[f * per:(f + 1) * per], W, H)
        Y = Y[t:16 * H - b, l:16 * W - r]
        Cb = Cb[t // 2:8 * H - b // 2, l // 2:8 * W - r // 2]
        Cr = Cr[t // 2:8 * H - b // 2, l // 2:8 * W - r // 2]
        out.append(Y.reshape(-1))
        if fmt == abi.OUT_NV12:
            out.append(np.stack([Cb, Cr], axis=-1).reshape(-1))
        else:
            out += [Cb.reshape(-1), Cr.reshape(-1)]
    return np.concatenate(out)


@pytest.mark.parametrize("geo", [(7, 5, 2, (0, 0, 0, 0)), (7, 5, 2, (2, 6, 4, 10)), (12, 9, 1, (0, 0, 0, 8)),
                                 (3, 2, 3, (14, 16, 2, 0)), (1, 1, 1, (0, 2, 0, 2)), (20, 4, 1, (6, 0, 30, 0))])
def test_output_stage_crop_and_nv12(recon_ctx, geo):
    """cropping and NV12 packing on the device. The reference has no such stage (it parses the SPS's
    cropping rectangle and ignores it), so the expectation is the stage's definition applied to the oracle's planes:
    rows / columns sliced, chroma at half the offsets, NV12 = Cb and Cr interleaved. Through both entry points."""
    import torch
    W, H, frames, crop = geo
    fp = abi.make_frame_params(W, H)
    mbs, co = synth.generate(fp, synth.config(i4x4=0.6, i8x8=0.0), 777 + W, 0, frames)
    st, full = oracle.reconstruct(fp, frames, mbs, co)
    assert st == 0
    for fmt in (abi.OUT_I420, abi.OUT_NV12):
        od = abi.make_output_desc(fmt, crop)
        want = _expected_packed(full, W, H, frames, fmt, crop)
        recon_ctx.submit(fp, frames, mbs, co)
        got = recon_ctx.wait_packed(od)
        assert got.size == want.size and np.array_equal(got, want), (fmt, int(np.flatnonzero(got != want)[0]))
        # device-resident: reconstruct, sync, pack, sync
        d_m = torch.from_numpy(mbs.view(np.uint8).reshape(-1)).cuda()
        d_c = torch.from_numpy(co).cuda()
        d_y = torch.zeros(full.size, dtype=torch.uint8, device="cuda")
        d_o = torch.zeros(want.size, dtype=torch.uint8, device="cuda")
        recon_ctx.submit_device(fp, frames, d_m.data_ptr(), d_c.data_ptr(), d_y.data_ptr())
        with pytest.raises(ReconError):                       # not while the batch is in flight
            recon_ctx.pack_device(fp, frames, d_y.data_ptr(), od, d_o.data_ptr())
        recon_ctx.sync()
        recon_ctx.pack_device(fp, frames, d_y.data_ptr(), od, d_o.data_ptr())
        recon_ctx.sync()
        assert np.array_equal(d_o.cpu().numpy(), want)
    # an invalid description leaves the batch in flight for a plain wait
    recon_ctx.submit(fp, frames, mbs, co)
    with pytest.raises(ReconError):
        recon_ctx.wait_packed(abi.make_output_desc(abi.OUT_I420, (0, 0, 16 * H, 0)))
    recon_ctx._keep = (mbs, co, fp)
    assert np.array_equal(recon_ctx.wait(), full)


def test_exactness_bound_adversarial(recon_ctx):
    """Every coefficient exactly at (and one above) the per-qp bound of the band kernel's 32-bit path, with the sign patterns
    that maximise the butterflies' growth; streams with the 8x8 transform likewise (their 8x8 blocks against the bound of
    the 8-point passes). All must equal the oracle's 64-bit arithmetic bit for bit."""
    V4 = np.array([[10, 16, 13], [11, 18, 14], [13, 20, 16], [14, 23, 18], [16, 25, 20], [18, 29, 23]])
    V8 = np.array([[20, 18, 32, 19, 25, 24], [22, 19, 35, 21, 28, 26], [26, 23, 42, 24, 33, 31], [28, 25, 45, 26, 35, 33],
                   [32, 28, 51, 30, 40, 38], [36, 32, 58, 34, 46, 43]])
    rng = np.random.default_rng(4)
    for t8 in (False, True):
        fp = abi.make_frame_params(6, 5, transform_8x8=t8)
        for qp in (0, 23, 24, 35, 36, 41, 51):
            qd, qm = qp // 6, qp % 6
            thr4 = min((1 << 26) // ((16 * V4[qm].max()) << max(qd - 4, 0)), 32767)
            thr8 = min((1 << 23) // ((16 * V8[qm].max()) << max(qd - 6, 0)), 32767)
            for bump in (0, 1):
                cfg = synth.config(i4x4=0.4, i8x8=0.4 if t8 else 0.0, coded=1.0, qp=(qp, qp))
                mbs, co = synth.generate(fp, cfg, 900 + qp, 0, 2)
                co = co.astype(np.int64)
                for a in range(co.shape[0]):
                    lim = min((thr8 if mbs["mb_kind"][a] == 1 else thr4) + bump, 32767)
                    pat = rng.integers(0, 3)
                    sign = np.ones(384, dtype=np.int64) if pat == 0 else (np.where(np.arange(384) % 2, -1, 1) if pat == 1
                                                                        else rng.choice([-1, 1], size=384))
                    co[a] = sign * lim
                assert_parity(recon_ctx, fp, 2, mbs, co.astype(np.int16))


@pytest.mark.gpu
def test_packed16_residual_path_at_its_bound(recon_ctx):
    """The band kernel's packed 16-bit residual path: every luma block's sum |c| at the bound that admits it, and one
    above (the step then takes the 32-bit path). Bit-exact against the oracle's 64-bit arithmetic either way."""
    fp = abi.make_frame_params(9, 6)
    for qp, bump, mbs, co in packed16_bound_batches(fp, synth, frames=2):
        assert_parity(recon_ctx, fp, 2, mbs, co)


@pytest.mark.gpu
def test_packed16_8x8_residual_path_at_its_bound(recon_ctx):
    """The packed 16-bit form of the 8x8 residual: every 8x8 block's sum |c| at the per-qp bound that admits it (T_THR8P),
    and one above (the step then takes the 32-bit passes). Bit-exact against the oracle's 64-bit arithmetic either way."""
    fp = abi.make_frame_params(9, 6, transform_8x8=True)
    for qp, bump, mbs, co in packed16_8x8_bound_batches(fp, synth, frames=2):
        assert_parity(recon_ctx, fp, 2, mbs, co)
