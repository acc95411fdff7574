"""The in-loop deblocking filter: the C oracle (oracle/dryv_deblock.c) against a second restatement of
H.264 clause 8.7 (oracle/deblock_model.py) and hand-derived known answers, on CPU; the HIP kernel against the oracle on
the GPU. The reference has no deblocking (README.md:15 unchecked): nothing here is parity with dryv, and neither
restatement is pinned to an independent decoder."""
import numpy as np
import pytest

import oracle
from oracle import deblock_model
from dryv_amd import abi, synth


def _picture(W, H, frames, seed, t8=False, **cfg):
    fp = abi.make_frame_params(W, H, transform_8x8=t8, cqo_cb=cfg.pop("cqo_cb", 0), cqo_cr=cfg.pop("cqo_cr", None))
    mbs, co = synth.generate(fp, synth.config(**cfg), seed, 0, frames)
    st, yuv = oracle.reconstruct(fp, frames, mbs, co)
    assert st == 0
    return fp, mbs, yuv


def test_known_answers():
    """Hand-derived from 8.7.2: two macroblocks side by side, left all 100, right all 120, qp 30 both.
    indexA = indexB = 30: alpha = 25, beta = 8. Macroblock edge, bS = 4: |p0-q0| = 20 < 25 and the flat sides pass beta;
    ap = aq = 0 < beta but |p0-q0| = 20 is not < (25 >> 2) + 2 = 8, so only p0/q0 change: p0' = (2*100+100+120+2)>>2 = 105,
    q0' = (2*120+120+100+2)>>2 = 115. Everything else stays. Chroma planes (both 128) stay. With qp 20 (indexA 20:
    alpha = 7) the edge is not filtered at all."""
    fp = abi.make_frame_params(2, 1)
    for qp, expect in ((30, (105, 115)), (20, (100, 120))):
        mbs = np.zeros(2, dtype=abi.MB_DESC_DTYPE)
        mbs["mb_kind"], mbs["qp"] = 2, qp
        yuv = np.full(2 * 384, 128, dtype=np.uint8)
        Y = yuv[:512].reshape(16, 32)
        Y[:, :16], Y[:, 16:] = 100, 120
        st, out = oracle.deblock(fp, abi.make_deblock_params(), 1, mbs, yuv)
        assert st == 0
        Yo = out[:512].reshape(16, 32)
        assert np.all(Yo[:, 15] == expect[0]) and np.all(Yo[:, 16] == expect[1])
        assert np.all(Yo[:, :15] == 100) and np.all(Yo[:, 17:] == 120) and np.all(out[512:] == 128)
    # bS = 3 inside a macroblock: a step of 4 at x = 4, qp 40: indexA 40: alpha 80, beta 13, tc0 = 7; ap, aq < beta -> tc = 9
    # delta = clip(((4 << 2) + 0 + 4) >> 3) = 2 -> p0 102, q0 102; p1' = p1 + clip((p2 + ((100+104+1)>>1) - 2 p1) >> 1) = 100 + 1,
    # q1' = 104 - 1. The next edge (x = 8) then sees p2 p1 p0 = 103 104 104 | 104...: delta 0, but ap = 1 < beta moves p1 by
    # clip((103 + 104 - 208) >> 1) = -1 -> x = 6 becomes 103.
    mbs = np.zeros(1, dtype=abi.MB_DESC_DTYPE)
    mbs["mb_kind"], mbs["qp"] = 0, 40
    yuv = np.full(384, 128, dtype=np.uint8)
    Y = yuv[:256].reshape(16, 16)
    Y[:, :4], Y[:, 4:] = 100, 104
    st, out = oracle.deblock(abi.make_frame_params(1, 1), abi.make_deblock_params(), 1, mbs, yuv)
    Yo = out[:256].reshape(16, 16)
    assert list(Yo[0, :10]) == [100, 100, 101, 102, 102, 103, 103, 104, 104, 104]
    # disable_deblocking_filter_idc = 1: untouched; out-of-range offsets are refused
    st, out = oracle.deblock(abi.make_frame_params(1, 1), abi.make_deblock_params(disable_idc=1), 1, mbs, yuv)
    assert st == 0 and np.array_equal(out, yuv)
    assert oracle.deblock(abi.make_frame_params(1, 1), abi.make_deblock_params(alpha_div2=7), 1, mbs, yuv)[0] == abi.DRYV_E_INVALID


@pytest.mark.parametrize("case", [(5, 4, False, 0, 0, 0, 0), (6, 3, True, 0, 0, 0, 0), (4, 5, True, -3, 5, 2, -1),
                                  (7, 2, False, 6, -6, -6, 6), (1, 1, True, 0, 0, 3, 3), (3, 3, False, 0, 0, 0, 0)])
def test_oracle_matches_second_restatement(case):
    W, H, t8, a2, b2, cb, cr = case
    frames = 2
    fp, mbs, yuv = _picture(W, H, frames, 17 + W, t8=t8, i4x4=0.4, i8x8=0.3 if t8 else 0.0, qp=(10, 51), cqo_cb=cb, cqo_cr=cr)
    dp = abi.make_deblock_params(0, a2, b2)
    st, got = oracle.deblock(fp, dp, frames, mbs, yuv)
    assert st == 0
    per = W * H
    changed = 0
    for f in range(frames):
        want = deblock_model.deblock(W, H, mbs["qp"][f * per:(f + 1) * per], mbs["mb_kind"][f * per:(f + 1) * per],
                                     yuv[f * per * 384:(f + 1) * per * 384], cb, cr, 0, a2, b2)
        assert np.array_equal(got[f * per * 384:(f + 1) * per * 384], want)
        changed += int((want != yuv[f * per * 384:(f + 1) * per * 384]).sum())
    if W * H > 1:
        assert changed > 0      # the filter did something


def test_blockiness_goes_down():
    """Sanity of the filter as a filter: on reconstructed random-residual pictures the mean step across macroblock edges
    shrinks, the picture inside 4x4 blocks (columns 1, 2 of every block) is untouched by vertical-edge filtering of bS 3
    beyond p1/q1, and a second application changes less than the first."""
    fp, mbs, yuv = _picture(12, 8, 1, 5, i4x4=0.6, i8x8=0.0, qp=(30, 44))
    st, out = oracle.deblock(fp, abi.make_deblock_params(), 1, mbs, yuv)
    Y0 = yuv[:12 * 8 * 256].reshape(128, 192).astype(np.int64)
    Y1 = out[:12 * 8 * 256].reshape(128, 192).astype(np.int64)
    edge0 = np.abs(Y0[:, 16::16] - Y0[:, 15:-1:16]).mean()
    edge1 = np.abs(Y1[:, 16::16] - Y1[:, 15:-1:16]).mean()
    assert edge1 < edge0
    st, out2 = oracle.deblock(fp, abi.make_deblock_params(), 1, mbs, out)
    assert (out2 != out).sum() < (out != yuv).sum()


# ---- the HIP kernel's source under the CPU lane emulator (tests/emu) ---------------------------------------------------------
@pytest.mark.parametrize("geo", [(1, 1, 1, False, 1, 0, 1), (2, 1, 2, False, 1, 0, 1), (1, 2, 1, True, 1, 0, 1), (5, 4, 2, False, 2, 1, -1),
                                 (6, 5, 2, True, 3, 0, 1), (5, 9, 2, True, 4, 3, -1), (7, 13, 1, False, 5, 2, 1), (11, 7, 1, True, 2, 0, 1)])
def test_emulated_deblock_kernel_matches_oracle(geo):
    """dryv_amd/csrc/deblock_kernel.h compiled for the lane emulator: one-macroblock pictures, single rows and columns,
    several bands per picture with several waves claiming them concurrently in either order (the side-buffer hand-off
    between bands runs for real), with and without 8x8-transform macroblocks, non-zero filter and chroma-qp offsets."""
    import os
    import sys
    sys.path.insert(0, os.path.join(os.path.dirname(__file__), "emu"))
    import emu
    W, H, frames, t8, waves, first, order = geo
    fp, mbs, yuv = _picture(W, H, frames, 31 + W, t8=t8, i4x4=0.4, i8x8=0.3 if t8 else 0.0, qp=(8, 51), cqo_cb=2, cqo_cr=-3)
    for dp in (abi.make_deblock_params(0, 1, -2), abi.make_deblock_params(2, -6, 6), abi.make_deblock_params(1, 0, 0)):
        st, want = oracle.deblock(fp, dp, frames, mbs, yuv)
        st2, got = emu.deblock(fp, dp, frames, mbs, yuv, n_waves=waves, first=first, order=order)
        assert st == 0 and st2 == 0 and np.array_equal(got, want)


# ---- the product path ---------------------------------------------------------------------------------------------------------
@pytest.mark.gpu
@pytest.mark.parametrize("geo", [(1, 1, 2, False), (7, 5, 3, True), (20, 15, 2, True), (13, 21, 2, False), (120, 68, 2, False),
                                 (240, 135, 1, True)])
def test_deblock_on_gpu(recon_ctx, geo):
    """dryv_recon_deblock_device on reconstructed pictures in device memory, against the oracle; after a reconstruction on
    the same context (submit_device -> sync -> deblock_device -> sync), as a decoder would chain them."""
    import torch
    W, H, frames, t8 = geo
    fp = abi.make_frame_params(W, H, transform_8x8=t8, cqo_cb=-2, cqo_cr=3)
    mbs, co = synth.generate(fp, synth.config(i4x4=0.4, i8x8=0.3 if t8 else 0.0, qp=(12, 51)), 55 + W, 0, frames)
    st, yuv = oracle.reconstruct(fp, frames, mbs, co)
    d_m = torch.from_numpy(mbs.view(np.uint8).reshape(-1)).cuda()
    d_c = torch.from_numpy(co).cuda()
    for dp in (abi.make_deblock_params(0, 0, 0), abi.make_deblock_params(0, 3, -4), abi.make_deblock_params(1, 0, 0)):
        st, want = oracle.deblock(fp, dp, frames, mbs, yuv)
        d_y = torch.zeros(yuv.size, dtype=torch.uint8, device="cuda")
        recon_ctx.submit_device(fp, frames, d_m.data_ptr(), d_c.data_ptr(), d_y.data_ptr())
        recon_ctx.sync()
        recon_ctx.deblock_device(fp, dp, frames, d_m.data_ptr(), d_y.data_ptr())
        recon_ctx.sync()
        got = d_y.cpu().numpy()
        assert np.array_equal(got, want), int(np.flatnonzero(got != want)[0])
    with pytest.raises(Exception):
        recon_ctx.deblock_device(fp, abi.make_deblock_params(0, 9, 0), frames, d_m.data_ptr(), d_y.data_ptr())


@pytest.mark.gpu
def test_host_path_with_deblocking_and_packing(recon_ctx):
    """dryv_recon_submit + dryv_recon_wait_filtered: reconstruction, deblocking and the output stage chained on the device,
    only the wanted bytes copied back: equal to oracle reconstruction -> oracle deblocking -> cropping in numpy."""
    from test_recon_gpu import _expected_packed
    W, H, frames = 12, 9, 2
    fp = abi.make_frame_params(W, H)
    mbs, co = synth.generate(fp, synth.config(i4x4=0.6, i8x8=0.0, qp=(20, 48)), 808, 0, frames)
    st, yuv = oracle.reconstruct(fp, frames, mbs, co)
    dp = abi.make_deblock_params(0, 2, 1)
    st, filt = oracle.deblock(fp, dp, frames, mbs, yuv)
    for od, crop in ((None, (0, 0, 0, 0)), (abi.make_output_desc(abi.OUT_NV12, (2, 0, 0, 8)), (2, 0, 0, 8))):
        recon_ctx.submit(fp, frames, mbs, co)
        got = recon_ctx.wait_filtered(dp, od)
        want = filt if od is None else _expected_packed(filt, W, H, frames, abi.OUT_NV12, crop)
        assert np.array_equal(got, want)
    recon_ctx.submit(fp, frames, mbs, co)
    assert np.array_equal(recon_ctx.wait_filtered(None, None), yuv)     # no stage: the plain pictures


@pytest.mark.gpu
def test_real_picture_reconstructed_and_deblocked(tmp_path):
    """The first picture of the real .mp4 with its slice header's own deblocking parameters, through `frame_harness
    decode-deblocked` (host parse -> GPU reconstruction -> GPU deblocking): equal to the oracles' result, different from the
    undeblocked picture, and -- a natural image coded at qp 31 -- less blocky across macroblock edges than it."""
    import os
    import subprocess
    from dryv_amd import _build, h264
    fixture = os.path.join(os.path.dirname(__file__), "golden", "realshort.mp4")
    exe = _build.build_harness()
    out = tmp_path / "deblocked.yuv"
    r = subprocess.run([exe, "decode-deblocked", fixture, str(out)], capture_output=True, text=True, timeout=120)
    assert r.returncode == 0 and "disable_idc 0" in r.stdout, (r.stdout, r.stderr)
    fp, mbs, co, info = h264.parse_first_islice(open(fixture, "rb").read())
    st, yuv = oracle.reconstruct(fp, 1, mbs, co)
    st, want = oracle.deblock(fp, info["deblock"], 1, mbs, yuv)
    got = np.fromfile(str(out), dtype=np.uint8)
    assert np.array_equal(got, want) and not np.array_equal(got, yuv)
    Y0 = yuv[:320 * 240].reshape(240, 320).astype(np.int64)
    Y1 = got[:320 * 240].reshape(240, 320).astype(np.int64)
    step0 = np.abs(Y0[:, 16::16] - Y0[:, 15:-1:16]).mean() + np.abs(Y0[16::16] - Y0[15:-1:16]).mean()
    step1 = np.abs(Y1[:, 16::16] - Y1[:, 15:-1:16]).mean() + np.abs(Y1[16::16] - Y1[15:-1:16]).mean()
    assert step1 < step0
