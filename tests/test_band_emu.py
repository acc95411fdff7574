"""The band kernel's source (dryv_amd/csrc/band_kernel.h), compiled for the CPU lane emulator of tests/emu, against
the oracle. Runs without a GPU: it checks the kernel's index / schedule / arithmetic logic (every lane's program,
cross-lane operations, LDS layout, the FRONT/BACK record protocol, band hand-off order with several teams running
concurrently); the GPU parity tests (`-m gpu`) remain the proof for the machine code. Bit-exact bar. Reference parity
itself stays unpinned (see oracle/dryv_oracle.c header).

Every case runs twice: as a stream without the 8x8 transform (the kernel's HAS_I8 = false build, Intra8x8 share moved to
the other kinds) and, where the case has one, with transform_8x8_mode_flag = 1 (HAS_I8 = true: Intra8x8 residuals, mode
derivation across Intra4x4 / Intra8x8 neighbours, the 8x8 block chain, quirk Q1)."""
import os
import sys

import numpy as np
import pytest

import oracle
from dryv_amd import abi, synth
from util import first_mismatch, packed16_bound_batches, packed16_8x8_bound_batches

sys.path.insert(0, os.path.join(os.path.dirname(__file__), "emu"))
import emu  # noqa: E402


def check(fp, frames, mbs, co, expect_status=0, teams=1, first=0, order=1, defs=()):
    st, want = oracle.reconstruct(fp, frames, mbs, co)
    st2, got = emu.reconstruct(fp, frames, mbs, co, teams, first, order, defs=defs)
    assert (st != 0) == (expect_status != 0)
    assert (st2 != 0) == (expect_status != 0)
    W, H = fp.pic_width_in_mbs, fp.pic_height_in_mbs
    assert np.array_equal(got, want), first_mismatch(got, want, W, H)


CASES = [
    ("i16_only", 7, 5, 2, dict(i4x4=0.0, i8x8=0.0), {}),
    ("i4x4_only", 7, 5, 2, dict(i4x4=1.0, i8x8=0.0), {}),
    ("i8x8_only", 7, 5, 2, dict(i4x4=0.0, i8x8=1.0), dict(transform_8x8=True)),
    ("c2_mix_small", 12, 9, 2, dict(i4x4=0.7, i8x8=0.0), {}),
    ("c3_mix_small", 12, 9, 2, dict(i4x4=0.35, i8x8=0.40), dict(transform_8x8=True)),
    ("single_mb", 1, 1, 3, dict(i4x4=0.5, i8x8=0.3), dict(transform_8x8=True)),
    ("single_row", 9, 1, 2, dict(i4x4=0.5, i8x8=0.3), dict(transform_8x8=True)),
    ("single_col", 1, 9, 2, dict(i4x4=0.5, i8x8=0.3), dict(transform_8x8=True)),
    ("wider_than_a_wave", 70, 5, 1, dict(i4x4=0.6, i8x8=0.3), dict(transform_8x8=True)),   # two batches of the mode pre-pass
    ("two_cols", 2, 17, 1, dict(i4x4=0.5, i8x8=0.3), dict(transform_8x8=True)),
    ("all_qp", 10, 8, 2, dict(i4x4=0.5, i8x8=0.3, qp=(0, 51)), dict(transform_8x8=True)),
    ("dense_big_levels", 8, 6, 2, dict(i4x4=0.5, i8x8=0.3, coded=1.0, p0=0.9, decay4=0.97, decay8=0.99, max_level=2047,
                                       qp=(0, 51)), dict(transform_8x8=True)),
    ("illegal_modes_q4", 9, 7, 2, dict(i4x4=0.5, i8x8=0.3, legal_modes_only=False), dict(transform_8x8=True)),
    ("chroma_qp_offsets", 9, 7, 2, dict(i4x4=0.5, i8x8=0.3, qp=(0, 51)), dict(transform_8x8=True, cqo_cb=-7, cqo_cr=11)),
    ("zero_residual", 9, 7, 1, dict(i4x4=0.5, i8x8=0.3, coded=0.0), dict(transform_8x8=True)),
    ("dark_q2_zeros", 9, 7, 2, dict(i4x4=0.4, i8x8=0.2, coded=1.0, p0=0.6, max_level=300, qp=(30, 51)),
     dict(transform_8x8=True)),
]


@pytest.mark.parametrize("with8", [False, True], ids=["no8x8", "8x8"])
@pytest.mark.parametrize("case", CASES, ids=[c[0] for c in CASES])
def test_emulated_band_kernel_matches_oracle(case, with8):
    name, W, H, frames, skw, fkw = case
    if with8 and not fkw.get("transform_8x8"):
        pytest.skip("case has no 8x8 variant")
    if not with8:
        skw, fkw = dict(skw, i8x8=0.0), dict(fkw, transform_8x8=False)
    fp = abi.make_frame_params(W, H, **fkw)
    mbs, co = synth.generate(fp, synth.config(**skw), 100 + CASES.index(case), 0, frames)
    check(fp, frames, mbs, co)


@pytest.mark.parametrize("geo", [(7, 5, 3, 6, 0, 1), (7, 5, 3, 6, 5, -1), (7, 5, 1, 2, 1, 1), (5, 13, 2, 8, 7, -1),
                                 (5, 13, 2, 3, 0, 1), (9, 9, 2, 5, 9, -1)])
def test_emulated_teams_run_concurrently(geo):
    """Several teams claim bands at once and are scheduled round-robin in either order (a wave runs until it polls in
    vain): bands below wait for bands above, BACK waits for FRONT's record and FRONT for BACK's buffer. A deadlock in
    any of these protocols stops the emulator with "nothing makes progress"."""
    W, H, frames, teams, first, order = geo
    for t8, cfg in ((False, dict(i4x4=0.6, i8x8=0.0)), (True, dict(i4x4=0.35, i8x8=0.4))):
        fp = abi.make_frame_params(W, H, transform_8x8=t8)
        mbs, co = synth.generate(fp, synth.config(**cfg), 100, 0, frames)
        check(fp, frames, mbs, co, teams=teams, first=first, order=order)


@pytest.mark.parametrize("nsy,nsc", [(2, 2), (8, 4), (4, 4), (2, 8)])
def test_emulated_staging_widths(nsy, nsc):
    """The output staging width (macroblocks per flushed row segment) is a build-time choice; every choice must give
    the same pictures, ragged right edges included (widths 11 and 17 are multiples of none of them)."""
    defs = ("-DDRYV_BAND_NSY=%d" % nsy, "-DDRYV_BAND_NSC=%d" % nsc)
    for W, H, frames, teams in ((11, 6, 2, 3), (17, 5, 1, 2), (3, 4, 2, 1)):
        fp = abi.make_frame_params(W, H)
        mbs, co = synth.generate(fp, synth.config(i4x4=0.6, i8x8=0.0), 321 + W, 0, frames)
        check(fp, frames, mbs, co, teams=teams, defs=defs)


def test_emulated_fuzz():
    rng = np.random.default_rng(77)
    for k in range(16):
        W, H, frames = int(rng.integers(1, 14)), int(rng.integers(1, 11)), int(rng.integers(1, 3))
        t8 = bool(k & 1)
        i4 = float(rng.choice([0.0, 0.3, 0.7, 1.0]))
        i8 = float(rng.choice([0.0, 0.3, 0.6])) if t8 else 0.0
        if i4 + i8 > 1.0:
            i4 = 1.0 - i8
        lo = int(rng.integers(0, 40))
        flat = rng.random() < 0.5
        fkw = dict(cqo_cb=int(rng.integers(-12, 13)), cqo_cr=int(rng.integers(-12, 13)), transform_8x8=t8)
        if not flat:
            fkw.update(scaling4x4=rng.integers(4, 48, size=(6, 16)))
            if t8:
                fkw.update(scaling8x8=rng.integers(4, 48, size=(6, 64)))
        skw = dict(i4x4=i4, i8x8=i8, qp=(lo, int(rng.integers(lo, 52))), coded=float(rng.choice([0.2, 0.6, 1.0])),
                   max_level=int(rng.choice([15, 300, 2047])) if flat else 200,
                   legal_modes_only=bool(rng.random() < 0.7), prev_flag=float(rng.choice([0.1, 0.5, 0.9])))
        fp = abi.make_frame_params(W, H, **fkw)
        mbs, co = synth.generate(fp, synth.config(**skw), 2000 + k, k, frames)
        check(fp, frames, mbs, co, teams=int(rng.integers(1, 4)))


@pytest.mark.parametrize("qp_range", [(0, 24), (25, 40), (41, 51)])
def test_emulated_full_int16_range_every_qp(qp_range):
    """The FFI carries int16 coefficients and scaling weights up to 255; the reference computes in 64-bit isize.
    Blocks whose coefficients exceed the per-qp int32-exactness bound take the kernel's 64-bit pass."""
    rng = np.random.default_rng(9 + qp_range[0])
    s4, s8 = rng.integers(1, 256, size=(6, 16)), rng.integers(1, 256, size=(6, 64))
    for lists, cfg in ((dict(), dict(i4x4=0.6, i8x8=0.0)), (dict(scaling4x4=s4), dict(i4x4=0.6, i8x8=0.0)),
                       (dict(transform_8x8=True), dict(i4x4=0.3, i8x8=0.5)),
                       (dict(transform_8x8=True, scaling4x4=s4, scaling8x8=s8), dict(i4x4=0.3, i8x8=0.5))):
        fp = abi.make_frame_params(6, 5, **lists)
        mbs, co = synth.generate(fp, synth.config(coded=1.0, p0=0.9, decay4=0.97, decay8=0.99, qp=qp_range, **cfg), 61, 0, 2)
        scale = rng.choice([1, 40, 700, 6000], size=(co.shape[0], 1))
        co = np.clip(co.astype(np.int64) * scale, -32768, 32767).astype(np.int16)
        check(fp, 2, mbs, co)


def test_emulated_fast_path_at_its_exactness_bound():
    """Adversarial input for the int32 fast path: every coefficient of every block sits exactly AT the per-qp bound under
    which the kernel stays in 32-bit arithmetic (KParams thr4 / thr8, recomputed here), with the sign patterns that
    maximise the butterflies' growth (all equal; the first basis functions' signs). One above the bound takes the 64-bit
    pass. Both must equal the oracle's 64-bit result bit for bit."""
    V4 = np.array([[10, 16, 13], [11, 18, 14], [13, 20, 16], [14, 23, 18], [16, 25, 20], [18, 29, 23]])
    V8 = np.array([[20, 18, 32, 19, 25, 24], [22, 19, 35, 21, 28, 26], [26, 23, 42, 24, 33, 31], [28, 25, 45, 26, 35, 33],
                   [32, 28, 51, 30, 40, 38], [36, 32, 58, 34, 46, 43]])
    rng = np.random.default_rng(4)
    for t8 in (False, True):
        fp = abi.make_frame_params(4, 4, transform_8x8=t8)
        for qp in (0, 11, 23, 24, 35, 36, 41, 47, 51):
            qd, qm = qp // 6, qp % 6
            thr4 = min((1 << 26) // ((16 * V4[qm].max()) << max(qd - 4, 0)), 32767)
            thr8 = min((1 << 23) // ((16 * V8[qm].max()) << max(qd - 6, 0)), 32767)
            for bump in (0, 1):
                cfg = synth.config(i4x4=0.4, i8x8=0.4 if t8 else 0.0, coded=1.0, qp=(qp, qp))
                mbs, co = synth.generate(fp, cfg, 900 + qp, 0, 1)
                co = co.astype(np.int64)
                for a in range(co.shape[0]):
                    lim = (thr8 if mbs["mb_kind"][a] == 1 else thr4) + bump
                    lim = min(lim, 32767)
                    pat = rng.integers(0, 3)
                    sign = np.ones(384, dtype=np.int64) if pat == 0 else (np.where(np.arange(384) % 2, -1, 1) if pat == 1
                                                                        else rng.choice([-1, 1], size=384))
                    co[a] = sign * lim
                check(fp, 1, mbs, co.astype(np.int16))


def test_emulated_packed16_8x8_path_at_its_bound():
    """The 8x8 residuals run on packed 16-bit pairs when every Intra8x8 block of a step has sum |c| within the per-qp bound
    (T_THR8P); one unit more sends the step through the 32-bit passes. Both must equal the oracle bit for bit."""
    fp = abi.make_frame_params(5, 4, transform_8x8=True)
    for qp, bump, mbs, co in packed16_8x8_bound_batches(fp, synth):
        check(fp, 1, mbs, co)


def test_emulated_packed16_path_at_its_bound():
    """4x4 residuals run on packed 16-bit pairs when every block of a step has sum |c| * max LS' (+ |dc|) <= 32700; a block
    above it sends the step through the 32-bit path. Blocks exactly at the bound and one above it, every qp class."""
    fp = abi.make_frame_params(5, 4)
    for qp, bump, mbs, co in packed16_bound_batches(fp, synth):
        check(fp, 1, mbs, co)
