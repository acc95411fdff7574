"""The oracle (a line-by-line restatement of the reference's Rust) against oracle/spec_model.py (written from the
standard's clauses, with the reference's quirks Q1-Q5 patched in explicitly): they must agree bit for bit on random
small pictures covering every macroblock kind, legal and illegal modes, every QP, non-flat lists and chroma offsets.
CPU only. Neither is pinned to the reference ("parity unpinned"); this catches transcription errors in either."""
import numpy as np
import pytest

import oracle
from oracle import spec_model
from dryv_amd import abi, synth
from util import first_mismatch

CASES = [
    ("i4x4", 5, 4, 2, dict(i4x4=1.0, i8x8=0.0), {}),
    ("i16", 5, 4, 2, dict(i4x4=0.0, i8x8=0.0), {}),
    ("i8x8", 5, 4, 2, dict(i4x4=0.0, i8x8=1.0), dict(transform_8x8=True)),
    ("mix_all_qp", 6, 5, 2, dict(i4x4=0.4, i8x8=0.3, qp=(0, 51)), dict(transform_8x8=True, cqo_cb=-5, cqo_cr=9)),
    ("illegal_modes", 6, 5, 2, dict(i4x4=0.4, i8x8=0.3, legal_modes_only=False), dict(transform_8x8=True)),
    ("dark_q2", 6, 5, 2, dict(i4x4=0.3, i8x8=0.2, coded=1.0, p0=0.6, max_level=300, qp=(30, 51)), dict(transform_8x8=True)),
    ("single_col", 1, 6, 1, dict(i4x4=0.4, i8x8=0.3), dict(transform_8x8=True)),
    ("single_row", 7, 1, 1, dict(i4x4=0.4, i8x8=0.3), dict(transform_8x8=True)),
]


@pytest.mark.parametrize("case", CASES, ids=[c[0] for c in CASES])
def test_oracle_matches_spec_model(case):
    name, W, H, frames, skw, fkw = case
    fp = abi.make_frame_params(W, H, **fkw)
    mbs, co = synth.generate(fp, synth.config(**skw), 700 + CASES.index(case), 0, frames)
    st, want = oracle.reconstruct(fp, frames, mbs, co)
    st2, got = spec_model.reconstruct(fp, frames, mbs, co)
    assert st == st2 == 0
    assert np.array_equal(got, want), first_mismatch(got, want, W, H)


def test_nonflat_lists_and_unsupported_record():
    rng = np.random.default_rng(11)
    fp = abi.make_frame_params(5, 4, transform_8x8=True, scaling4x4=rng.integers(4, 48, size=(6, 16)),
                               scaling8x8=rng.integers(4, 48, size=(6, 64)))
    mbs, co = synth.generate(fp, synth.config(i4x4=0.4, i8x8=0.3, max_level=200), 77, 0, 1)
    mbs[7]["mb_kind"] = 25  # I_PCM: outside the domain; stays zero, counts as available Intra16x16 for its neighbours
    st, want = oracle.reconstruct(fp, 1, mbs, co)
    st2, got = spec_model.reconstruct(fp, 1, mbs, co)
    assert st == st2 == abi.DRYV_E_UNSUPPORTED
    assert np.array_equal(got, want), first_mismatch(got, want, 5, 4)


def test_quirks_matter():
    """With the patches switched off the model is the standard, and the reference's output differs from it
    on inputs that reach Q1/Q2/Q4 (so the patches above are load-bearing, not decoration)."""
    name, W, H, frames, skw, fkw = CASES[4]
    fp = abi.make_frame_params(W, H, **fkw)
    mbs, co = synth.generate(fp, synth.config(**dict(skw, legal_modes_only=True, i4x4=0.0, i8x8=1.0)), 5, 0, 1)
    _, want = oracle.reconstruct(fp, 1, mbs, co)
    spec_model.QUIRKS = False
    try:
        _, got = spec_model.reconstruct(fp, 1, mbs, co)
    finally:
        spec_model.QUIRKS = True
    assert not np.array_equal(got, want)  # Q1 hits the column-0 Intra8x8 blocks of every row but the first
