"""Synthetic all-intra batches at the FFI boundary, via libdryv_synth.so.

Standard workloads (BASELINE.json `configs`):
  C2  1080p (120x68 MBs), 4x4 transform only: 70 % Intra4x4 / 30 % Intra16x16
  C3  4K (240x135 MBs), 8x8 enabled: 40 % Intra8x8 / 35 % Intra4x4 / 25 % Intra16x16
All: 4:2:0, 8-bit, flat-16 scaling lists, chroma qp offsets 0, qp uniform 20..40, blocks coded
w.p. 0.6, P(nonzero at scan k) = 0.5*0.8^k (8x8: 0.5*0.93^k), |level| = 1+Geom(0.5) <= 2047.
PRNG: splitmix64 seeded 0x64727976_00000000 ^ (config_id << 24) ^ frame_index.
"""
import ctypes as C

import numpy as np

from . import _build, abi


class SynthConfig(C.Structure):
    _fields_ = [(n, C.c_uint32) for n in (
        "permille_i4x4", "permille_i8x8", "qp_min", "qp_max", "permille_coded", "p0_q16", "decay4_q16",
        "decay8_q16", "max_level", "legal_modes_only", "permille_prev_flag")]


def _q16(x):
    return int(round(x * 65536))


def config(i4x4=0.7, i8x8=0.0, qp=(20, 40), coded=0.6, p0=0.5, decay4=0.8, decay8=0.93, max_level=2047,
           legal_modes_only=True, prev_flag=0.5):
    c = SynthConfig()
    c.permille_i4x4 = int(round(i4x4 * 1000))
    c.permille_i8x8 = int(round(i8x8 * 1000))
    c.qp_min, c.qp_max = qp
    c.permille_coded = int(round(coded * 1000))
    c.p0_q16 = _q16(p0)
    c.decay4_q16 = _q16(decay4)
    c.decay8_q16 = _q16(decay8)
    c.max_level = max_level
    c.legal_modes_only = int(legal_modes_only)
    c.permille_prev_flag = int(round(prev_flag * 1000))
    return c


# name -> (config_id, width_mbs, height_mbs, frames, transform_8x8, SynthConfig kwargs)
WORKLOADS = {
    "C2_1080p_intra_4x4": (2, 120, 68, 300, False, dict(i4x4=0.7, i8x8=0.0)),
    "C3_4k_intra_8x8": (3, 240, 135, 100, True, dict(i4x4=0.35, i8x8=0.40)),
}

_lib = None


def _load():
    global _lib
    if _lib is None:
        _lib = C.CDLL(_build.build_synth())
        _lib.dryv_synth_generate.restype = C.c_int
        _lib.dryv_synth_generate.argtypes = [C.POINTER(abi.FrameParams), C.POINTER(SynthConfig), C.c_uint64,
                                             C.c_uint32, C.c_uint32, C.c_void_p, C.c_void_p]
    return _lib


def generate(fp, cfg, config_id, first_frame, n_frames, out_mbs=None, out_coeffs=None):
    """Returns (mbs, coeffs): MB_DESC_DTYPE[n_mbs], int16[n_mbs, 384]."""
    n_mbs = n_frames * fp.pic_width_in_mbs * fp.pic_height_in_mbs
    mbs = np.empty(n_mbs, dtype=abi.MB_DESC_DTYPE) if out_mbs is None else out_mbs
    coeffs = np.empty((n_mbs, abi.COEFFS_PER_MB), dtype=np.int16) if out_coeffs is None else out_coeffs
    st = _load().dryv_synth_generate(C.byref(fp), C.byref(cfg), config_id, first_frame, n_frames,
                                     mbs.ctypes.data, coeffs.ctypes.data)
    if st != 0:
        raise RuntimeError("dryv_synth_generate failed: %d" % st)
    return mbs, coeffs


def workload(name, n_frames=None, first_frame=0):
    """(fp, mbs, coeffs, n_frames) for a named BASELINE.json workload."""
    cid, w, h, frames, t8, kw = WORKLOADS[name]
    fp = abi.make_frame_params(w, h, transform_8x8=t8)
    n = frames if n_frames is None else n_frames
    mbs, coeffs = generate(fp, config(**kw), cid, first_frame, n)
    return fp, mbs, coeffs, n
