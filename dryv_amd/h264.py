"""ctypes view of libdryv_h264.so — the host producer: mp4 / Annex-B demux + SPS/PPS/slice header +
I-slice CABAC parse into the batch the reconstruction ABI takes, and the CABAC I-slice encoder that writes synthetic
batches as real Annex-B streams. Host code only; nothing here touches a GPU."""
import ctypes as C
import os

import numpy as np

from . import _build, abi

SO = os.path.join(_build.LIB, "libdryv_h264.so")
_lib = None


def build(force=False):
    host = os.path.join(_build.HERE, "host")
    srcs = [os.path.join(host, f) for f in ("h264_capi.cpp", "h264_islice.hpp", "cabac_tables.inc")]
    os.makedirs(_build.LIB, exist_ok=True)
    if force or _build._stale(SO, srcs + [os.path.join(_build.HERE, "..", "include", "dryv_recon.h")]):
        _build._run(["g++", "-O2", "-std=c++17", "-Wall", "-fPIC", "-shared", "-pthread", "-o", SO, srcs[0]])
    return SO


def _load():
    global _lib
    if _lib is None:
        _lib = C.CDLL(build())
        _lib.dryv_h264_parse.restype = C.c_void_p
        _lib.dryv_h264_parse.argtypes = [C.c_void_p, C.c_size_t]
        _lib.dryv_h264_last_error.restype = C.c_char_p
        _lib.dryv_h264_free.argtypes = [C.c_void_p]
        for n in ("dryv_h264_params", "dryv_h264_mbs", "dryv_h264_coeffs"):
            getattr(_lib, n).restype = C.c_void_p
            getattr(_lib, n).argtypes = [C.c_void_p]
        _lib.dryv_h264_info.argtypes = [C.c_void_p, C.c_void_p]
        _lib.dryv_h264_encode_idr.restype = C.c_longlong
        _lib.dryv_h264_encode_idr.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_void_p, C.c_size_t]
        _lib.dryv_h264_encode_idr_cropped.restype = C.c_longlong
        _lib.dryv_h264_encode_idr_cropped.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_void_p, C.c_void_p,
                                                      C.c_size_t]
        _lib.dryv_h264_crop.argtypes = [C.c_void_p, C.c_void_p]
        _lib.dryv_h264_deblock_params.argtypes = [C.c_void_p, C.c_void_p]
        _lib.dryv_h264_parse_all.restype = C.c_void_p
        _lib.dryv_h264_parse_all.argtypes = [C.c_void_p, C.c_size_t, C.c_size_t]
        _lib.dryv_h264_stream_params.restype = C.c_longlong
        _lib.dryv_h264_stream_params.argtypes = [C.c_void_p, C.c_size_t, C.c_void_p]
        _lib.dryv_h264_parse_all_into.restype = C.c_longlong
        _lib.dryv_h264_parse_all_into.argtypes = [C.c_void_p, C.c_size_t, C.c_size_t, C.c_uint, C.c_void_p, C.c_void_p, C.c_size_t,
                                                  C.c_void_p, C.c_void_p]
        _lib.dryv_h264_parse_all_mt.restype = C.c_void_p
        _lib.dryv_h264_parse_all_mt.argtypes = [C.c_void_p, C.c_size_t, C.c_size_t, C.c_uint]
        _lib.dryv_h264_batch_free.argtypes = [C.c_void_p]
        for n, rt in (("pictures", C.c_size_t), ("skipped", C.c_size_t), ("params", C.c_void_p), ("mbs", C.c_void_p),
                      ("coeffs", C.c_void_p), ("tails_ok", C.c_int)):
            getattr(_lib, "dryv_h264_batch_" + n).restype = rt
            getattr(_lib, "dryv_h264_batch_" + n).argtypes = [C.c_void_p]
        _lib.dryv_h264_batch_crop.argtypes = [C.c_void_p, C.c_void_p]
        _lib.dryv_h264_bin_log.restype = C.c_longlong
        _lib.dryv_h264_bin_log.argtypes = [C.c_void_p, C.c_size_t, C.c_size_t, C.c_void_p, C.c_size_t, C.c_void_p]
        _lib.dryv_h264_encode_stream.restype = C.c_longlong
        _lib.dryv_h264_encode_stream.argtypes = [C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_int, C.c_void_p, C.c_void_p,
                                                 C.c_size_t]
    return _lib


class H264Error(RuntimeError):
    pass


def parse_first_islice(data):
    """data: bytes of an .mp4 (ISO-BMFF, avc1) or Annex-B stream. Returns (fp, mbs, coeffs, info)."""
    lib = _load()
    buf = np.frombuffer(data, dtype=np.uint8)
    h = lib.dryv_h264_parse(buf.ctypes.data, buf.size)
    if not h:
        raise H264Error(lib.dryv_h264_last_error().decode())
    try:
        fp = abi.FrameParams()
        C.memmove(C.addressof(fp), lib.dryv_h264_params(h), C.sizeof(fp))
        n = fp.pic_width_in_mbs * fp.pic_height_in_mbs
        mbs = np.empty(n, dtype=abi.MB_DESC_DTYPE)
        C.memmove(mbs.ctypes.data, lib.dryv_h264_mbs(h), n * 16)
        coeffs = np.empty((n, 384), dtype=np.int16)
        C.memmove(coeffs.ctypes.data, lib.dryv_h264_coeffs(h), n * 768)
        info = np.zeros(8, dtype=np.int64)
        lib.dryv_h264_info(h, info.ctypes.data)
        crop = np.zeros(4, dtype=np.int32)
        lib.dryv_h264_crop(h, crop.ctypes.data)
        dbp = abi.DeblockParams()
        lib.dryv_h264_deblock_params(h, C.addressof(dbp))
    finally:
        lib.dryv_h264_free(h)
    keys = ("bins", "slice_bytes", "bits_unread", "tail_ok", "n_i4x4", "n_i8x8", "n_i16x16", "slice_qp")
    d = dict(zip(keys, (int(v) for v in info)))
    d["crop"] = tuple(int(v) for v in crop)   # luma samples: left, right, top, bottom (sps.rs:252-267)
    d["deblock"] = dbp                        # the slice header's deblocking syntax elements (header.rs:609-640)
    return fp, mbs, coeffs, d


def encode_idr(fp, mbs, coeffs, slice_qp=26, crop=None):
    """One picture (flat scaling lists) -> Annex-B bytes (SPS, PPS, IDR I slice, CABAC). crop: optional frame cropping
    rectangle (left, right, top, bottom) in luma samples, written to the SPS."""
    lib = _load()
    mbs = np.ascontiguousarray(mbs)
    coeffs = np.ascontiguousarray(coeffs, dtype=np.int16)
    n = fp.pic_width_in_mbs * fp.pic_height_in_mbs
    assert mbs.size == n and coeffs.size == n * 384
    cap = 64 + n * 1200
    out = np.empty(cap, dtype=np.uint8)
    cr = np.asarray(crop if crop is not None else (0, 0, 0, 0), dtype=np.int32)

    def call(buf):
        return lib.dryv_h264_encode_idr_cropped(C.addressof(fp), mbs.ctypes.data, coeffs.ctypes.data, slice_qp,
                                                cr.ctypes.data, buf.ctypes.data, buf.size)
    r = call(out)
    if r < 0:
        out = np.empty(-r, dtype=np.uint8)
        r = call(out)
    if r <= 0:
        raise H264Error(lib.dryv_h264_last_error().decode())
    return out[:r].tobytes()


def parse_all_islices(data, max_pictures=0, threads=1):
    """Every picture of an .mp4 / Annex-B stream that is a single I slice, as one batch for the reconstruction ABI.
    Returns (fp, n_pictures, mbs, coeffs, info); inter pictures in between are skipped (info["skipped"]). threads:
    pictures are parsed in parallel (0 = all hardware threads)."""
    lib = _load()
    buf = np.frombuffer(data, dtype=np.uint8)
    h = lib.dryv_h264_parse_all_mt(buf.ctypes.data, buf.size, int(max_pictures), int(threads))
    if not h:
        raise H264Error(lib.dryv_h264_last_error().decode())
    try:
        n_pic = lib.dryv_h264_batch_pictures(h)
        fp = abi.FrameParams()
        C.memmove(C.addressof(fp), lib.dryv_h264_batch_params(h), C.sizeof(fp))
        n = n_pic * fp.pic_width_in_mbs * fp.pic_height_in_mbs
        mbs = np.empty(n, dtype=abi.MB_DESC_DTYPE)
        C.memmove(mbs.ctypes.data, lib.dryv_h264_batch_mbs(h), n * 16)
        coeffs = np.empty((n, 384), dtype=np.int16)
        C.memmove(coeffs.ctypes.data, lib.dryv_h264_batch_coeffs(h), n * 768)
        crop = np.zeros(4, dtype=np.int32)
        lib.dryv_h264_batch_crop(h, crop.ctypes.data)
        info = {"skipped": int(lib.dryv_h264_batch_skipped(h)), "tails_ok": int(lib.dryv_h264_batch_tails_ok(h)),
                "crop": tuple(int(v) for v in crop)}
    finally:
        lib.dryv_h264_batch_free(h)
    return fp, int(n_pic), mbs, coeffs, info


def encode_stream(fp, n_pictures, mbs, coeffs, slice_qp=26, crop=None):
    """n_pictures pictures (flat scaling lists) -> one all-intra Annex-B stream: SPS, PPS, one IDR slice per picture."""
    lib = _load()
    mbs = np.ascontiguousarray(mbs)
    coeffs = np.ascontiguousarray(coeffs, dtype=np.int16)
    n = n_pictures * fp.pic_width_in_mbs * fp.pic_height_in_mbs
    assert mbs.size == n and coeffs.size == n * 384
    cr = np.asarray(crop if crop is not None else (0, 0, 0, 0), dtype=np.int32)
    out = np.empty(64 + n * 1200, dtype=np.uint8)

    def call(buf):
        return lib.dryv_h264_encode_stream(C.addressof(fp), int(n_pictures), mbs.ctypes.data, coeffs.ctypes.data, slice_qp,
                                           cr.ctypes.data, buf.ctypes.data, buf.size)
    r = call(out)
    if r < 0:
        out = np.empty(-r, dtype=np.uint8)
        r = call(out)
    if r <= 0:
        raise H264Error(lib.dryv_h264_last_error().decode())
    return out[:r].tobytes()


def parse_all_islices_into(data, mbs_out, coeffs_out, max_pictures=0, threads=0):
    """parse_all_islices straight into the caller's batch buffers (numpy arrays of abi.MB_DESC_DTYPE / int16, e.g. page-locked
    ones from ReconContext.alloc_host): no intermediate copies. Returns (fp, n_pictures, info)."""
    lib = _load()
    buf = np.frombuffer(data, dtype=np.uint8)
    fp = abi.FrameParams()
    info = np.zeros(4, dtype=np.int64)
    capacity = min(mbs_out.size, coeffs_out.size // 384)   # in macroblocks; the C side counts pictures
    # capacity in pictures is only known once the SPS is parsed: pass macroblocks / 1 and let the C side check per picture
    n = lib.dryv_h264_parse_all_into(buf.ctypes.data, buf.size, int(max_pictures), int(threads), mbs_out.ctypes.data,
                                     coeffs_out.ctypes.data, _capacity_pictures(data, capacity), C.addressof(fp), info.ctypes.data)
    if n <= 0:
        raise H264Error(lib.dryv_h264_last_error().decode())
    return fp, int(n), {"skipped": int(info[1]), "tails_ok": int(info[2])}


def stream_params(data):
    """(fp, n_coded_slices) from the stream's parameter sets alone: what a caller needs to size its batch buffers."""
    lib = _load()
    buf = np.frombuffer(data, dtype=np.uint8)
    fp = abi.FrameParams()
    n = lib.dryv_h264_stream_params(buf.ctypes.data, buf.size, C.addressof(fp))
    if n <= 0:
        raise H264Error(lib.dryv_h264_last_error().decode())
    return fp, int(n)


def _capacity_pictures(data, capacity_mbs):
    fp, _ = stream_params(data)
    return capacity_mbs // (fp.pic_width_in_mbs * fp.pic_height_in_mbs)


def bin_log(data, picture=0):
    """Test hook: (bins, W, H, transform_8x8_mode_flag, slice_qp) of the stream's `picture`-th intra picture: every CABAC bin the
    parser decoded, in order, as value | kind << 1 (kind 0 context-coded, 1 bypass, 2 terminate). Input of the independent
    restatement of the slice-data syntax in oracle/islice_syntax.py."""
    lib = _load()
    buf = np.frombuffer(data, dtype=np.uint8)
    hdr = np.zeros(4, dtype=np.int32)
    need = lib.dryv_h264_bin_log(buf.ctypes.data, buf.size, int(picture), None, 0, hdr.ctypes.data)
    if need == 0:
        raise H264Error(lib.dryv_h264_last_error().decode())
    out = np.zeros(-need, dtype=np.uint8)
    got = lib.dryv_h264_bin_log(buf.ctypes.data, buf.size, int(picture), out.ctypes.data, out.size, hdr.ctypes.data)
    assert got == out.size
    return out, int(hdr[0]), int(hdr[1]), int(hdr[2]), int(hdr[3])
