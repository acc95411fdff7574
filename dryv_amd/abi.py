"""ctypes view of include/dryv_recon.h — the C ABI of the reconstruction backend.

The record layouts here must stay byte-identical to the header; tests/test_abi.py checks sizes,
offsets and that every declared symbol is exported by libdryv_recon.so.
"""
import ctypes as C
import os

import numpy as np

from . import _build

DRYV_OK = 0
DRYV_E_INVALID = -1
DRYV_E_UNSUPPORTED = -2
DRYV_E_DEVICE = -3
DRYV_E_NOMEM = -4
DRYV_E_STATE = -5
DRYV_E_NODEVICE = -6

COEFFS_PER_MB = 384


class FrameParams(C.Structure):
    """dryv_frame_params (496 bytes)."""
    _fields_ = [
        ("pic_width_in_mbs", C.c_uint16),
        ("pic_height_in_mbs", C.c_uint16),
        ("chroma_array_type", C.c_uint8),
        ("bit_depth_y", C.c_uint8),
        ("bit_depth_c", C.c_uint8),
        ("chroma_qp_index_offset", C.c_int8),
        ("second_chroma_qp_index_offset", C.c_int8),
        ("constrained_intra_pred_flag", C.c_uint8),
        ("transform_8x8_mode_flag", C.c_uint8),
        ("reserved", C.c_uint8 * 5),
        ("scaling_list4x4", (C.c_uint8 * 16) * 6),
        ("scaling_list8x8", (C.c_uint8 * 64) * 6),
    ]


class MbDesc(C.Structure):
    """dryv_mb_desc (16 bytes)."""
    _fields_ = [
        ("mb_kind", C.c_uint8),
        ("i16_pred_mode", C.c_uint8),
        ("intra_chroma_pred_mode", C.c_uint8),
        ("qp", C.c_uint8),
        ("prev_flags", C.c_uint16),
        ("rem_modes", C.c_uint8 * 8),
        ("nz_mask", C.c_uint16),
    ]


class OutputDesc(C.Structure):
    """dryv_output_desc (12 bytes): output format and cropping rectangle (luma samples) of the output stage."""
    _fields_ = [("format", C.c_uint8), ("reserved", C.c_uint8 * 3), ("crop_left", C.c_uint16), ("crop_right", C.c_uint16),
                ("crop_top", C.c_uint16), ("crop_bottom", C.c_uint16)]


OUT_I420, OUT_NV12 = 0, 1


class DeblockParams(C.Structure):
    """dryv_deblock_params (4 bytes): the slice header's deblocking syntax elements."""
    _fields_ = [("disable_deblocking_filter_idc", C.c_uint8), ("slice_alpha_c0_offset_div2", C.c_int8),
                ("slice_beta_offset_div2", C.c_int8), ("reserved", C.c_uint8)]


def make_deblock_params(disable_idc=0, alpha_div2=0, beta_div2=0):
    dp = DeblockParams()
    dp.disable_deblocking_filter_idc, dp.slice_alpha_c0_offset_div2, dp.slice_beta_offset_div2 = disable_idc, alpha_div2, beta_div2
    return dp


def make_output_desc(fmt=OUT_I420, crop=(0, 0, 0, 0)):
    od = OutputDesc()
    od.format = fmt
    od.crop_left, od.crop_right, od.crop_top, od.crop_bottom = crop
    return od


# numpy view of the same 16 bytes, for bulk handling
MB_DESC_DTYPE = np.dtype([
    ("mb_kind", "u1"), ("i16_pred_mode", "u1"), ("intra_chroma_pred_mode", "u1"), ("qp", "u1"),
    ("prev_flags", "<u2"), ("rem_modes", "u1", (8,)), ("nz_mask", "<u2"),
])
assert MB_DESC_DTYPE.itemsize == 16 and C.sizeof(MbDesc) == 16 and C.sizeof(FrameParams) == 496

# every function include/dryv_recon.h declares: name -> (restype, argtypes)
SYMBOLS = {
    "dryv_recon_frame_bytes": (C.c_size_t, [C.POINTER(FrameParams)]),
    "dryv_recon_check_params": (C.c_int, [C.POINTER(FrameParams), C.c_uint32]),
    "dryv_recon_create": (C.c_int, [C.POINTER(C.c_void_p), C.c_int]),
    "dryv_recon_destroy": (None, [C.c_void_p]),
    "dryv_recon_submit": (C.c_int, [C.c_void_p, C.POINTER(FrameParams), C.c_uint32, C.c_void_p, C.c_void_p]),
    "dryv_recon_wait": (C.c_int, [C.c_void_p, C.c_void_p, C.c_size_t]),
    "dryv_recon_submit_host": (C.c_int, [C.c_void_p, C.POINTER(FrameParams), C.c_uint32, C.c_void_p, C.c_void_p,
                                         C.c_void_p, C.c_size_t]),
    "dryv_recon_alloc_host": (C.c_void_p, [C.c_size_t]),
    "dryv_recon_free_host": (None, [C.c_void_p]),
    "dryv_recon_submit_device": (C.c_int, [C.c_void_p, C.POINTER(FrameParams), C.c_uint32, C.c_void_p,
                                           C.c_void_p, C.c_void_p]),
    "dryv_recon_sync": (C.c_int, [C.c_void_p]),
    "dryv_recon_submit_device_queued": (C.c_int, [C.c_void_p, C.POINTER(FrameParams), C.c_uint32, C.c_void_p,
                                                  C.c_void_p, C.c_void_p]),
    "dryv_recon_kernel_ms_stats": (C.c_int, [C.c_void_p, C.c_uint32, C.POINTER(C.c_float), C.POINTER(C.c_float),
                                             C.POINTER(C.c_float)]),
    "dryv_recon_wide_rerun_stats": (C.c_int, [C.c_void_p, C.POINTER(C.c_int), C.POINTER(C.c_int)]),
    "dryv_recon_set_queue_lanes": (C.c_int, [C.c_void_p, C.c_int]),
    "dryv_recon_output_bytes": (C.c_size_t, [C.POINTER(FrameParams), C.POINTER(OutputDesc)]),
    "dryv_recon_pack_device": (C.c_int, [C.c_void_p, C.POINTER(FrameParams), C.c_uint32, C.c_void_p, C.POINTER(OutputDesc),
                                         C.c_void_p]),
    "dryv_recon_wait_packed": (C.c_int, [C.c_void_p, C.POINTER(OutputDesc), C.c_void_p, C.c_size_t]),
    "dryv_recon_wait_filtered": (C.c_int, [C.c_void_p, C.POINTER(DeblockParams), C.POINTER(OutputDesc), C.c_void_p, C.c_size_t]),
    "dryv_recon_deblock_device": (C.c_int, [C.c_void_p, C.POINTER(FrameParams), C.POINTER(DeblockParams), C.c_uint32, C.c_void_p,
                                            C.c_void_p]),
    "dryv_recon_last_kernel_ms": (C.c_int, [C.c_void_p, C.POINTER(C.c_float)]),
    "dryv_recon_stream": (C.c_void_p, [C.c_void_p]),
    "dryv_recon_strerror": (C.c_char_p, [C.c_int]),
    "dryv_recon_last_device_error": (C.c_char_p, [C.c_void_p]),
    "dryv_recon_abi_version": (C.c_int, []),
    "dryv_math_clamp": (C.c_int64, [C.c_int64, C.c_int64, C.c_int64]),
    "dryv_math_inverse_raster_scan": (C.c_int64, [C.c_int64] * 5),
}

_lib = None


def _preload_torch_hip_runtime():
    """PyTorch-ROCm wheels bundle their own libamdhip64.so.7 / libhsa-runtime64; libdryv_recon.so links
    the same SONAME from /opt/rocm. A process must only ever hold ONE HIP runtime, and torch cannot
    initialise on a foreign one, so when torch is installed its copy is loaded first (without
    importing torch) and libdryv_recon.so binds to it. Without torch the system runtime is used."""
    import importlib.util
    import sys
    if "torch" in sys.modules:
        return
    try:
        spec = importlib.util.find_spec("torch")
    except (ImportError, ValueError):
        spec = None
    if spec is None or not spec.submodule_search_locations:
        return
    libdir = os.path.join(list(spec.submodule_search_locations)[0], "lib")
    for name in ("libhsa-runtime64.so", "libamdhip64.so"):
        cand = os.path.join(libdir, name)
        if os.path.exists(cand):
            try:
                C.CDLL(cand, mode=C.RTLD_GLOBAL)
            except OSError:
                return


def load_library(path=None):
    """Loads libdryv_recon.so (building it first if the sources are newer). Raises if the HIP
    extension is missing: there is no Python or CPU fallback for the path."""
    global _lib
    if _lib is not None and path is None:
        return _lib
    # DRYV_RECON_LIB: tuning only (tools/pmc_variant.sh profiles -DDRYV_SKIP_* builds through bench.py)
    so = path or os.environ.get("DRYV_RECON_LIB") or _build.RECON_SO
    if path is None and not os.path.exists(so):
        so = _build.build_recon()
    _preload_torch_hip_runtime()
    lib = C.CDLL(so)
    for name, (res, args) in SYMBOLS.items():
        fn = getattr(lib, name)  # AttributeError if a declared symbol is not exported
        fn.restype = res
        fn.argtypes = args
    if path is None:
        _lib = lib
    return lib


def make_frame_params(width_mbs, height_mbs, cqo_cb=0, cqo_cr=None, transform_8x8=False,
                      scaling4x4=None, scaling8x8=None, constrained_intra=False):
    """Builds a dryv_frame_params. Scaling lists default to flat 16 (reference: header.rs:330)."""
    fp = FrameParams()
    fp.pic_width_in_mbs = width_mbs
    fp.pic_height_in_mbs = height_mbs
    fp.chroma_array_type = 1
    fp.bit_depth_y = 8
    fp.bit_depth_c = 8
    fp.chroma_qp_index_offset = cqo_cb
    fp.second_chroma_qp_index_offset = cqo_cb if cqo_cr is None else cqo_cr
    fp.constrained_intra_pred_flag = int(constrained_intra)
    fp.transform_8x8_mode_flag = int(transform_8x8)
    for l in range(6):
        for k in range(16):
            fp.scaling_list4x4[l][k] = 16 if scaling4x4 is None else int(scaling4x4[l][k])
        for k in range(64):
            fp.scaling_list8x8[l][k] = 16 if scaling8x8 is None else int(scaling8x8[l][k])
    return fp


def strerror(status):
    return load_library().dryv_recon_strerror(status).decode()
