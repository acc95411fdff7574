"""The C-ABI library: loads without a GPU, exports every symbol include/dryv_recon.h declares,
and the ctypes/numpy record layouts match the header byte for byte. No compute calls here."""
import ctypes as C
import os
import re
import subprocess
import sys
import tempfile

import numpy as np
import pytest

from dryv_amd import _build, abi

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HEADER = os.path.join(ROOT, "include", "dryv_recon.h")


def declared_functions():
    src = open(HEADER).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    return sorted(set(re.findall(r"\b(dryv_[a-z0-9_]+)\s*\(", src)))


def test_library_builds_and_exports_every_declared_symbol():
    _build.build_recon()
    lib = C.CDLL(_build.RECON_SO)
    names = declared_functions()
    assert len(names) >= 14
    for n in names:
        assert hasattr(lib, n), "libdryv_recon.so does not export %s" % n
    # and the ctypes table covers exactly the header
    assert sorted(abi.SYMBOLS) == names


def test_record_layout_matches_header():
    prog = r'''
#include <stdio.h>
#include <stddef.h>
#include "dryv_recon.h"
int main(void) {
  printf("%zu %zu %zu %zu %zu %zu\n", sizeof(dryv_frame_params), offsetof(dryv_frame_params, scaling_list4x4),
         offsetof(dryv_frame_params, scaling_list8x8), offsetof(dryv_frame_params, chroma_qp_index_offset),
         offsetof(dryv_frame_params, second_chroma_qp_index_offset), offsetof(dryv_frame_params, transform_8x8_mode_flag));
  printf("%zu %zu %zu %zu %zu\n", sizeof(dryv_mb_desc), offsetof(dryv_mb_desc, qp), offsetof(dryv_mb_desc, prev_flags),
         offsetof(dryv_mb_desc, rem_modes), offsetof(dryv_mb_desc, nz_mask));
  return 0;
}
'''
    with tempfile.TemporaryDirectory() as td:
        c = os.path.join(td, "l.c")
        open(c, "w").write(prog)
        exe = os.path.join(td, "l")
        subprocess.check_call(["gcc", "-std=c11", "-I", os.path.join(ROOT, "include"), c, "-o", exe])
        out = subprocess.check_output([exe], text=True).split("\n")
    fpv = [int(x) for x in out[0].split()]
    mbv = [int(x) for x in out[1].split()]
    F = abi.FrameParams
    assert fpv == [C.sizeof(F), F.scaling_list4x4.offset, F.scaling_list8x8.offset, F.chroma_qp_index_offset.offset,
                   F.second_chroma_qp_index_offset.offset, F.transform_8x8_mode_flag.offset]
    assert fpv[0] == 496
    M = abi.MbDesc
    assert mbv == [C.sizeof(M), M.qp.offset, M.prev_flags.offset, M.rem_modes.offset, M.nz_mask.offset]
    d = abi.MB_DESC_DTYPE
    assert [d.itemsize, d.fields["qp"][1], d.fields["prev_flags"][1], d.fields["rem_modes"][1],
            d.fields["nz_mask"][1]] == mbv


def test_host_only_entry_points():
    lib = abi.load_library()
    assert lib.dryv_recon_abi_version() == 1
    fp = abi.make_frame_params(120, 68)
    assert lib.dryv_recon_frame_bytes(C.byref(fp)) == 3133440          # bytes per 1080p frame
    assert lib.dryv_recon_frame_bytes(C.byref(abi.make_frame_params(240, 135))) == 12441600
    assert lib.dryv_recon_frame_bytes(None) == 0
    assert b"CPU" in lib.dryv_recon_strerror(abi.DRYV_E_NODEVICE)
    # math.rs:109-125
    assert lib.dryv_math_clamp(300, 0, 255) == 255 and lib.dryv_math_clamp(-1, 0, 255) == 0
    assert lib.dryv_math_inverse_raster_scan(125, 16, 16, 1920, 0) == 80
    assert lib.dryv_math_inverse_raster_scan(125, 16, 16, 1920, 1) == 16


def test_no_cpu_fallback_without_device():
    """Without a GPU the library must fail loudly, not compute on the CPU."""
    code = ("import ctypes as C, sys; sys.path.insert(0, %r); from dryv_amd import abi; "
            "lib = abi.load_library(); h = C.c_void_p(); st = lib.dryv_recon_create(C.byref(h), 0); "
            "print(st)" % ROOT)
    env = dict(os.environ, HIP_VISIBLE_DEVICES="-1", ROCR_VISIBLE_DEVICES="-1")
    out = subprocess.check_output([sys.executable, "-c", code], env=env, text=True).strip()
    assert int(out.split()[-1]) == abi.DRYV_E_NODEVICE


def test_product_does_not_reference_the_oracle():
    """Nothing under dryv_amd/ (sources or the built .so) may import, link or mention oracle/."""
    pkg = os.path.join(ROOT, "dryv_amd")
    for dp, _, files in os.walk(pkg):
        for fn in files:
            if fn.endswith((".py", ".hip", ".h", ".hpp", ".cpp", ".c")):
                txt = open(os.path.join(dp, fn), errors="ignore").read()
                assert "libdryv_oracle" not in txt and "import oracle" not in txt and "from oracle" not in txt, fn
                assert "dryv_oracle_" not in txt, fn
    syms = subprocess.check_output(["nm", "-D", _build.RECON_SO], text=True)
    assert "dryv_oracle" not in syms
    needed = subprocess.check_output(["readelf", "-d", _build.RECON_SO], text=True)
    assert "oracle" not in needed


def test_parameter_bounds_are_checked_without_a_device():
    """The kernels use 32-bit offsets inside a frame and 31-bit macroblock counts; anything beyond must be refused by
    the boundary, not wrapped on the device (round-1 finding: H was unbounded)."""
    lib = abi.load_library()
    ok = abi.make_frame_params(120, 68)
    assert lib.dryv_recon_check_params(ok, 300) == abi.DRYV_OK
    assert lib.dryv_recon_check_params(ok, 0) == abi.DRYV_E_INVALID
    assert lib.dryv_recon_check_params(abi.make_frame_params(1024, 5461), 1) == abi.DRYV_OK        # 4,294,574,080 B? no: just under
    assert lib.dryv_recon_check_params(abi.make_frame_params(1024, 5462), 1) == abi.DRYV_E_INVALID   # 768 B/MB x W x H >= 2^32
    assert lib.dryv_recon_check_params(abi.make_frame_params(1024, 20000), 1) == abi.DRYV_E_INVALID
    assert lib.dryv_recon_check_params(abi.make_frame_params(1025, 4), 1) == abi.DRYV_E_INVALID
    assert lib.dryv_recon_check_params(ok, 263172) == abi.DRYV_OK          # 8160 x 263172 = 2^31 - 4,288 macroblocks
    assert lib.dryv_recon_check_params(ok, 263173) == abi.DRYV_E_INVALID   # >= 2^31
    bad = abi.make_frame_params(4, 4)
    bad.bit_depth_y = 10
    assert lib.dryv_recon_check_params(bad, 1) == abi.DRYV_E_UNSUPPORTED


def test_output_stage_geometry_and_validation():
    """dryv_recon_output_bytes (pure host code): display size after cropping, both formats the same byte count, odd or
    oversized crops and unknown formats rejected (0)."""
    lib = abi.load_library()
    fp = abi.make_frame_params(120, 68)                       # 1920 x 1088 coded
    full = lib.dryv_recon_output_bytes(C.byref(fp), C.byref(abi.make_output_desc()))
    assert full == 1920 * 1088 * 3 // 2 == lib.dryv_recon_frame_bytes(C.byref(fp))
    od = abi.make_output_desc(abi.OUT_NV12, (0, 0, 0, 8))     # 1080p: frame_crop_bottom_offset = 4 units
    assert lib.dryv_recon_output_bytes(C.byref(fp), C.byref(od)) == 1920 * 1080 * 3 // 2
    od = abi.make_output_desc(abi.OUT_I420, (2, 6, 4, 10))
    assert lib.dryv_recon_output_bytes(C.byref(fp), C.byref(od)) == (1920 - 8) * (1088 - 14) * 3 // 2
    for bad in (abi.make_output_desc(abi.OUT_I420, (1, 0, 0, 0)), abi.make_output_desc(abi.OUT_I420, (0, 0, 0, 3)),
                abi.make_output_desc(abi.OUT_I420, (960, 960, 0, 0)), abi.make_output_desc(abi.OUT_I420, (0, 0, 1000, 88)),
                abi.make_output_desc(7, (0, 0, 0, 0))):
        assert lib.dryv_recon_output_bytes(C.byref(fp), C.byref(bad)) == 0
    assert C.sizeof(abi.OutputDesc) == 12


def _pred_table(n):
    """recon_params.h build_pred_table, restated: per (mode, x, y) which of E / F (3 taps) / G (2 taps) at which index of the
    edge line [left column bottom..top | corner | top row + top right]."""
    E, F, G = 0, 1, 2
    C, T0, L0 = n, n + 1, n - 1
    t = {}
    for y in range(n):
        for x in range(n):
            t[0, x, y] = (E, T0 + x)
            t[1, x, y] = (E, L0 - y)
            t[3, x, y] = (F, T0 + 1 + x + y)
            t[4, x, y] = (F, C + x - y)
            z, k = 2 * x - y, x - (y >> 1)
            t[5, x, y] = ((F if z & 1 else G), C + k) if z >= 0 else (F, C) if z == -1 else (F, T0 - y + 2 * x)
            z, k = 2 * y - x, y - (x >> 1)
            t[6, x, y] = ((F, C - k) if z & 1 else (G, L0 - k)) if z >= 0 else (F, C) if z == -1 else (F, L0 + x - 2 * y)
            t[7, x, y] = (F, T0 + 1 + x + (y >> 1)) if y & 1 else (G, T0 + x + (y >> 1))
            z, k, zmax = x + 2 * y, y + (x >> 1), 2 * n - 3
            t[8, x, y] = ((F if z & 1 else G), n - 2 - k) if z < zmax else (F, 0) if z == zmax else (E, 0)
    return t


@pytest.mark.parametrize("n,jmax", [(4, 12), (8, 24)])
def test_prediction_tables_fit_aligned_windows(n, jmax):
    """The block chain (Intra4x4) and BACK8 (Intra8x8) read a pixel PAIR's reference bytes from ONE 4-byte-aligned 8-byte window of
    the edge line and pick them with v_perm_b32 (band_kernel.h build_tables: T_T4W, T_T8S / T_T8O): every pair's taps, as the
    table builders derive them from build_pred_table, must lie within 8 bytes of the aligned start."""
    E, F, G = 0, 1, 2
    t = _pred_table(n)
    for mode in (0, 1, 3, 4, 5, 6, 7, 8):
        for y in range(n):
            for pair in range(n // 2):
                taps = []
                for x in (2 * pair, 2 * pair + 1):
                    s, j = t[mode, x, y]
                    taps += [min(max(v, 0), jmax) for v in ((j - 1 if s == F else j), (j + 1 if s == G else j), (j + 1 if s == F else j))]
                off = min(taps) & ~3
                assert max(taps) - off <= 7, (n, mode, y, pair, taps)
