"""ctypes loader for the band-kernel CPU emulator (tests/emu/band_emu.cpp). Test infrastructure only."""
import ctypes as C
import os
import subprocess

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
SO = os.path.join(HERE, "libdryv_emu.so")
CLANG = "/opt/rocm/lib/llvm/bin/clang++"
_libs = {}


def build(force=False, defs=()):
    """defs: extra -D flags (the kernel's build-time variants, e.g. staging widths); one library per set."""
    SO = os.path.join(HERE, "libdryv_emu%s.so" % "".join("_" + d.replace("-D", "").replace("=", "") for d in defs))
    csrc = os.path.join(HERE, "..", "..", "dryv_amd", "csrc")
    deps = [os.path.join(HERE, "band_emu.cpp"), os.path.join(HERE, "wave_emu.h")] + [os.path.join(csrc, f) for f in
                                                    ("band_kernel.h", "band_diag.h", "wave.h", "kparams.h", "recon_params.h", "deblock_kernel.h",
                                                     "deblock_kernel_params.h", "deblock_params.h")]
    if force or not os.path.exists(SO) or any(os.path.getmtime(d) > os.path.getmtime(SO) for d in deps):
        r = subprocess.run([CLANG, "-x", "c++", "-std=c++17", "-O1", "-g", "-fPIC", "-shared", "-DDRYV_EMU", "-Wall",
                            "-Wno-unused-function", *defs, "-o", SO, deps[0]], stdout=subprocess.PIPE,
                           stderr=subprocess.STDOUT, text=True)
        if r.returncode != 0:
            raise RuntimeError("emulator build failed:\n" + r.stdout)
    return SO


def reconstruct(fp, n_frames, mbs, coeffs, n_teams=1, first=0, order=1, defs=()):
    defs = tuple(defs)
    if defs not in _libs:
        lib = C.CDLL(build(defs=defs))
        lib.dryv_emu_reconstruct.restype = C.c_int
        lib.dryv_emu_reconstruct.argtypes = [C.c_void_p, C.c_uint32, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p,
                                             C.c_int, C.c_int, C.c_int]
        _libs[defs] = lib
    _lib = _libs[defs]
    mbs = np.ascontiguousarray(mbs)
    coeffs = np.ascontiguousarray(coeffs, dtype=np.int16)
    n_mbs = n_frames * fp.pic_width_in_mbs * fp.pic_height_in_mbs
    # padded: the kernel's 16-byte coefficient loads may start up to 2 bytes before a block's first AC entry
    yuv = np.zeros(n_mbs * 384, dtype=np.uint8)
    status = C.c_uint(0)
    st = _lib.dryv_emu_reconstruct(C.addressof(fp), n_frames, mbs.ctypes.data, coeffs.ctypes.data, yuv.ctypes.data,
                                   C.addressof(status), n_teams, first, order)
    assert st == 0, st
    return int(status.value), yuv


def deblock(fp, dp, n_frames, mbs, yuv, n_waves=1, first=0, order=1):
    """The deblocking kernel's source (dryv_amd/csrc/deblock_kernel.h) under the lane emulator. Returns (status, filtered copy)."""
    if () not in _libs:
        lib = C.CDLL(build())
        lib.dryv_emu_reconstruct.restype = C.c_int
        lib.dryv_emu_reconstruct.argtypes = [C.c_void_p, C.c_uint32, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p,
                                             C.c_int, C.c_int, C.c_int]
        _libs[()] = lib
    lib = _libs[()]
    lib.dryv_emu_deblock.restype = C.c_int
    lib.dryv_emu_deblock.argtypes = [C.c_void_p, C.c_void_p, C.c_uint32, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int]
    mbs = np.ascontiguousarray(mbs)
    out = np.array(yuv, dtype=np.uint8, copy=True)
    status = C.c_uint(0)
    st = lib.dryv_emu_deblock(C.addressof(fp), C.addressof(dp), n_frames, mbs.ctypes.data, out.ctypes.data, C.addressof(status),
                              n_waves, first, order)
    assert st == 0, st
    return int(status.value), out
