"""Builds the native libraries of the package in-tree (dryv_amd/lib/*.so).

* libdryv_recon.so — the product: gfx950 HIP kernels + the extern "C" boundary
  (include/dryv_recon.h). Cross-compiles without a GPU.
* libdryv_synth.so — host-only synthetic batch generator (tests / bench inputs).
"""
import os
import shutil
import subprocess

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
LIB = os.path.join(HERE, "lib")
RECON_SO = os.path.join(LIB, "libdryv_recon.so")
SYNTH_SO = os.path.join(LIB, "libdryv_synth.so")

HIPCC = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"


def _stale(target, sources):
    if not os.path.exists(target):
        return True
    t = os.path.getmtime(target)
    return any(os.path.getmtime(s) > t for s in sources)


def _run(cmd):
    """Runs a compiler command whose output file follows "-o". The output is written under a private name and moved
    into place atomically: several ranks of one job may find the same library stale at the same time."""
    cmd = list(cmd)
    i = cmd.index("-o") + 1
    final = cmd[i]
    cmd[i] = "%s.tmp%d" % (final, os.getpid())
    r = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)
    if r.returncode != 0:
        try:
            os.unlink(cmd[i])
        except OSError:
            pass
        raise RuntimeError("build failed: %s\n%s" % (" ".join(cmd), r.stdout))
    os.replace(cmd[i], final)
    return r.stdout


def build_recon(force=False):
    srcs = [os.path.join(CSRC, f) for f in ("recon_band.hip", "output_pack.hip", "deblock.hip", "recon_api.hip")]
    deps = srcs + [os.path.join(CSRC, f) for f in ("band_launch.h", "band_kernel.h", "band_diag.h", "wave.h", "output_pack.h", "deblock_kernel.h", "deblock_kernel_params.h",
                                                   "deblock_params.h", "deblock_launch.h",
                                                   "kparams.h", "recon_params.h")] + [
                   os.path.join(HERE, "..", "include", "dryv_recon.h")]
    os.makedirs(LIB, exist_ok=True)
    if force or _stale(RECON_SO, deps):
        _run([HIPCC, "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-shared",
              "-o", RECON_SO] + srcs)
    return RECON_SO


def build_synth(force=False):
    src = os.path.join(CSRC, "synth.c")
    deps = [src, os.path.join(HERE, "..", "include", "dryv_recon.h")]
    os.makedirs(LIB, exist_ok=True)
    if force or _stale(SYNTH_SO, deps):
        _run(["gcc", "-O2", "-fPIC", "-std=c11", "-Wall", "-shared", "-o", SYNTH_SO, src])
    return SYNTH_SO


HARNESS = os.path.join(LIB, "frame_harness")


def build_harness(force=False):
    """C++ host mirror of the reference's Frame interface (host/frame.hpp) + its harness binary."""
    host = os.path.join(HERE, "host")
    srcs = [os.path.join(host, f) for f in ("frame_harness.cpp", "frame.hpp", "h264_islice.hpp", "cabac_tables.inc")]
    build_recon(force)
    if force or _stale(HARNESS, srcs + [RECON_SO]):
        _run(["g++", "-O2", "-std=c++17", "-Wall", "-pthread", "-o", HARNESS, srcs[0], "-L" + LIB, "-ldryv_recon",
              "-Wl,-rpath," + LIB, "-Wl,-rpath,/opt/rocm/lib", "-L/opt/rocm/lib", "-lamdhip64"])
    return HARNESS


def build_h264(force=False):
    from . import h264
    return h264.build(force)


def build_all(force=False):
    return [build_recon(force), build_synth(force), build_harness(force), build_h264(force)]


if __name__ == "__main__":
    for p in build_all(force=True):
        print("built", p)
