"""The registers, spills and scratch the compiler reports for every build of the band kernel, against the numbers DESIGN.md
(section 4.2) documents. hipcc cross-compiles for gfx950 without a GPU; tools/resource_usage.py parses
-Rpass-analysis=kernel-resource-usage. A change that makes a build spill more, or costs it a wave per SIMD, fails here
before it reaches the GPU box."""
import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tools"))
from resource_usage import resource_usage  # noqa: E402

# build -> (VGPRs at most, waves per SIMD at least, spilled VGPRs at most, scratch bytes per lane at most, spilled SGPRs at
# most). The spilled SGPRs are lanes of one VGPR (v_writelane / v_readlane), all of them in the FRONT wave and nearly all
# outside its step loop (per task: claim, geometry, pointers): DESIGN.md section 4.
DOCUMENTED = {
    "void dryv::band_kernel<false, false>": (80, 6, 0, 0, 46),     # the bench configuration: no 8x8 transform
    "void dryv::band_kernel<true, false>": (96, 5, 0, 0, 54),      # streams with the 8x8 transform
    "void dryv::band_kernel<false, true>": (128, 4, 0, 0, 60),     # WIDE builds: re-run of a flagged batch only
    "void dryv::band_kernel<true, true>": (128, 4, 0, 0, 58),
}


@pytest.fixture(scope="module")
def usage():
    return resource_usage(os.path.join(ROOT, "dryv_amd", "csrc", "recon_band.hip"))


@pytest.mark.parametrize("build", sorted(DOCUMENTED))
def test_band_kernel_build_resources(usage, build):
    got = [v for k, v in usage.items() if k.startswith(build + "(")]
    assert len(got) == 1, sorted(usage)
    got = got[0]
    vgprs, occupancy, spills, scratch, sgpr_spills = DOCUMENTED[build]
    assert got["VGPRs"] <= vgprs, got
    assert got["Occupancy"] >= occupancy, got
    assert got["VGPRs Spill"] <= spills, got
    assert got["ScratchSize"] <= scratch, got
    assert got["SGPRs Spill"] <= sgpr_spills, got
    assert got["AGPRs"] == 0


def test_other_kernels_do_not_spill():
    for src in ("output_pack.hip", "deblock.hip"):
        for name, got in resource_usage(os.path.join(ROOT, "dryv_amd", "csrc", src)).items():
            assert got["VGPRs Spill"] == 0 and got["ScratchSize"] == 0, (name, got)
