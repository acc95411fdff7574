#!/usr/bin/env python3
"""Compiler-reported resources of every kernel of a HIP source (hipcc -Rpass-analysis=kernel-resource-usage; cross-compiles
without a GPU). Prints / returns {kernel: {VGPRs, VGPRs Spill, SGPRs Spill, ScratchSize, Occupancy, ...}}.
usage: tools/resource_usage.py dryv_amd/csrc/recon_band.hip [-D...]"""
import os, re, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def resource_usage(src, flags=()):
    cmd = ["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-I%s/include" % ROOT,
           "-Rpass-analysis=kernel-resource-usage", "-c", "--cuda-device-only", "-o", "/dev/null", src, *flags]
    r = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)
    if r.returncode != 0:
        raise RuntimeError(r.stdout[-3000:])
    out, cur = {}, None
    for line in r.stdout.splitlines():
        m = re.search(r"remark:\s+Function Name: (\S+)", line)
        if m:
            name = subprocess.run(["c++filt", m.group(1)], stdout=subprocess.PIPE, text=True).stdout.strip()
            cur = out.setdefault(name, {})
            continue
        m = re.search(r"remark:\s+([A-Za-z ]+?)(?: \[bytes/lane\]| \[waves/SIMD\]| \[bytes/workgroup\])?: (\d+)", line)
        if m and cur is not None:
            cur[m.group(1).strip()] = int(m.group(2))
    return out


if __name__ == "__main__":
    for k, v in resource_usage(sys.argv[1], sys.argv[2:]).items():
        print(k[:70])
        print("   ", ", ".join("%s %d" % kv for kv in v.items()))
