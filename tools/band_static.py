#!/usr/bin/env python3
"""Analysis only: vector / scalar / LDS instructions of band_kernel<false,false> between the phase marks of the source
(-DDRYV_BAND_MARK turns every PH(k) into an assembly comment). Linear in the assembly: a section that the compiler laid
out elsewhere is counted where it landed. Usage: tools/band_static.py [extra -D flags]"""
import re, subprocess, sys, os, collections
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
out = "/tmp/band_static.s"
cmd = ["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-DDRYV_BAND_MARK", "-I%s/include" % root,
       "-S", "--cuda-device-only", "-o", out, "%s/dryv_amd/csrc/recon_band.hip" % root] + sys.argv[1:]
subprocess.check_call(cmd)
lines = open(out).read().split("\n")
start = [i for i, l in enumerate(lines) if l.startswith(os.environ.get("BAND_SYM", "_ZN4dryv11band_kernelILb0ELb0E"))][0]
end = [i for i, l in enumerate(lines) if i > start and "s_endpgm" in l][0]
src = open("%s/dryv_amd/csrc/band_kernel.h" % root).read().split("\n")
cur = "prologue"
acc = collections.OrderedDict()
for l in lines[start:end]:
    m = re.search(r"; DRYV_MARK (\d+) (\d+)", l)
    if m:
        ln = int(m.group(1))
        cur = "after L%d %s" % (ln, src[ln - 1].strip()[:60])
        continue
    t = l.strip()
    if not t or t.startswith((".", ";")) or t.endswith(":"):
        continue
    c = acc.setdefault(cur, collections.Counter())
    op = t.split()[0]
    c["valu" if op.startswith("v_") else "salu" if op.startswith("s_") else "lds" if op.startswith("ds_") else "mem"] += 1
for k, c in acc.items():
    print("%5d valu %5d salu %4d lds %3d mem   %s" % (c["valu"], c["salu"], c["lds"], c["mem"], k))
