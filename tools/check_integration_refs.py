#!/usr/bin/env python3
"""Mechanical validation of INTEGRATION.md against the reference tree (build container only: /root/reference does not
travel). The Rust shim cannot be compiled here (no rustc), so this is what can be checked: every `path.rs:line[-line]` the
document cites must name a file that exists under the reference's src/ with those lines in range, and a Rust identifier
quoted in backticks or used in code on the same line as a citation must occur in that file (within a window around the
cited lines when it is not a declaration elsewhere in the file).
usage: tools/check_integration_refs.py [doc ...]   (default: INTEGRATION.md)   exit status 1 on any finding"""
import os
import re
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REF = "/root/reference/src"
CITE = re.compile(r"(?<![\w/.])((?:[\w]+/)*[\w]+\.rs):(\d+)(?:-(\d+))?(?:,\s*(\d+)-(\d+))?")
IDENT = re.compile(r"`([A-Za-z_][\w:.<>&\[\]; ]*?)(?:\(.*?\))?`")
# words in backticks that are not identifiers of the reference (this library's own names, Rust keywords, prose)
OWN = re.compile(r"^(dryv_|DRYV_|test_|Decoder(Error)?::Recon|i16|u8|u16|i8|isize|usize|todo!|\.take|Result|write_all|extern|frame_crop_|more_rbsp_data|README)")


def ref_files():
    out = {}
    for d, _, fs in os.walk(REF):
        for f in fs:
            if f.endswith(".rs"):
                p = os.path.join(d, f)
                out[os.path.relpath(p, REF)] = p
    return out


def resolve(cite, files):
    """A citation names a path suffix ('frame/mod.rs', 'decoder.rs', 'pps.rs'): it must match exactly one file, or several
    of which exactly one lies on the decode path (src/video/...)."""
    cite = cite[4:] if cite.startswith("src/") else cite
    hits = [rel for rel in files if rel == cite or rel.endswith("/" + cite)]
    if len(hits) > 1:
        vid = [h for h in hits if h.startswith("video/")]
        if len(vid) == 1:
            hits = vid
    return hits


def check(doc, files, idents):
    findings = []
    text = open(doc).read().splitlines()
    n_cites = 0
    for ln, line in enumerate(text, 1):
        cites = list(CITE.finditer(line))
        for m in cites:
            n_cites += 1
            hits = resolve(m.group(1), files)
            if len(hits) != 1:
                findings.append("%s:%d: `%s` names %d files of the reference" % (doc, ln, m.group(1), len(hits)))
                continue
            src = open(files[hits[0]], errors="replace").read().splitlines()
            for a, b in ((m.group(2), m.group(3)), (m.group(4), m.group(5))):
                if a is None:
                    continue
                lo, hi = int(a), int(b or a)
                if not (1 <= lo <= hi <= len(src)):
                    findings.append("%s:%d: %s:%d-%d is outside the file (%d lines)" % (doc, ln, hits[0], lo, hi, len(src)))
        if cites and idents:
            # identifiers quoted on a line that cites the reference must occur in a file cited on this or a neighbouring
            # line -- or, failing that, somewhere in the reference (the document wraps its lines)
            near = []
            for l2 in text[max(0, ln - 2):ln + 1]:
                for m in CITE.finditer(l2):
                    h = resolve(m.group(1), files)
                    if len(h) == 1:
                        near.append(open(files[h[0]], errors="replace").read())
            for im in IDENT.finditer(line):
                name = im.group(1).strip()
                if OWN.match(name) or CITE.search(name) or "/" in name or " " in name or name.endswith(".md"):
                    continue
                last = re.sub(r"[^\w]", "", re.split(r"::|\.", name.strip("&"))[-1])
                if len(last) < 4 or any(last in b for b in near):
                    continue
                if not any(last in open(pth, errors="replace").read() for pth in files.values()):
                    findings.append("%s:%d: `%s` occurs nowhere in the reference" % (doc, ln, name))
    return n_cites, findings


def main(argv):
    docs = argv or [os.path.join(ROOT, "INTEGRATION.md")]
    if not os.path.isdir(REF):
        print("reference tree absent: nothing to check")
        return 0
    files = ref_files()
    bad = 0
    for d in docs:
        n, findings = check(d, files, os.path.basename(d) == "INTEGRATION.md")
        print("%s: %d citations checked, %d findings" % (os.path.relpath(d, ROOT), n, len(findings)))
        for f in findings:
            print("  " + f)
        bad += len(findings)
    return 1 if bad else 0


if __name__ == "__main__":
    sys.exit(main(sys.argv[1:]))
