#!/usr/bin/env python3
"""Print the fingerprint table of the reference's pack-format kernel sources for
phaneron_amd/csrc/ph_program.cpp (kKnownSources).  Build container only.

A fingerprint is FNV-1a/64 over the kernel text with every whitespace byte removed - a number, not
source text.  `createProgram` (packer.ts:97-103) hands the library the whole format source (both the
`read` and the `write` kernel); an exact fingerprint match identifies the format without guessing.
"""
import glob
import os

HERE = os.path.dirname(os.path.abspath(__file__))
CL = os.path.join(HERE, "..", "_ref", "work", "cl")
FORMATS = ["v210", "yuv422p10", "yuv422p8", "yuv420p", "nv12", "rgba8", "bgra8"]


def fingerprint(text):
    h = 0xCBF29CE484222325
    for b in text.encode():
        if b in b" \t\r\n\f\v":
            continue
        h = ((h ^ b) * 0x100000001B3) & 0xFFFFFFFFFFFFFFFF
    return h


if __name__ == "__main__":
    for f in FORMATS:
        print('    {0x%016xull, PH_FMT_%s},  // src/process/%s.ts' % (fingerprint(open(os.path.join(CL, f + ".cl")).read()),
                                                                  f.upper().replace("YUV422P10", "YUV422P10"), f))
