#!/usr/bin/env python3
"""One-off extraction of the *numeric constant tables* the reference embeds in C++ source
(substitution-rate matrices and the integer Needleman-Wunsch score tables) into plain
whitespace-separated data files under prographmsa_amd/host/data/.

Only numbers are extracted (published scientific constants: WAG, Whelan & Goldman 2001;
ECM, Kosiol et al. 2007; BLOSUM-derived integer scores); no reference code is copied.
Run in the build container only (the reference tree does not exist on the GPU box).

    python tools/extract_reference_tables.py /root/reference/src

Layout of every output file: first line "<rows> <cols>", then rows*cols numbers in
COLUMN-MAJOR order (element (i,j) at index i + rows*j), exactly the order in which the
reference maps its literal arrays (ModelFactoryWag.cpp:420, ModelFactoryEcm.cpp:3743,
DistanceFactoryAlign.cpp:31,232).
"""
import os
import re
import sys


def arrays(path, ctype):
    txt = open(path).read()
    for m in re.finditer(r"static\s+%s\s+data\[\]\s*=\s*\{([^}]*)\}" % ctype, txt):
        yield [t for t in re.split(r"[\s,]+", m.group(1).strip()) if t]


def emit(out, dim, vals):
    assert len(vals) == dim * dim, (out, len(vals), dim)
    with open(out, "w") as f:
        f.write("%d %d\n" % (dim, dim))
        for j in range(dim):
            f.write(" ".join(vals[j * dim:(j + 1) * dim]) + "\n")
    print("wrote", out, len(vals))


def main(src):
    here = os.path.dirname(os.path.abspath(__file__))
    dst = os.path.join(here, "..", "prographmsa_amd", "host", "data")
    os.makedirs(dst, exist_ok=True)
    emit(os.path.join(dst, "wag.qmat"), 20, next(arrays(os.path.join(src, "ModelFactoryWag.cpp"), "double")))
    emit(os.path.join(dst, "ecm.qmat"), 61, next(arrays(os.path.join(src, "ModelFactoryEcm.cpp"), "double")))
    it = arrays(os.path.join(src, "DistanceFactoryAlign.cpp"), "int")
    emit(os.path.join(dst, "nw_aa.imat"), 21, next(it))
    emit(os.path.join(dst, "nw_codon.imat"), 62, next(it))


if __name__ == "__main__":
    main(sys.argv[1] if len(sys.argv) > 1 else "/root/reference/src")
