"""Deterministic synthetic sequence-family generators.

Pure python `random` so the same text is produced here and on the GPU box.
Usage: gen.py N L SEED [codon]   -> FASTA on stdout, names seq%04d
"""
import random
import sys

AA = "ACDEFGHIKLMNPQRSTVWY"
# 61 sense codons in TCAG order (stop codons TAA, TAG, TGA removed)
CODONS = [a + b + c for a in "TCAG" for b in "TCAG" for c in "TCAG"
          if a + b + c not in ("TAA", "TAG", "TGA")]


def gen(n, L, seed, sub=0.08, indel=0.01, alphabet=AA):
    rng = random.Random(seed)
    root = [rng.choice(alphabet) for _ in range(L)]
    seqs = [root]
    while len(seqs) < n:
        new = []
        for s in seqs:
            for _ in range(2):
                t = []
                for c in s:
                    r = rng.random()
                    if r < indel / 2:
                        continue                      # deletion
                    if r < indel:                     # insertion after c
                        t.append(c)
                        t.append(rng.choice(alphabet))
                        continue
                    t.append(rng.choice(alphabet) if rng.random() < sub else c)
                new.append(t)
        seqs = new
    return ["".join(s) for s in seqs[:n]]


def gen_codon(n, L, seed, sub=0.08, indel=0.01):
    alphabet = [chr(0x100 + i) for i in range(61)]
    seqs = gen(n, L, seed, sub, indel, alphabet)
    return ["".join(CODONS[ord(c) - 0x100] for c in s) for s in seqs]


def fasta(seqs):
    return "".join(">seq%04d\n%s\n" % (i, s) for i, s in enumerate(seqs))


def genlib(K, seed, ncols=13):
    """Synthetic context-profile library in the K4000.lib text format (Appendix D)."""
    import math
    rng = random.Random(seed)
    letters = "ARNDCQEGHILKMFPSTWYV"
    out = ["ProfileLibrary", "NPROF\t%d" % K, "NCOLS\t%d" % ncols, "ITERS\t0", "LOG\t1"]
    for k in range(K):
        out += ["ContextProfile", "INDEX\t%d" % k, "PRIOR\t%r" % (1.0 / K),
                "NCOLS\t%d" % ncols, "ALPH\t20", "LOG\t1", "\t" + "\t".join(letters)]
        for col in range(1, ncols + 1):
            p = [rng.gammavariate(0.3, 1.0) + 1e-4 for _ in range(20)]
            s = sum(p)
            out.append("%d\t" % col + "\t".join(str(int(round(-1000 * math.log2(x / s)))) for x in p))
        out.append("//")
    return "\n".join(out) + "\n"


if __name__ == "__main__":
    n, L, seed = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3])
    if len(sys.argv) > 4 and sys.argv[4] == "codon":
        sys.stdout.write(fasta(gen_codon(n, L, seed)))
    else:
        sys.stdout.write(fasta(gen(n, L, seed)))


def gen_repeat_family(n, L, seed, unit_len=7, sub=0.08, indel=0.01, annotate_every=1):
    """A family (gen) whose members each carry a tandem-repeat region — 2-5 copies of a family-wide unit, with substitutions and
    deletions inside the copies — and the T-REKS report that annotates those regions (the format `--read_repeats` parses,
    reference src/RepeatDetectionTReks.cpp:62-151): per sequence a '>' line, a header line 'Length: ... from S to E ...' (S
    1-based, in the sequence after start stripping) and the aligned units, one per line ('-' = gap), up to a line of asterisks.
    Returns (sequences, report text)."""
    rng = random.Random(seed * 7919 + 13)
    alphabet = "ACDEFGHIKLMNPQRSTVWY"
    seqs = gen(n, L, seed, sub, indel)
    unit = [rng.choice(alphabet) for _ in range(unit_len)]
    out, trd = [], []
    for i, s in enumerate(seqs):
        if s[0] == "M":
            s = "A" + s[1:]   # (a leading M would be stripped before the repeats are mapped: keep the coordinates simple)
        units = []
        for _ in range(rng.randint(2, 5)):
            u = []
            for ch in unit:
                r = rng.random()
                u.append("-" if r < 0.08 else (rng.choice(alphabet) if r < 0.18 else ch))
            if all(x == "-" for x in u):
                u[0] = unit[0]
            units.append("".join(u))
        pos = rng.randint(5, len(s) - 5)
        region = "".join(u.replace("-", "") for u in units)
        out.append(s[:pos] + region + s[pos:])
        if i % annotate_every == 0:
            trd.append(">seq%04d\nLength: %d residues - nb: %d  from  %d to %d - Psim:0.80 region Length:%d\n%s\n**********************\n\n"
                       % (i, unit_len, len(units), pos + 1, pos + len(region), len(region), "\n".join(units)))
    return out, "".join(trd)
