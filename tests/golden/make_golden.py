#!/usr/bin/env python3
"""Regenerates the golden fixtures from the reference's prebuilt binary (build container only).

    python tests/golden/make_golden.py [/root/reference/bin/ProGraphMSA_64]

The reference cannot be compiled here (Eigen/TCLAP absent) and has no test fixtures of its own, so
every golden vector is an output of the reference binary (rev 5e7b708) on deterministic synthetic
inputs from tests/gen.py.  The binary never travels to the GPU box; only these small text files do.
"""
import hashlib
import json
import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))
import gen  # noqa: E402

ONLY = None
if "--only" in sys.argv:   # regenerate one section only (nw_trees | full_size | ancestral | repeats | angle | early)
    i = sys.argv.index("--only")
    ONLY = sys.argv[i + 1]
    del sys.argv[i:i + 2]
BIN = sys.argv[1] if len(sys.argv) > 1 else "/root/reference/bin/ProGraphMSA_64"


def run(args):
    return subprocess.run([BIN] + args, check=True, capture_output=True, text=True).stdout


def w(name, text):
    with open(os.path.join(HERE, name), "w") as f:
        f.write(text)


NW_TREE_CASES = [  # (taxa, length, seed, substitution rate, indel rate): odd and even sizes, see nw_trees()
    (5, 250, 585126, 0.05, 0.005), (6, 250, 81768, 0.05, 0.005), (7, 120, 59769, 0.1, 0.005), (9, 60, 815905, 0.2, 0.005),
    (9, 250, 4816, 0.1, 0.02), (11, 120, 348741, 0.05, 0.02), (13, 120, 133400, 0.2, 0.02), (16, 120, 653397, 0.2, 0.005),
    (16, 60, 944662, 0.2, 0.005), (16, 250, 545337, 0.1, 0.02), (21, 120, 295589, 0.05, 0.02), (30, 60, 7, 0.1, 0.02)]


def nw_trees():
    """BioNJ guide trees from alignment p-distances (-a -T -i 0).  Every BioNJ run ends in the exact 4-cluster tie
    Q(0,1) = Q(2,3) that the last bit of Eigen's vectorised column sums decides (TreeNJ.cpp:157), and for odd matrix sizes
    the sums depend on the 16-byte alignment of each column, so these trees pin that arithmetic.  (With -m the tie is
    decided by the last bits of the ML distances, i.e. of Eigen's EigenSolver output: not reproducible, see DESIGN.md.)"""
    out = []
    for (n, L, seed, sub, indel) in NW_TREE_CASES:
        w("nwt.fa.tmp", gen.fasta(gen.gen(n, L, seed, sub=sub, indel=indel)))
        out.append(dict(n=n, L=L, seed=seed, sub=sub, indel=indel, tree=run(["-a", "-T", "-i", "0", "nwt.fa.tmp"])))
    os.remove("nwt.fa.tmp")
    w("nw_trees.json", json.dumps(out, indent=0))


def full_size():
    """Full-size BASELINE configs 4 and 5, one progressive pass on a committed guide tree, md5 only (the inputs are
    regenerated from the seeds by tests/gen.py): 128 x 1000 codons (--codon) and 1024 x 600 aa with the synthetic
    K = 4000 context-profile library (gen.genlib(4000, 11), ~5 MB of text, not committed).  The reference needs ~10 s
    and ~3-8 min for these two lines."""
    md5 = json.load(open("md5.json"))
    fa = gen.fasta(gen.gen_codon(128, 1000, 4))
    w("c4.fa.tmp", fa)
    w("c4.tree", run(["--codon", "-T", "-i", "0", "c4.fa.tmp"]))
    md5["c4.fa"] = hashlib.md5(fa.encode()).hexdigest()
    md5["c4.out.fa"] = hashlib.md5(run(["--codon", "--fasta", "-t", "c4.tree", "c4.fa.tmp"]).encode()).hexdigest()
    os.remove("c4.fa.tmp")
    fa = gen.fasta(gen.gen(1024, 600, 6))
    w("c5.fa.tmp", fa)
    lib = gen.genlib(4000, 11)
    w("K4000syn.lib.tmp", lib)
    w("c5.tree", run(["-T", "-i", "0", "c5.fa.tmp"]))
    md5["c5.fa"] = hashlib.md5(fa.encode()).hexdigest()
    md5["K4000syn.lib"] = hashlib.md5(lib.encode()).hexdigest()
    md5["c5.cs.out.fa"] = hashlib.md5(run(["--fasta", "-t", "c5.tree", "--cs_profile", "K4000syn.lib.tmp", "c5.fa.tmp"]).encode()).hexdigest()
    md5["c5.out.fa"] = hashlib.md5(run(["--fasta", "-t", "c5.tree", "c5.fa.tmp"]).encode()).hexdigest()
    os.remove("c5.fa.tmp")
    os.remove("K4000syn.lib.tmp")
    w("md5.json", json.dumps(md5, indent=1))


def ancestral():
    """Merge numerics (SURVEY 8c probe list): --ancestral_seqs --profile_out writes, per ancestor, the merged graph's node
    profiles (.* pi, normalised to sum 1) at 6 significant digits and adds the ancestral rows to the FASTA.  A 3 x 30 family in
    full text, larger ones (8 x 120, 64 x 400, 6 x 60 codons) by md5."""
    md5 = json.load(open("md5.json"))
    w("a1.fa", gen.fasta(gen.gen(3, 30, 31, sub=0.15, indel=0.03)))
    w("a1.tree", run(["-T", "-i", "0", "a1.fa"]))
    w("a1.anc.out.fa", run(["--fasta", "--ancestral_seqs", "--profile_out", "a1.anc.prof", "-t", "a1.tree", "a1.fa"]))
    for name, flags in (("c1", []), ("c2", []), ("cd1", ["--codon"])):
        out = run(flags + ["--fasta", "--ancestral_seqs", "--profile_out", "anc.prof.tmp", "-t", name + ".tree", name + ".fa"])
        md5[name + ".anc.out.fa"] = hashlib.md5(out.encode()).hexdigest()
        if not flags:   # (codons: P(d) of the 61-state ECM model differs from Eigen's EigenSolver in the last bits, DESIGN section 1: the
            md5[name + ".anc.prof"] = hashlib.md5(open("anc.prof.tmp", "rb").read()).hexdigest()   # 6-digit text is not reproducible, the FASTA is)
    os.remove("anc.prof.tmp")
    w("md5.json", json.dumps(md5, indent=1))


REPEAT_CASES = [(4, 70, 3, 1), (8, 90, 4, 1), (12, 80, 9, 1), (16, 100, 10, 2), (8, 60, 22, 1), (24, 150, 23, 1)]   # (taxa, length, seed, annotate every k-th sequence)


def repeats():
    """Tandem-repeat edges (Graph::addRepeats, PredIterator's repeat arm, markAlternativePath, n_tr_indels): families whose
    members carry an annotated repeat region (gen.gen_repeat_family writes the T-REKS report), aligned with
    `--fasta -R --read_repeats f.trd -t tree` (no Java needed: the report is read, not produced).  Kept: the guide tree, the
    FASTA, the "TR indels" lines of stderr (per internal node and in total), and for two families the default flow
    (`--fasta -a --read_repeats`)."""
    out = []
    for (n, L, seed, ae) in REPEAT_CASES:
        seqs, trd = gen.gen_repeat_family(n, L, seed, annotate_every=ae)
        w("rep.fa.tmp", gen.fasta(seqs))
        w("rep.trd.tmp", trd)
        tree = run(["-T", "-i", "0", "rep.fa.tmp"])
        w("rep.tree.tmp", tree)
        r = subprocess.run([BIN, "--fasta", "-R", "--read_repeats", "rep.trd.tmp", "-t", "rep.tree.tmp", "rep.fa.tmp"], check=True, capture_output=True, text=True)
        plain = run(["--fasta", "-t", "rep.tree.tmp", "rep.fa.tmp"])
        rec = dict(n=n, L=L, seed=seed, annotate_every=ae, tree=tree, out=r.stdout, tr_lines=[ln for ln in r.stderr.splitlines() if "TR indels" in ln],
                   changes_alignment=(r.stdout != plain))
        if n <= 8:
            rec["out_default_flow"] = run(["--fasta", "-a", "--read_repeats", "rep.trd.tmp", "rep.fa.tmp"])
        out.append(rec)
    for f in ("rep.fa.tmp", "rep.trd.tmp", "rep.tree.tmp"):
        os.remove(f)
    w("repeats.json", json.dumps(out, indent=0))


EARLY_CASES = [   # (name, input, extra arguments in front of --early_refinement; the tree is <input>.tree unless the flow builds its own)
    ("c1", "c1", ["--fasta", "--tree", "c1.tree"]), ("c1_m", "c1", ["--fasta", "-m", "--tree", "c1.tree"]),
    ("m1", "m1", ["--fasta", "--tree", "m1.tree"]), ("x1", "x1", ["--fasta", "--tree", "x1.tree"]),
    ("c1_cs", "c1", ["--fasta", "--tree", "c1.tree", "--cs_profile", "K50.lib"]),
    ("c1_default_flow", "c1", ["--fasta"]), ("c1_a_flow", "c1", ["--fasta", "-a"]),
    ("cd1", "cd1", ["--codon", "--fasta", "-t", "cd1.tree"]),
    ("c2_m", "c2", ["--fasta", "-m", "--tree", "c2.tree"]),
]


def early():
    """--early_refinement (ProgressiveAlignment.h:102-243): after every internal node with a grandchild the node's graph is
    aligned again with the graphs of its grandchildren — the second call site of alignGraphs (:170), on graphs that are not
    cleaned — and rebuilt with mergeGraphsIncremental.  Kept: the FASTA of the committed families with the flag (64 x 400 as md5),
    whether the flag changes the alignment, and the repeat families with -R --read_repeats --early_refinement."""
    out = {}
    for name, inp, args in EARLY_CASES:
        fa = run(args + ["--early_refinement", inp + ".fa"])
        plain = run(args + [inp + ".fa"])
        rec = dict(input=inp + ".fa", args=args, changes_alignment=(fa != plain))
        if len(fa) > 20000: rec["md5"] = hashlib.md5(fa.encode()).hexdigest()
        else: rec["out"] = fa
        out[name] = rec
    reps = []
    for (n, L, seed, ae) in REPEAT_CASES[:4]:
        seqs, trd = gen.gen_repeat_family(n, L, seed, annotate_every=ae)
        w("rep.fa.tmp", gen.fasta(seqs))
        w("rep.trd.tmp", trd)
        tree = run(["-T", "-i", "0", "rep.fa.tmp"])
        w("rep.tree.tmp", tree)
        r = subprocess.run([BIN, "--fasta", "-R", "--read_repeats", "rep.trd.tmp", "--early_refinement", "-t", "rep.tree.tmp", "rep.fa.tmp"], check=True, capture_output=True, text=True)
        reps.append(dict(n=n, L=L, seed=seed, annotate_every=ae, tree=tree, out=r.stdout, tr_lines=[ln for ln in r.stderr.splitlines() if "TR indels" in ln]))
    for f in ("rep.fa.tmp", "rep.trd.tmp", "rep.tree.tmp"):
        os.remove(f)
    out["repeats"] = reps
    w("early.json", json.dumps(out, indent=0))


def angle():
    """The reference's flow WITHOUT -a: initial guide tree from the k-mer angle distances (DistanceFactoryAngle.h:55-131), then the
    usual two rounds of alignment + tree re-estimation and the final alignment.  BASELINE config 2 as worded (`--fasta c2.fa`),
    config 3's progressive part (`--fasta --mldist c3.fa`, ~30 s), the small families in text; and the initial trees alone
    (-T -i 0) of the twelve NW_TREE_CASES families: every BioNJ run ends in an exact tie that the last bits of the distances
    decide, and the cosine matrix comes out of Eigen's GEMM, whose summation order the sources do not show — the restatement
    reproduces about two trees in three (tests/test_oracle_golden.py lists which), the FASTA of the full flow every time."""
    md5 = json.load(open("md5.json"))
    w("c1.default.out.fa", run(["--fasta", "c1.fa"]))
    w("cd1.default.out.fa", run(["--codon", "--fasta", "cd1.fa"]))
    md5["c2.default.out.fa"] = hashlib.md5(run(["--fasta", "c2.fa"]).encode()).hexdigest()
    w("c3.fa.tmp", gen.fasta(gen.gen(256, 1000, 3)))
    md5["c3.default_m.out.fa"] = hashlib.md5(run(["--fasta", "--mldist", "c3.fa.tmp"]).encode()).hexdigest()
    os.remove("c3.fa.tmp")
    out = []
    for (n, L, seed, sub, indel) in NW_TREE_CASES:
        w("ang.fa.tmp", gen.fasta(gen.gen(n, L, seed, sub=sub, indel=indel)))
        out.append(dict(n=n, L=L, seed=seed, sub=sub, indel=indel, tree=run(["-T", "-i", "0", "ang.fa.tmp"]), fasta=run(["--fasta", "ang.fa.tmp"])))
    os.remove("ang.fa.tmp")
    w("angle_trees.json", json.dumps(out, indent=0))
    w("md5.json", json.dumps(md5, indent=1))


def main():
    os.chdir(HERE)
    if ONLY == "angle":
        return angle()
    if ONLY == "repeats":
        return repeats()
    if ONLY == "early":
        return early()
    if ONLY == "ancestral":
        return ancestral()
    if ONLY == "nw_trees":
        return nw_trees()
    if ONLY == "full_size":
        return full_size()
    md5 = {}
    # c1: 8 x 120 aa (BASELINE config 1), guide tree from the reference (-T), then --fasta --tree
    w("c1.fa", gen.fasta(gen.gen(8, 120, 1)))
    w("c1.tree", run(["-T", "c1.fa"]))
    w("c1.out.fa", run(["--fasta", "--tree", "c1.tree", "c1.fa"]))
    # c2: 64 x 400 aa (config 2)
    w("c2.fa", gen.fasta(gen.gen(64, 400, 2)))
    w("c2.tree", run(["-T", "-i", "0", "c2.fa"]))
    w("c2.out.fa", run(["--fasta", "--tree", "c2.tree", "c2.fa"]))
    # c3: 256 x 1000 aa (config 3, --mldist): the input is regenerated from the seed, tree committed,
    # alignment pinned by md5 (838 KB of FASTA is not committed)
    w("c3.fa.tmp", gen.fasta(gen.gen(256, 1000, 3)))
    w("c3.tree", run(["-m", "-T", "c3.fa.tmp"]))
    out = run(["--fasta", "-m", "-t", "c3.tree", "c3.fa.tmp"])
    md5["c3.out.fa"] = hashlib.md5(out.encode()).hexdigest()
    # the config's first stage: all 32 640 alignPair calls + ML distances + BioNJ (the reference needs ~15 min for this line)
    md5["c3.nw_ml.tree"] = hashlib.md5(run(["-m", "-a", "-T", "-i", "0", "c3.fa.tmp"]).encode()).hexdigest()
    # the reference's default flow from sequences alone on the headline family (another ~15 min: all-pairs tree, two rounds
    # of alignment + tree re-estimation, final alignment)
    md5["c3.a_iter.out.fa"] = hashlib.md5(run(["--fasta", "-a", "c3.fa.tmp"]).encode()).hexdigest()
    md5["c3.fa"] = hashlib.md5(open("c3.fa.tmp", "rb").read()).hexdigest()
    # sequences starting with M exercise the start-stripping path (main.cpp:332-353)
    seqs = gen.gen(6, 90, 11, sub=0.12, indel=0.02)
    seqs = ["M" + s if i % 2 == 0 else s for i, s in enumerate(seqs)]
    w("m1.fa", gen.fasta(seqs))
    w("m1.tree", run(["-T", "-i", "0", "m1.fa"]))
    w("m1.out.fa", run(["--fasta", "--tree", "m1.tree", "m1.fa"]))
    # invalid residues (X / B) -> uniform profile columns
    seqs = gen.gen(4, 70, 12, sub=0.1, indel=0.02)
    seqs = [s[:10] + "X" + s[11:30] + "B" + s[31:] for s in seqs]
    w("x1.fa", gen.fasta(seqs))
    w("x1.tree", run(["-T", "-i", "0", "x1.fa"]))
    w("x1.out.fa", run(["--fasta", "--tree", "x1.tree", "x1.fa"]))
    # pairwise alignGraphs probes: 2 sequences + fixed 2-leaf tree (the two rows ARE mapping1/mapping2)
    pairs = {}
    import random
    for seed in range(100, 124):
        s = gen.gen(2, 80, seed, sub=0.15, indel=0.03)
        d = random.Random(seed).uniform(0.03, 0.4)
        fa = ">a\n%s\n>b\n%s\n" % (s[0], s[1])
        tree = "(a:%g,b:%g);\n" % (d, d * 0.7)
        w("pair.fa.tmp", fa)
        w("pair.tree.tmp", tree)
        pairs[str(seed)] = dict(fasta=fa, tree=tree, out=run(["--fasta", "--tree", "pair.tree.tmp", "pair.fa.tmp"]),
                                out_mldist=run(["--fasta", "-m", "--tree", "pair.tree.tmp", "pair.fa.tmp"]))
    w("pairs.json", json.dumps(pairs, indent=0))
    # alignPair + distance estimation: -a [-m] -T -i 0 -> newick at 6 s.f. (TreeNJ.cpp:259-262)
    nw = {}
    for seed in range(200, 212):
        s = gen.gen(2, 150, seed, sub=0.2, indel=0.04)
        fa = ">a\n%s\n>b\n%s\n" % (s[0], s[1])
        w("pair.fa.tmp", fa)
        nw[str(seed)] = dict(fasta=fa, ml=run(["-a", "-m", "-T", "-i", "0", "pair.fa.tmp"]),
                             pdist=run(["-a", "-T", "-i", "0", "pair.fa.tmp"]))
    w("nw_pairs.json", json.dumps(nw, indent=0))
    # all-pairs NW guide tree of the 8x120 family (BioNJ + midpoint root), with and without ML distances
    w("c1.nw_ml.tree", run(["-a", "-m", "-T", "-i", "0", "c1.fa"]))
    w("c1.nw_p.tree", run(["-a", "-T", "-i", "0", "c1.fa"]))
    w("c2.nw_ml.tree", run(["-a", "-m", "-T", "-i", "0", "c2.fa"]))
    # the reference's default flow from sequences alone: all-pairs guide tree, then two rounds of progressive alignment +
    # guide-tree re-estimation from the induced distances (DistanceFactoryPrealigned), then the final alignment
    w("c1.a_iter.out.fa", run(["--fasta", "-a", "c1.fa"]))
    w("c1.a_iter.tree", run(["-a", "-T", "c1.fa"]))
    nw_trees()
    # context-specific profiles: small synthetic library (K=50), 8x120 family
    w("K50.lib", gen.genlib(50, 7))
    w("c1.cs.out.fa", run(["--fasta", "--tree", "c1.tree", "--cs_profile", "K50.lib", "c1.fa"]))
    w("c1.cs_ml.out.fa", run(["--fasta", "-m", "--tree", "c1.tree", "--cs_profile", "K50.lib", "c1.fa"]))
    # codon mode (61-state ECM model, BASELINE config 4 in small): in-frame sense codons
    w("cd1.fa", gen.fasta(gen.gen_codon(6, 60, 21, sub=0.06, indel=0.01)))
    w("cd1.tree", run(["--codon", "-T", "-i", "0", "cd1.fa"]))
    w("cd1.out.fa", run(["--codon", "--fasta", "-t", "cd1.tree", "cd1.fa"]))
    w("cd2.fa", gen.fasta(gen.gen_codon(16, 150, 22, sub=0.05, indel=0.008)))
    w("cd2.tree", run(["--codon", "-T", "-i", "0", "cd2.fa"]))
    w("cd2.out.fa", run(["--codon", "--fasta", "-t", "cd2.tree", "cd2.fa"]))
    # the 64 x 400 family with context-specific profiles and with --mldist, md5 only
    md5["c2.cs.out.fa"] = hashlib.md5(run(["--fasta", "--tree", "c2.tree", "--cs_profile", "K50.lib", "c2.fa"]).encode()).hexdigest()
    md5["c2.m.out.fa"] = hashlib.md5(run(["--fasta", "-m", "--tree", "c2.tree", "c2.fa"]).encode()).hexdigest()
    md5["c2.a_iter.out.fa"] = hashlib.md5(run(["--fasta", "-a", "c2.fa"]).encode()).hexdigest()
    md5["c2.a_iter.tree"] = hashlib.md5(run(["-a", "-T", "c2.fa"]).encode()).hexdigest()
    # larger codon families, md5 only (the FASTA is regenerated by tests/gen.py): 40 x 330 and 64 x 500 codons
    for name, fam in (("cd3", gen.gen_codon(40, 330, 23, sub=0.05, indel=0.008)), ("cd4", gen.gen_codon(64, 500, 24, sub=0.04, indel=0.005))):
        fa = gen.fasta(fam)
        w(name + ".fa.tmp", fa)
        w(name + ".tree", run(["--codon", "-T", "-i", "0", name + ".fa.tmp"]))
        md5[name + ".fa"] = hashlib.md5(fa.encode()).hexdigest()
        md5[name + ".out.fa"] = hashlib.md5(run(["--codon", "--fasta", "-t", name + ".tree", name + ".fa.tmp"]).encode()).hexdigest()
        os.remove(name + ".fa.tmp")
    # codon alignPair + ML distance, 2 sequences each (a 6-taxon BioNJ tree would end in the exact 4-taxon NJ tie again)
    cnw = {}
    for seed in range(300, 306):
        s = gen.gen_codon(2, 70, seed, sub=0.1, indel=0.02)
        fa = ">a\n%s\n>b\n%s\n" % (s[0], s[1])
        w("pair.fa.tmp", fa)
        cnw[str(seed)] = dict(fasta=fa, ml=run(["--codon", "-a", "-m", "-T", "-i", "0", "pair.fa.tmp"]))
    w("nw_pairs_codon.json", json.dumps(cnw, indent=0))
    w("md5.json", json.dumps(md5, indent=1))
    full_size()
    ancestral()
    repeats()
    angle()
    early()
    for f in ("pair.fa.tmp", "pair.tree.tmp"):
        os.remove(f)
    print("golden fixtures regenerated in", HERE)


if __name__ == "__main__":
    main()
