"""Differential campaign (build container only): random families through the reference binary and through the CPU oracle
driver (oracle/_build/pgmsa_oracle); any FASTA / newick difference is printed.  Used to look for parity gaps beyond the
committed fixtures (this is how the denominator association of c2.m.out.fa was found).  `--angle`: the guide trees from the
k-mer angle distances instead (round 4: 23 of 60 differed before the orientation / depth-block / libm findings, none after).
`--early`: every case with --early_refinement (the second call site of alignGraphs, ProgressiveAlignment.h:170)."""
import os, random, subprocess, sys, tempfile
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tests"))
import gen
REF = "/root/reference/bin/ProGraphMSA_64"
OUR = os.path.join(ROOT, "oracle", "_build", "pgmsa_oracle")
ANGLE = "--angle" in sys.argv   # the initial guide tree from the sequences alone (k-mer angle distances, BioNJ, midpoint root): -T -i 0
if ANGLE:
    sys.argv.remove("--angle")
EARLY = "--early" in sys.argv
if EARLY:
    sys.argv.remove("--early")
tmp = tempfile.mkdtemp()
rng = random.Random(int(sys.argv[1]) if len(sys.argv) > 1 else 1)
ncases = int(sys.argv[2]) if len(sys.argv) > 2 else 40
bad = 0
for case in range(ncases):
    codon = rng.random() < float(os.environ.get("CAMPAIGN_CODON", "0.25"))
    n = rng.choice([4, 6, 9, 12, 16, 24, 32, 48])
    L = rng.choice([40, 80, 150, 250, 400]) if not codon else rng.choice([30, 60, 120, 200])
    sub, indel = rng.choice([0.03, 0.06, 0.1, 0.2]), rng.choice([0.003, 0.01, 0.03])
    seed = rng.randrange(10 ** 6)
    fam = gen.gen_codon(n, L, seed, sub=sub, indel=indel) if codon else gen.gen(n, L, seed, sub=sub, indel=indel)
    fa = os.path.join(tmp, "f.fa"); open(fa, "w").write(gen.fasta(fam))
    base = ["--codon"] if codon else []
    if ANGLE:
        n = rng.choice([4, 5, 6, 7, 8, 9, 10, 12, 13, 15, 16, 17, 20, 24, 31, 40])
        fam = gen.gen_codon(n, L, seed, sub=sub, indel=indel) if codon else gen.gen(n, L, seed, sub=sub, indel=indel)
        open(fa, "w").write(gen.fasta(fam))
        a = subprocess.run([REF] + base + ["-T", "-i", "0", fa], capture_output=True, text=True)
        b = subprocess.run([OUR] + base + ["-T", "-i", "0", fa], capture_output=True, text=True)
        if not (a.returncode == 0 and b.returncode == 0 and a.stdout == b.stdout):
            bad += 1
            print("case %d DIFFERS: n=%d L=%d sub=%g indel=%g seed=%d codon=%d" % (case, n, L, sub, indel, seed, codon), flush=True)
        continue
    flags = base + rng.choice([[], ["-m"], ["-m"], ["-M"]] if not codon else [[], ["-m"]]) + (["--early_refinement"] if EARLY else [])
    tr = subprocess.run([REF] + base + ["-T", "-i", "0", fa], capture_output=True, text=True)
    if tr.returncode:
        print("case %d: reference tree failed" % case); continue
    tree = os.path.join(tmp, "t.tree"); open(tree, "w").write(tr.stdout)
    a = subprocess.run([REF, "--fasta"] + flags + ["--tree", tree, fa], capture_output=True, text=True)
    b = subprocess.run([OUR, "--fasta"] + flags + ["--tree", tree, fa], capture_output=True, text=True)
    ok = a.returncode == 0 and b.returncode == 0 and a.stdout == b.stdout
    if not ok:
        bad += 1
        keep = os.path.join(tmp, "bad%d" % case); os.makedirs(keep)
        open(os.path.join(keep, "f.fa"), "w").write(gen.fasta(fam)); open(os.path.join(keep, "t.tree"), "w").write(tr.stdout)
        print("case %d DIFFERS: n=%d L=%d sub=%g indel=%g seed=%d flags=%s rc=%d/%d kept in %s %s" % (case, n, L, sub, indel, seed, flags, a.returncode, b.returncode, keep, b.stderr[:100]), flush=True)
print("campaign seed %s: %d cases, %d differing" % (sys.argv[1] if len(sys.argv) > 1 else "1", ncases, bad))
