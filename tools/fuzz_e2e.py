"""Differential campaign of the whole product path (GPU box): random families through `pgmsa` (HIP backend: all-pairs farm,
level-batched alignGraphs, node profiles and pair counts on the device) and through the same host mirror on the CPU oracle
(oracle/_build/pgmsa_oracle); FASTA and newick output must be byte-identical.  usage: tools/fuzz_e2e.py SECONDS [SEED]"""
import os, random, subprocess, sys, tempfile, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import gen
import prographmsa_amd as pg
ORACLE = os.path.join(ROOT, "oracle", "_build", "pgmsa_oracle")
K50 = os.path.join(ROOT, "tests", "golden", "K50.lib")
budget = float(sys.argv[1]) if len(sys.argv) > 1 else 60.0
seed0 = int(sys.argv[2]) if len(sys.argv) > 2 else 1
rng = random.Random(seed0)
tmp = tempfile.mkdtemp()
t_end = time.time() + budget
case = bad = 0
kinds = {}
while time.time() < t_end:
    case += 1
    codon = rng.random() < 0.2
    n = rng.choice([2, 3, 5, 8, 13, 24, 40])
    L = rng.choice([20, 60, 150, 300]) if codon else rng.choice([30, 100, 250, 500, 900])
    sub, indel = rng.choice([0.03, 0.08, 0.2]), rng.choice([0.0, 0.01, 0.04])
    seed = rng.randrange(10 ** 6)
    fam = gen.gen_codon(n, L, seed, sub=sub, indel=indel) if codon else gen.gen(n, L, seed, sub=sub, indel=indel)
    fa = os.path.join(tmp, "f.fa"); open(fa, "w").write(gen.fasta(fam))
    if codon:
        flags = rng.choice([["--codon", "--fasta", "-a", "-i", "0"], ["--codon", "--fasta", "-a"], ["--codon", "-T", "-i", "0"], ["--codon", "--fasta", "-a", "--early_refinement", "-i", "0"]])
    else:
        flags = rng.choice([["--fasta", "-a"], ["--fasta", "-a", "-m"], ["--fasta", "-a", "-M", "-i", "0"], ["--fasta", "-a", "-i", "1"],
                            ["--fasta", "-a", "--cs_profile", K50, "-i", "0"], ["-a", "-m", "-T", "-i", "0"],
                            ["--fasta"], ["-T", "-i", "0"],    # (these two: the default flow without -a — k-mer angle distances)
                            ["--fasta", "-a", "--early_refinement", "-i", "0"], ["--fasta", "--early_refinement"]])   # (alignGraphs' second call site: graphs that are not cleaned)
    key = " ".join(f for f in flags if f != K50)
    a = subprocess.run([pg.PGMSA_PATH] + flags + [fa], capture_output=True, text=True)
    b = subprocess.run([ORACLE] + flags + [fa], capture_output=True, text=True)
    ok = a.returncode == 0 and b.returncode == 0 and a.stdout == b.stdout and len(a.stdout) > 0
    kinds[key] = kinds.get(key, 0) + 1
    if not ok:
        bad += 1
        print("case %d DIFFERS: n=%d L=%d sub=%g indel=%g seed=%d codon=%d flags=%s rc=%d/%d %s %s" % (
            case, n, L, sub, indel, seed, codon, key, a.returncode, b.returncode, a.stderr[-200:], b.stderr[-200:]), flush=True)
    if case % 25 == 0:
        print("... %d cases, %d differing" % (case, bad), flush=True)
print("FUZZ-E2E %s: %d cases, %d differing (seed %d); %s" % ("OK" if bad == 0 else "FAILED", case, bad, seed0, kinds))
sys.exit(0 if bad == 0 else 1)
