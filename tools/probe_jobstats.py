"""How the nodes of every job of a config are served (GPU box): PGM_JOB_STATS lines of the C API for PROBE_CFG = c3 | c4 | c5."""
import os, subprocess, sys, tempfile
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import gen
import prographmsa_amd as pg
tmp = tempfile.mkdtemp()
cfg = os.environ.get("PROBE_CFG", "c3")
fam, flags = {"c3": (lambda: gen.gen(256, 1000, 3), ["--fasta", "-m", "-t", os.path.join(ROOT, "tests/golden/c3.tree")]),
              "c4": (lambda: gen.gen_codon(128, 1000, 4), ["--codon", "--fasta", "-t", os.path.join(ROOT, "tests/golden/c4.tree")]),
              "c5": (lambda: gen.gen(1024, 600, 6), ["--fasta", "-t", os.path.join(ROOT, "tests/golden/c5.tree")])}[cfg]
fa = os.path.join(tmp, "f.fa"); open(fa, "w").write(gen.fasta(fam()))
env = dict(os.environ, PGM_JOB_STATS="1")
r = subprocess.run([pg.PGMSA_PATH] + flags + ["-o", os.path.join(tmp, "o.fa"), fa], env=env, capture_output=True, text=True)
sys.stdout.write(r.stderr)
