import os, subprocess, sys, tempfile
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import gen
import prographmsa_amd as pg
out = sys.argv[1]
tmp = tempfile.mkdtemp()
fa = os.path.join(tmp, "f.fa"); open(fa, "w").write(gen.fasta(gen.gen(256, 1000, 3)))
subprocess.run([pg.PGMSA_PATH, "--fasta", "-m", "-t", os.path.join(ROOT, "tests/golden/c3.tree"), "--dump_jobs", out, "-o", os.path.join(tmp, "o.fa"), fa], check=True)
print("dumped", out)
