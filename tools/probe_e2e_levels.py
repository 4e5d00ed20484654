"""Per-level breakdown of the product driver's progressive pass (GPU box): PGM_HOST_PROFILE=1 pgmsa on the headline family."""
import os, subprocess, sys, tempfile
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import gen
import prographmsa_amd as pg
tmp = tempfile.mkdtemp()
fa = os.path.join(tmp, "f.fa"); open(fa, "w").write(gen.fasta(gen.gen(256, 1000, 3)))
VARIANTS = [dict(x.split("=") for x in v.split(",") if x) for v in os.environ.get("PROBE_VARIANTS", "").split(";")]
for rep, var in enumerate(VARIANTS * 2):
    env = dict(os.environ, PGM_HOST_PROFILE="1", **var)
    print("==== variant", var)
    r = subprocess.run([pg.PGMSA_PATH, "--fasta", "-m", "-t", os.path.join(ROOT, "tests/golden/c3.tree"), "--stats", "-o", os.path.join(tmp, "o.fa"), fa],
                       capture_output=True, text=True, env=env)
    print("---- rep", rep, "rc", r.returncode)
    print(r.stderr)
