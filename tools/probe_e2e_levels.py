"""Per-level breakdown of the product driver's progressive pass (GPU box): PGM_HOST_PROFILE=1 pgmsa on the headline family."""
import os, subprocess, sys, tempfile
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import gen
import prographmsa_amd as pg
tmp = tempfile.mkdtemp()
CFG = os.environ.get("PROBE_CFG", "c3")
fam, tree, extra = {"c3": (gen.gen(256, 1000, 3), "c3.tree", ["-m"]), "c5": (gen.gen(1024, 600, 6), "c5.tree", []),
                    "c4": (gen.gen_codon(128, 1000, 4), "c4.tree", ["--codon"]),
                    "default": (gen.gen(256, 1000, 3), None, ["-a"])}[CFG]
fa = os.path.join(tmp, "f.fa"); open(fa, "w").write(gen.fasta(fam))
VARIANTS = [dict(x.split("=") for x in v.split(",") if x) for v in os.environ.get("PROBE_VARIANTS", "").split(";")]
for rep, var in enumerate(VARIANTS * 2):
    env = dict(os.environ, PGM_HOST_PROFILE="1", **var)
    print("==== variant", var)
    import time
    t_wall = time.time()
    r = subprocess.run([pg.PGMSA_PATH, "--fasta"] + extra + (["-t", os.path.join(ROOT, "tests/golden", tree)] if tree else []) + ["--stats", "-o", os.path.join(tmp, "o.fa"), fa],
                       capture_output=True, text=True, env=env)
    print("---- rep", rep, "rc", r.returncode, "wall %.1f ms" % ((time.time() - t_wall) * 1e3))
    print(r.stderr)
