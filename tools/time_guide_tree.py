"""Wall time of the product driver's all-pairs guide tree for the 256 x 1000 family (GPU box): pgmsa -m -a -T -i 0."""
import hashlib, json, os, subprocess, sys, tempfile, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import gen
import prographmsa_amd as pg
tmp = tempfile.mkdtemp()
fa = os.path.join(tmp, "c3.fa"); open(fa, "w").write(gen.fasta(gen.gen(256, 1000, 3)))
want = json.load(open(os.path.join(ROOT, "tests/golden/md5.json")))["c3.nw_ml.tree"]
for k in range(3):
    t0 = time.perf_counter()
    out = subprocess.run([pg.PGMSA_PATH, "-m", "-a", "-T", "-i", "0", "--stats", fa], check=True, capture_output=True, text=True)
    dt = time.perf_counter() - t0
    print("run %d: %.3f s wall, newick md5 %s; %s" % (k, dt, "matches the reference" if hashlib.md5(out.stdout.encode()).hexdigest() == want else "DIFFERS",
          " ".join(l for l in out.stderr.splitlines() if "nw" in l.lower() or "mldist" in l.lower())[:300]), flush=True)
