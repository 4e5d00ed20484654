"""Condenses a tools/profile_bench.sh run into the small files kept under profiles/:
  <tag>_kernel_stats.csv  (rocprofv3 --kernel-trace --stats summary, our kernels)
  <tag>_pmc.json          (per-launch FETCH_SIZE / WRITE_SIZE of each kernel, KB as rocprofv3 reports them, and the SQ counters, summed over the dispatch)"""
import csv, glob, json, os, sys
out, tag = sys.argv[1], sys.argv[2]
dst = os.path.join(out, "summary"); os.makedirs(dst, exist_ok=True)
stats = glob.glob(os.path.join(out, "stats", "**", "*kernel_stats.csv"), recursive=True)
if stats:
    rows = list(csv.reader(open(stats[0])))
    with open(os.path.join(dst, "%s_kernel_stats.csv" % tag), "w", newline="") as f:
        w = csv.writer(f); w.writerow(rows[0])
        for r in rows[1:]:
            if "pgm_" in r[0]:
                w.writerow(r)
pmc = {}
for name in ("fetch", "write", "sq"):
    for path in glob.glob(os.path.join(out, name, "**", "*counter_collection.csv"), recursive=True):
        for r in csv.DictReader(open(path)):
            k = r["Kernel_Name"].split("(")[0].replace("void ", "")
            if "pgm_" not in k:
                continue
            d = pmc.setdefault(k, {}).setdefault(r["Counter_Name"], [])
            d.append(float(r["Counter_Value"]))
summary = {k: {c: {"launches": len(v), ("mean_kb" if c.endswith("_SIZE") else "mean"): sum(v) / len(v)} for c, v in cs.items()} for k, cs in pmc.items()}
json.dump(summary, open(os.path.join(dst, "%s_pmc.json" % tag), "w"), indent=1, sort_keys=True)
print(json.dumps(summary, indent=1, sort_keys=True))
