"""Condenses a tools/profile_bench.sh run into the small files kept under profiles/:
  <tag>_kernel_stats.csv  (rocprofv3 --kernel-trace --stats summary, our kernels)
  <tag>_pmc.json          (per-launch FETCH_SIZE / WRITE_SIZE of each kernel, KB as rocprofv3 reports them, and the SQ counters, summed over the dispatch)"""
import csv, glob, json, os, sys
out, tag = sys.argv[1], sys.argv[2]
dst = os.path.join(out, "summary"); os.makedirs(dst, exist_ok=True)
stats = sorted(glob.glob(os.path.join(out, "stats", "**", "*kernel_stats.csv"), recursive=True), key=os.path.getsize)
if stats:
    rows = list(csv.reader(open(stats[-1])))
    with open(os.path.join(dst, "%s_kernel_stats.csv" % tag), "w", newline="") as f:
        w = csv.writer(f); w.writerow(rows[0])
        for r in rows[1:]:
            if "pgm_" in r[0]:
                w.writerow(r)
pmc = {}
for name in ("fetch", "write", "sq"):
    # one file per process: bench.py itself and the product driver it runs once to capture the jobs; keep bench.py's (the largest)
    paths = sorted(glob.glob(os.path.join(out, name, "**", "*counter_collection.csv"), recursive=True), key=os.path.getsize)
    for path in paths[-1:]:
        for r in csv.DictReader(open(path)):
            k = r["Kernel_Name"].split("(")[0].replace("void ", "")
            if "pgm_" not in k:
                continue
            d = pmc.setdefault(k, {}).setdefault(r["Counter_Name"], [])
            d.append(float(r["Counter_Value"]))
summary = {k: {c: ({"launches": len(v), "mean_kb": sum(v) / len(v), "total_kb": sum(v)} if c.endswith("_SIZE") else {"launches": len(v), "mean": sum(v) / len(v)}) for c, v in cs.items()} for k, cs in pmc.items()}
# steps of the profiled run (bench.py --only-headline: every launch belongs to a step; one emission kernel per step)
em = next((v for k, v in summary.items() if k.startswith("pgm_emission_skew_kernel")), {})
summary["_steps"] = em.get("FETCH_SIZE", em.get("WRITE_SIZE", {})).get("launches", 0)
json.dump(summary, open(os.path.join(dst, "%s_pmc.json" % tag), "w"), indent=1, sort_keys=True)
print(json.dumps(summary, indent=1, sort_keys=True))

# The fill stage is several kernels side by side: per step, the span from the first start to the last end of its launches (kernel
# trace), next to each kernel's own mean duration -> <tag>_stage_span.json; bench.py's roofline.ms.fill_and_traceback is this span.
trace = sorted(glob.glob(os.path.join(out, "stats", "**", "*kernel_trace.csv"), recursive=True), key=os.path.getsize)
if trace:
    STAGE = ("pgm_fill_kernel", "pgm_crit_kernel", "pgm_band_kernel", "pgm_lean_kernel", "pgm_tb_kernel")
    rows = [r for r in csv.DictReader(open(trace[-1])) if "pgm_" in r["Kernel_Name"]]
    rows.sort(key=lambda r: int(r["Start_Timestamp"]))
    steps, cur = [], None
    for r in rows:
        name = r["Kernel_Name"].replace("void ", "").split("(")[0].split("<")[0]
        if name == "pgm_prep_kernel":
            cur = {"start": None, "end": 0, "kernels": {}}
            steps.append(cur)
        elif cur is not None and name in STAGE:
            a, b = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
            cur["start"] = a if cur["start"] is None else min(cur["start"], a)
            cur["end"] = max(cur["end"], b)
            cur["kernels"].setdefault(name, []).append((a, b))
    steps = [st for st in steps if st["start"] is not None][1:]          # (the first step is the warm-up)
    if steps:
        span = [(st["end"] - st["start"]) / 1e6 for st in steps]
        per = {}
        for st in steps:
            for k, v in st["kernels"].items():
                for i, (a, b) in enumerate(sorted(v)):
                    per.setdefault("%s #%d" % (k, i), []).append(((a - st["start"]) / 1e6, (b - st["start"]) / 1e6))
        doc = {"steps": len(steps), "stage_span_ms_mean": sum(span) / len(span), "stage_span_ms_min": min(span), "stage_span_ms_max": max(span),
               "launches_ms_after_stage_start": {k: {"start": round(sum(x[0] for x in v) / len(v), 3), "end": round(sum(x[1] for x in v) / len(v), 3)} for k, v in sorted(per.items())}}
        json.dump(doc, open(os.path.join(dst, "%s_stage_span.json" % tag), "w"), indent=1)
        print(json.dumps(doc, indent=1))
