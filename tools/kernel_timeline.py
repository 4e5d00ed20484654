"""Timeline of the kernels of one step of `bench.py --only-headline` from a rocprofv3 --kernel-trace CSV: start and end of every
launch relative to the step's pgm_prep_kernel (us).  usage: kernel_timeline.py <dir with *_kernel_trace.csv> [step]"""
import csv, glob, sys
f = glob.glob(sys.argv[1] + "/**/*kernel_trace.csv", recursive=True)[0]
rows = [r for r in csv.DictReader(open(f)) if r["Kernel_Name"].startswith("pgm_") or "pgm_" in r["Kernel_Name"]]
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
preps = [i for i, r in enumerate(rows) if "pgm_prep_kernel" in r["Kernel_Name"]]
k = int(sys.argv[2]) if len(sys.argv) > 2 else len(preps) - 2
lo, hi = preps[k], preps[k + 1] if k + 1 < len(preps) else len(rows)
t0 = int(rows[lo]["Start_Timestamp"])
for r in rows[lo:hi]:
    name = r["Kernel_Name"].split("(")[0].split("<")[0].replace("void ", "")
    print("%-28s grid %6d x %4d  start %8.1f  end %8.1f" % (name, int(r["Grid_Size_X"]) // max(1, int(r["Workgroup_Size_X"])), int(r["Workgroup_Size_X"]), (int(r["Start_Timestamp"]) - t0) / 1e3, (int(r["End_Timestamp"]) - t0) / 1e3))
