#!/bin/bash
# GPU box: SQ counters of the prep and emission kernels, two passes.  With tools/experiments/r4_emission_scalar_columns.patch applied the
# first pass runs pgm_emission_rows_kernel and the second (PGM_X_OLD_EMISSION=1) the shipped pgm_emission_skew_kernel; without it both
# passes run the shipped kernel.
cd "$(dirname "$0")/.."
export TMPDIR=/tmp
OUT=gpurun_out/pmc_em; rm -rf $OUT; mkdir -p $OUT
B="python3 bench.py --warmup 1 --only-headline --steps 2"
CNT="SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_SMEM SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU SQ_WAVES"
rocprofv3 --pmc $CNT --output-format csv -d $OUT/new -o run -- $B > /dev/null 2> $OUT/new.err && \
PGM_X_OLD_EMISSION=1 rocprofv3 --pmc $CNT --output-format csv -d $OUT/old -o run -- $B > /dev/null 2> $OUT/old.err
python3 - <<'PY'
import csv, glob, collections
for v in ("new", "old"):
    acc = collections.defaultdict(lambda: collections.defaultdict(list))
    for f in glob.glob("gpurun_out/pmc_em/%s/**/*counter_collection.csv" % v, recursive=True):
        for r in csv.DictReader(open(f)):
            k = r["Kernel_Name"]
            if "emission" in k or "prep" in k: acc[k[:60]][r["Counter_Name"]].append(float(r["Counter_Value"]))
    for k, c in acc.items():
        print(v, k, {n: round(sum(x) / len(x)) for n, x in c.items()}, "launches", len(next(iter(c.values()))))
PY
tail -3 $OUT/new.err
