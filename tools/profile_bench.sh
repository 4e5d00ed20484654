#!/bin/bash
# Runs on the GPU box (through gpurun): rocprofv3 kernel trace + separate PMC passes over bench.py, then the summaries the
# repository keeps under profiles/ (tools/summarize_profiles.py).  PMC passes are separate runs with nothing but --pmc, as
# the MI355X guide asks (FETCH_SIZE and WRITE_SIZE do not fit one pass; the SQ block has 8 slots).
#   usage: tools/profile_bench.sh <tag>      e.g. r2
set -e
cd "$(dirname "$0")/.."
OUT=gpurun_out/prof_$1
rm -rf "$OUT"; mkdir -p "$OUT"
export TMPDIR=/tmp
B="python3 bench.py --warmup 1 --only-headline"
rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/stats" -o run -- $B --steps 10 > "$OUT/bench_under_rocprof.json" 2> "$OUT/stats.err"
echo "kernel trace done"
rocprofv3 --pmc FETCH_SIZE --output-format csv -d "$OUT/fetch" -o run -- $B --steps 2 > /dev/null 2> "$OUT/fetch.err"
echo "FETCH_SIZE pass done"
rocprofv3 --pmc WRITE_SIZE --output-format csv -d "$OUT/write" -o run -- $B --steps 2 > /dev/null 2> "$OUT/write.err"
echo "WRITE_SIZE pass done"
rocprofv3 --pmc SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_WAIT_ANY SQ_WAIT_INST_ANY --output-format csv -d "$OUT/sq" -o run -- $B --steps 2 > /dev/null 2> "$OUT/sq.err"
echo "SQ pass done"
python3 tools/summarize_profiles.py "$OUT" "$1"
# the satellites (all-pairs alignPair, context profiles) and the heavy-tailed configs: PMC passes of their own
S="python3 bench.py --only-satellites"
mkdir -p "$OUT/sat" "$OUT/c4" "$OUT/c5"
rocprofv3 --pmc SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_WAIT_ANY SQ_WAIT_INST_ANY --output-format csv -d "$OUT/sat/sq" -o run -- $S > /dev/null 2> "$OUT/sat/sq.err"
echo "satellites SQ pass done"
python3 tools/summarize_profiles.py "$OUT/sat" "$1_satellites" > /dev/null
for c in c4 c5; do
  rocprofv3 --pmc FETCH_SIZE --output-format csv -d "$OUT/$c/fetch" -o run -- python3 bench.py --only-config $c > /dev/null 2> "$OUT/$c/fetch.err"
  rocprofv3 --pmc WRITE_SIZE --output-format csv -d "$OUT/$c/write" -o run -- python3 bench.py --only-config $c > /dev/null 2> "$OUT/$c/write.err"
  python3 tools/summarize_profiles.py "$OUT/$c" "$1_$c" > /dev/null
  echo "$c passes done"
done
