#!/bin/bash
# Runs on the GPU box (through gpurun): rocprofv3 kernel trace + two PMC passes over bench.py, then the summaries the
# repository keeps under profiles/ (tools/summarize_profiles.py).  PMC passes are separate runs, as the MI355X guide asks.
set -e
cd "$(dirname "$0")/.."
OUT=gpurun_out/prof_$1
rm -rf "$OUT"; mkdir -p "$OUT"
export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/stats" -o run -- python3 bench.py --steps 5 --warmup 2 --no-cpu-baseline > "$OUT/bench_under_rocprof.json" 2> "$OUT/stats.err"
echo "kernel trace done"
rocprofv3 --pmc FETCH_SIZE --output-format csv -d "$OUT/fetch" -o run -- python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline > /dev/null 2> "$OUT/fetch.err"
echo "FETCH_SIZE pass done"
rocprofv3 --pmc WRITE_SIZE --output-format csv -d "$OUT/write" -o run -- python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline > /dev/null 2> "$OUT/write.err"
echo "WRITE_SIZE pass done"
python3 tools/summarize_profiles.py "$OUT" "$1"
