#!/bin/bash
# headline batch under different MODE 2 thresholds (tools build: PGM_MODE2_BANDS = fewest bands of a job swept one band per worker)
cd "$(dirname "$0")/.."
for v in 20 23 26 30 40; do echo "== PGM_MODE2_BANDS=$v"; PGM_TOOLS_LIB=1 PGM_MODE2_BANDS=$v python tools/probe_all.py 2>&1 | tail -1; done
