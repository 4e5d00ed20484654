#!/bin/bash
# headline batch under different MODE 2 thresholds (tools build: PGM_MODE2_BANDS = fewest bands, PGM_MODE2_HD = shallowest history of a job swept one band per worker)
cd "$(dirname "$0")/.."
for v in "20 32" "20 64" "20 128" "24 128" "30 128"; do set -- $v; echo "== PGM_MODE2_BANDS=$1 PGM_MODE2_HD=$2"; PGM_HOST_PROFILE=1 PGM_TOOLS_LIB=1 PGM_MODE2_BANDS=$1 PGM_MODE2_HD=$2 python tools/probe_all.py 2>&1 | grep -a "work lists\|fill" | tail -2 | cut -c1-250; done
