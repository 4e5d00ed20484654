#!/bin/bash
# headline batch under different CU shares of pgm_band_kernel (tools build: PGM_BAND_CUS, PGM_LEAN_CUS)
cd "$(dirname "$0")/.."
PGM_HOST_PROFILE=1 PGM_TOOLS_LIB=1 python tools/probe_all.py 2>&1 | grep -a "work lists\|fill" | tail -2
for v in ${BAND_CUS:-40 60 80 100 120}; do echo "== PGM_BAND_CUS=$v"; PGM_TOOLS_LIB=1 PGM_BAND_CUS=$v python tools/probe_all.py 2>&1 | tail -1; done
echo "== PGM_NO_BANDK=1"; PGM_TOOLS_LIB=1 PGM_NO_BANDK=1 python tools/probe_all.py 2>&1 | tail -1
