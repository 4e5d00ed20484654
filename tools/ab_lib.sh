#!/bin/bash
# usage: ab_lib.sh variant ... : runs tools/probe_root.py with lib/libpgm_hip_<variant>.so in place of the release library
# (the release library is put back when the script ends, however it ends)
cd "$(dirname "$0")/.." || exit 1
keep=$(mktemp /tmp/libpgm_keep.XXXXXX.so)
cp prographmsa_amd/lib/libpgm_hip.so "$keep"
trap 'cp "$keep" prographmsa_amd/lib/libpgm_hip.so; rm -f "$keep"' EXIT
export PROBE_DUMP=$(mktemp /tmp/ab_jobs.XXXXXX.bin)   # written by the first variant: put one with valid results first
rm -f "$PROBE_DUMP"
for v in "$@"; do
  cp prographmsa_amd/lib/libpgm_hip_$v.so prographmsa_amd/lib/libpgm_hip.so
  echo "== $v"; if [ "$v" = notb ]; then PROBE_NOFETCH=1 python tools/probe_root.py; else python tools/probe_root.py; fi
done
