#!/bin/bash
# usage: ab_lib.sh variant ... : runs tools/probe_root.py with lib/libpgm_hip_<variant>.so in place of the release library
cd /root/repo
cp prographmsa_amd/lib/libpgm_hip.so /tmp/libpgm_keep.so
export PROBE_DUMP=/tmp/ab_jobs.bin   # written by the first variant: put one with valid results first
for v in "$@"; do
  cp prographmsa_amd/lib/libpgm_hip_$v.so prographmsa_amd/lib/libpgm_hip.so
  echo "== $v"; if [ "$v" = notb ]; then PROBE_NOFETCH=1 python tools/probe_root.py; else python tools/probe_root.py; fi
done
cp /tmp/libpgm_keep.so prographmsa_amd/lib/libpgm_hip.so
