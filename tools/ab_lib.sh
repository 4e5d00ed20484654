#!/bin/bash
# usage: ab_lib.sh variant ... : runs tools/probe_root.py with lib/libpgm_hip_<variant>.so in place of the release library
cd /root/repo
cp prographmsa_amd/lib/libpgm_hip.so /tmp/libpgm_keep.so
for v in "$@"; do
  cp prographmsa_amd/lib/libpgm_hip_$v.so prographmsa_amd/lib/libpgm_hip.so
  echo "== $v"; python tools/probe_root.py
done
cp /tmp/libpgm_keep.so prographmsa_amd/lib/libpgm_hip.so
