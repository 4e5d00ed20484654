#!/bin/bash
# usage: ab_cfg.sh variant ... : fill stage of configs 3, 4 and 5 (tools/probe_jobtimes.py) with lib/libpgm_hip_<variant>.so in place of the release library
cd "$(dirname "$0")/.."
cp prographmsa_amd/lib/libpgm_hip.so /tmp/libpgm_keep.so
for v in "$@"; do
  cp prographmsa_amd/lib/libpgm_hip_$v.so prographmsa_amd/lib/libpgm_hip.so
  for c in c3 c4 c5; do echo "== $v $c: $(PROBE_CFG=$c python tools/probe_jobtimes.py 2>&1 | grep -a 'launch 1' | tail -1)"; done
done
cp /tmp/libpgm_keep.so prographmsa_amd/lib/libpgm_hip.so
