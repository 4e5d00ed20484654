#!/bin/bash
# usage: ab_jobtimes.sh variant ... : tools/probe_jobtimes.py (the headline batch: when each job's sweeps ended, when its traceback was
# published) with lib/libpgm_hip_<variant>.so in place of the release library; the first run uses the release library and makes the
# job dump (timing-experiment variants compute garbage).  The release library is put back when the script ends, however it ends.
cd "$(dirname "$0")/.." || exit 1
keep=$(mktemp /tmp/libpgm_keep.XXXXXX.so)
cp prographmsa_amd/lib/libpgm_hip.so "$keep"
trap 'cp "$keep" prographmsa_amd/lib/libpgm_hip.so; rm -f "$keep"' EXIT
export PROBE_DUMP=$(mktemp /tmp/ab_jobs.XXXXXX.bin)
rm -f "$PROBE_DUMP"
echo "== release"; python tools/probe_jobtimes.py 2>&1 | grep -E "^launch|^  [c<>]|job 254|job   0"
for v in "$@"; do
  cp prographmsa_amd/lib/libpgm_hip_$v.so prographmsa_amd/lib/libpgm_hip.so
  echo "== $v"; timeout -k 10 120 python tools/probe_jobtimes.py 2>&1 | grep -E "^launch|^  [c<>]|job 254|job   0"
done
