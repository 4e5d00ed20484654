#!/bin/bash
# A/B of several builds of libpgm_hip.so on the same GPU box: tools/ab_bench.sh  (expects lib/libpgm_hip_<name>.so files)
cd "$(dirname "$0")/.."
cp prographmsa_amd/lib/libpgm_hip.so /tmp/libpgm_hip_keep.so
for round in 1 2; do
for f in prographmsa_amd/lib/libpgm_hip_*.so; do
  v=$(basename $f .so); v=${v#libpgm_hip_}
  cp $f prographmsa_amd/lib/libpgm_hip.so
  python bench.py --no-cpu-baseline --no-extra 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('$v', d['value'], d['ms_per_step'], d['roofline']['ms'])"
done
done
cp /tmp/libpgm_hip_keep.so prographmsa_amd/lib/libpgm_hip.so
