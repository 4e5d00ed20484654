#!/bin/bash
# A/B of two builds of libpgm_hip.so on the same GPU box: tools/ab_bench.sh  (expects lib/libpgm_hip_A.so and _B.so)
cd "$(dirname "$0")/.."
for round in 1 2; do
for v in A B; do
  cp prographmsa_amd/lib/libpgm_hip_$v.so prographmsa_amd/lib/libpgm_hip.so
  python bench.py --no-cpu-baseline 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('$v', d['value'], d['ms_per_step'], d['roofline']['ms'])"
done
done
