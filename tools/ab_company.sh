#!/bin/bash
# usage: ab_company.sh [NBLOCKS] : the root job of the headline batch alone (tools/probe_jobtimes.py, PROBE_TOP=1) while another process
# keeps NBLOCKS CUs busy with one kind of work (tools/micro/company): which shared resource slows a latency-bound sweep down?
cd "$(dirname "$0")/.." || exit 1
nb=${1:-200}
export PROBE_DUMP=$(mktemp /tmp/ab_jobs.XXXXXX.bin); rm -f "$PROBE_DUMP"
echo "== alone"; PROBE_TOP=1 PROBE_REPS=6 python tools/probe_jobtimes.py 2>&1 | grep -E "job   0"
for m in ${COMPANY_MODES:-alu lds poll mem}; do
  tools/micro/company $m $nb 60 & cp=$!
  sleep 4
  echo "== $m on $nb CUs"; PROBE_TOP=1 PROBE_REPS=6 timeout -k 10 120 python tools/probe_jobtimes.py 2>&1 | grep -E "job   0"
  kill $cp; wait $cp 2>/dev/null
done
