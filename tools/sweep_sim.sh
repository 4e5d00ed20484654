#!/bin/bash
# Sweep of the host work-list simulation parameters (experiments): tools/sweep_sim.sh
cd "$(dirname "$0")/.."
run() { python bench.py --no-cpu-baseline 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('$1', d['value'], d['ms_per_step'], d['roofline']['ms']['fill_and_traceback'])"; }
run default
PGM_SIM_TAU_X=0.6 run taux0.6
PGM_SIM_TAU_X=0.55 PGM_SIM_TAU_C=0.37 run taux0.55
PGM_SIM_TAU_X=0.9 run taux0.9
PGM_SIM_TAU_C=0.30 run tauc0.30
PGM_SIM_TAU_C=0.45 run tauc0.45
PGM_SIM_EAGER=0.5 run eager0.5
PGM_SIM_EAGER=0.85 run eager0.85
PGM_SIM_EAGER=1.1 run eager_off
PGM_FILL_WORKERS=640 run workers640
PGM_FILL_WORKERS=768 run workers768
run default
