#!/bin/bash
# Sweep of the host work-list simulation parameters (experiments): tools/sweep_sim.sh
cd "$(dirname "$0")/.."
run() { python bench.py --no-cpu-baseline --no-extra 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('$1', d['value'], d['ms_per_step'], d['roofline']['ms']['fill_and_traceback'])"; }
run default
PGM_SIM_TAU_2=0.6 run tau2_0.6
PGM_SIM_TAU_2=0.3 run tau2_0.3
PGM_SIM_TAU_X=0.5 run taux0.5
PGM_SIM_TAU_X=0.9 run taux0.9
PGM_SIM_TAU_C=0.35 run tauc0.35
PGM_SIM_TAU_C=0.6 run tauc0.6
PGM_SIM_EAGER=0.5 run eager0.5
PGM_SIM_EAGER=0.85 run eager0.85
PGM_SIM_EAGER=1.1 run eager_off
PGM_MODE2_BANDS=12 run mode2bands12
PGM_MODE2_BANDS=30 run mode2bands30
run default
