#!/bin/bash
# step time and one kernel's time for several values of a tuning knob: bash scripts/dev/tune_step.sh <knob> "<v1 v2 ...>" <kernel key>
knob=$1; vals=$2; key=$3
for rep in 1 2; do for v in $vals; do
  python bench.py --no-cpu-baseline --no-other-configs --steps 200 --warmup 20 --tune $knob=$v 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.readlines()[-1]); print('$knob=$v', round(d['ms_per_step']*1000,1), '$key', d['kernels_us'].get('$key'))"
done; done
