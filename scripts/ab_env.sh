#!/bin/bash
# A/B of one environment switch on one box: bash scripts/ab_env.sh VAR valueA valueB [bench args]
VAR=$1; A=$2; B=$3; shift 3
for i in 1 2; do
for v in $A $B; do
env $VAR=$v python bench.py --no-cpu-baseline --no-infer-leg --steps 8 --warmup 3 "$@" 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$VAR=$v', round(d['value'],1), round(d['ms_per_step'],2))"
done; done
