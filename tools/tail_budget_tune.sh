#!/bin/bash
# tools/tail_budget_tune.sh <size> "<budgets>": step time against the voxels of GLR given to the
# side stream at the tail hook (0 = every band that is ready)
size=${1:-900}
for b in ${2:-0 4e8 6e8 8.5e8 12e8}; do
timeout -k 10 400 python bench.py --size $size --steps 4 --warmup 2 --check off --no-cpu-baseline --e2e-size 0 --tail-early-budget $b 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print('size', $size, 'budget', '$b', d['ms_per_step'], d['without_local_max']['ms_per_step'], d['sequential']['ms_per_step'], d['config']['pca']['glr_bands'], d['wall_ms_per_step_by_phase'])"
echo "exit status $?"
done
