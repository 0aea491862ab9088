#!/bin/bash
# step time with / without the GLR started inside the PCA's tail, for the BASELINE field sizes
for n in ${1:-200 300 600 900}; do
timeout -k 10 400 python bench.py --size $n --steps ${2:-6} --warmup 2 --check off --no-cpu-baseline --e2e-size 0 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print('size', $n, 'chained', d['ms_per_step'], 'no-lm', d['without_local_max']['ms_per_step'], 'sequential', d['sequential']['ms_per_step'], d['config']['pca'].get('glr_bands'), 'iters', d['config']['pca']['pca_iters'])"
done
