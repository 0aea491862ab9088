#!/bin/bash
# tools/tail_overlap_tune.sh "<max_active list>" "<reserve list>": bench step time with the GLR's
# early bands in the shadow of the greedy PCA's tail, for the hook threshold and the CUs the side
# stream leaves to the PCA (run through gpurun from the repo root)
for ma in ${1:-1 2 3}; do for rs in ${2:-32 64 96}; do
ORIGIN_GLR_SIDE_RESERVE=$rs timeout -k 10 300 python bench.py --steps 6 --warmup 2 --check off --no-cpu-baseline --e2e-size 0 --tail-max-active $ma 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print('max_active', $ma, 'reserve', $rs, d['ms_per_step'], d['without_local_max']['ms_per_step'], d['sequential']['ms_per_step'], d['config']['pca']['glr_bands'])"
done; done
