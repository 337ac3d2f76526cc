#!/bin/bash
cd ${GRAFT_REPO_ROOT:-.}
mkdir -p gpurun_out
for cfg in "4 64" "4 16" "4 8" "8 16" "8 8" "4 64" "4 16" "8 16"; do
set -- $cfg
NM_SEARCH_THREADS=$1 NM_SEARCH_MIN_PARALLEL=$2 NM_SEARCH_TIMING=1 timeout 900 python bench.py --workload e2e --steps 1 --warmup 0 --cpu-bins 0 > gpurun_out/e2e_z.json 2> gpurun_out/e2e_z.err
echo "threads $1 min_parallel $2: $(grep 'nm_search. 1000' gpurun_out/e2e_z.err | cut -c1-120) $(python -c "
import json; d=json.loads(open('gpurun_out/e2e_z.json').read().strip().splitlines()[-1]); t=d['timings_rank0']; print(round(d['value'],4), round(t['native_search_s'],4))")"
done
