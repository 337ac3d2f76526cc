#!/bin/bash
# host threads of the native search after the round was split in two phases
cd ${GRAFT_REPO_ROOT:-.}
mkdir -p gpurun_out
for th in 1 2 4 8 1 2 4 8; do
NM_SEARCH_THREADS=$th NM_SEARCH_TIMING=1 timeout 900 python bench.py --workload e2e --steps 1 --warmup 0 --cpu-bins 0 > gpurun_out/e2e_z.json 2> gpurun_out/e2e_z.err
echo "threads $th: $(grep nm_search gpurun_out/e2e_z.err)"
python -c "
import json; d=json.loads(open('gpurun_out/e2e_z.json').read().strip().splitlines()[-1]); t=d['timings_rank0']; print(d['value'], t['native_search_s'])"
done
