#!/bin/bash
# where the wall time of the 1 Gbp end-to-end run goes: lock-step loop split (NM_SEARCH_TIMING) + phase timings
cd ${GRAFT_REPO_ROOT:-.}
mkdir -p gpurun_out
for i in 1 2; do
NM_SEARCH_TIMING=1 timeout 900 python bench.py --workload e2e --steps 1 --warmup 0 --cpu-bins 0 > gpurun_out/e2e_timing_$i.json 2> gpurun_out/e2e_timing_$i.err
grep nm_search gpurun_out/e2e_timing_$i.err
python -c "
import json; d=json.loads(open('gpurun_out/e2e_timing_$i.json').read().strip().splitlines()[-1]); t=d['timings_rank0']; print(d['value'], {k: round(v,4) for k,v in t.items() if k.endswith('_s')})"
done
