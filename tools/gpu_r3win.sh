#!/bin/bash
# window batches: fewer, fatter workgroups when the batch is large (NM_WIN_THIN=1: one per 256 words as before)
cd ${GRAFT_REPO_ROOT:-.}
mkdir -p gpurun_out
timeout 600 python -m pytest tests/test_gpu_windows.py -q 2>&1 | grep -E "passed|failed" | tail -1
for mode in fat thin fat thin; do
if [ $mode = thin ]; then export NM_WIN_THIN=1; else unset NM_WIN_THIN; fi
NM_SEARCH_TIMING=1 timeout 600 python bench.py --workload e2e --steps 1 --warmup 0 --cpu-bins 0 > gpurun_out/e2e_w.json 2> gpurun_out/e2e_w.err
grep "nm_search. 1000" gpurun_out/e2e_w.err | cut -c1-200
python -c "
import json; d=json.loads(open('gpurun_out/e2e_w.json').read().strip().splitlines()[-1]); t=d['timings_rank0']; print('$mode', round(d['value'],4), d['per_rank'][0]['motif_rows'], {k: round(v,4) for k,v in t.items() if k in ('native_search_s','gpu_busy_s')})"
done
