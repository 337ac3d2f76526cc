#!/bin/bash
cd ${GRAFT_REPO_ROOT:-.}
mkdir -p gpurun_out
timeout 1200 python -m pytest tests/test_gpu_windows.py tests/test_gpu_lanes.py tests/test_gpu_cli.py tests/test_gpu_baseline_configs.py -x -q 2>&1 | grep -E "passed|failed|error" | tail -3
for mode in a b c; do
NM_SEARCH_TIMING=1 timeout 600 python bench.py --workload e2e --steps 1 --warmup 0 --cpu-bins 0 > gpurun_out/e2e_d.json 2> gpurun_out/e2e_d.err
grep "nm_search. 1000" gpurun_out/e2e_d.err | cut -c1-200
python -c "
import json; d=json.loads(open('gpurun_out/e2e_d.json').read().strip().splitlines()[-1]); t=d['timings_rank0']; print('$mode', round(d['value'],4), d['per_rank'][0]['motif_rows'], d['per_rank'][0]['planted_recovered'], {k: round(v,4) for k,v in t.items() if k in ('upload_filter_s','background_s','native_search_s','postprocess_s','gpu_busy_s')})"
done
