#!/bin/bash
# native search with the window / scoring halves overlapped and the per-task memo: parity tests, then the 1 Gbp split
cd ${GRAFT_REPO_ROOT:-.}
mkdir -p gpurun_out
timeout 1500 python -m pytest tests/test_gpu_cli.py tests/test_gpu_baseline_configs.py tests/test_gpu_windows.py tests/test_gpu_parity.py -x -q 2>&1 | tail -4
for mode in memo memo nomemo; do
if [ $mode = nomemo ]; then export NM_SEARCH_NO_MEMO=1; fi
NM_SEARCH_TIMING=1 timeout 900 python bench.py --workload e2e --steps 1 --warmup 0 --cpu-bins 0 > gpurun_out/e2e_x_$mode.json 2> gpurun_out/e2e_x_$mode.err
grep nm_search gpurun_out/e2e_x_$mode.err
python -c "
import json; d=json.loads(open('gpurun_out/e2e_x_$mode.json').read().strip().splitlines()[-1]); t=d['timings_rank0']; print('$mode', d['value'], d['per_rank'][0]['motif_rows'], d['per_rank'][0]['planted_recovered'], t['rounds'], t['candidates'], {k: round(v,4) for k,v in t.items() if k.endswith('_s')})"
done
