#!/bin/bash
# host threads of the native search (spinning workers, packed-key index)
cd ${GRAFT_REPO_ROOT:-.}
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_lanes.py tests/test_gpu_cli.py -x -q 2>&1 | tail -3
for th in 1 2 4 8 1 2 4 8; do
NM_SEARCH_THREADS=$th NM_SEARCH_TIMING=1 timeout 900 python bench.py --workload e2e --steps 1 --warmup 0 --cpu-bins 0 > gpurun_out/e2e_z.json 2> gpurun_out/e2e_z.err
echo "threads $th: $(grep nm_search gpurun_out/e2e_z.err)"
python -c "
import json; d=json.loads(open('gpurun_out/e2e_z.json').read().strip().splitlines()[-1]); t=d['timings_rank0']; print(d['value'], t['native_search_s'], d['per_rank'][0]['motif_rows'])"
done
