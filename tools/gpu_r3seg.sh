#!/bin/bash
# a batch whose candidates sit in few bins brings its own segment table (NM_ALL_SEGMENTS=1: the static table of every bin);
# compile + factor of small batches in LDS
cd ${GRAFT_REPO_ROOT:-.}
mkdir -p gpurun_out
timeout 1500 python -m pytest tests/test_gpu_parity.py tests/test_gpu_fuzz.py tests/test_gpu_lanes.py tests/test_gpu_per_contig.py tests/test_gpu_cli.py tests/test_gpu_baseline_configs.py -q 2>&1 | grep -E "passed|failed" | tail -2
for mode in own own all; do
if [ $mode = all ]; then export NM_ALL_SEGMENTS=1; else unset NM_ALL_SEGMENTS; fi
NM_SEARCH_TIMING=1 timeout 600 python bench.py --workload e2e --steps 1 --warmup 0 --cpu-bins 0 > gpurun_out/e2e_s.json 2> gpurun_out/e2e_s.err
grep "nm_search. 1000" gpurun_out/e2e_s.err | cut -c1-220
python -c "
import json; d=json.loads(open('gpurun_out/e2e_s.json').read().strip().splitlines()[-1]); t=d['timings_rank0']; print('$mode', round(d['value'],4), d['per_rank'][0]['motif_rows'], d['per_rank'][0]['planted_recovered'], {k: round(v,4) for k,v in t.items() if k in ('native_search_s','postprocess_s','gpu_busy_s')})"
done
