#!/bin/bash
# the end-to-end run (bench.py --workload e2e) with every phase's own timing line
cd ${GRAFT_REPO_ROOT:-.}
mkdir -p gpurun_out/e2e_timing
for rep in 1 2; do
NM_PLAN_TIMING=1 NM_POST_TIMING=1 NM_INGEST_TIMING=1 NM_SEARCH_TIMING=1 timeout 900 python bench.py --workload e2e --steps 3 --warmup 1 > gpurun_out/e2e_timing/e2e_$rep.json 2> gpurun_out/e2e_timing/e2e_$rep.log
echo "rc=$?"
grep -E "^\[nm_|^\[main|^\[bed" gpurun_out/e2e_timing/e2e_$rep.log | tail -12 | cut -c1-600
python3 - gpurun_out/e2e_timing/e2e_$rep.json <<'PY'
import json, sys
d = json.loads(open(sys.argv[1]).read().strip().split("\n")[-1]); t = d.get('timings_rank0', d.get('e2e', {}).get('timings_rank0', {}))
print({k: round(t.get(k, 0), 4) for k in ('upload_filter_s', 'window_pipeline_s', 'plan_s', 'background_s', 'native_search_s', 'postprocess_s', 'gpu_busy_s')}, 'ms/step', round(d.get('ms_per_step'), 2), 'busy/wall', d.get('gpu_busy_over_wall'))
PY
done
