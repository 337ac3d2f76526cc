#!/bin/bash
# native post-processing: the CLI / pipeline parity tests, then the 1 Gbp end-to-end split (native against the Python twin)
cd ${GRAFT_REPO_ROOT:-.}
mkdir -p gpurun_out
timeout 1500 python -m pytest tests/test_gpu_cli.py tests/test_gpu_pipeline.py tests/test_gpu_search.py -x -q 2>&1 | tail -5
for mode in native native twin; do
if [ $mode = twin ]; then export NANOMOTIF_PY_POST=1; fi
NM_SEARCH_TIMING=1 timeout 900 python bench.py --workload e2e --steps 1 --warmup 0 --cpu-bins 0 > gpurun_out/e2e_post_$mode.json 2> gpurun_out/e2e_post_$mode.err
python -c "
import json; d=json.loads(open('gpurun_out/e2e_post_$mode.json').read().strip().splitlines()[-1]); t=d['timings_rank0']; print('$mode', d['value'], d['per_rank'][0]['motif_rows'], d['per_rank'][0]['planted_recovered'], {k: round(v,4) for k,v in t.items() if k.endswith('_s')})"
done
