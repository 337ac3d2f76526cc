#!/bin/bash
# host-count batches: preparation on the scoring stream + compile and common in one launch, A/B against the previous chain
cd ${GRAFT_REPO_ROOT:-.}
mkdir -p gpurun_out
timeout 1500 python -m pytest tests/test_gpu_parity.py tests/test_gpu_fuzz.py tests/test_gpu_cli.py tests/test_gpu_per_contig.py -x -q 2>&1 | tail -3
for mode in new old new old; do
if [ $mode = old ]; then export NM_NO_INLINE_PREP=1; else unset NM_NO_INLINE_PREP; fi
NM_SEARCH_TIMING=1 timeout 900 python bench.py --workload e2e --steps 1 --warmup 0 --cpu-bins 0 > gpurun_out/e2e_y_$mode.json 2> gpurun_out/e2e_y_$mode.err
grep nm_search gpurun_out/e2e_y_$mode.err
python -c "
import json; d=json.loads(open('gpurun_out/e2e_y_$mode.json').read().strip().splitlines()[-1]); t=d['timings_rank0']; print('$mode', d['value'], d['per_rank'][0]['motif_rows'], d['per_rank'][0]['planted_recovered'], {k: round(v,4) for k,v in t.items() if k.endswith('_s')})"
done
