#!/bin/bash
cd ${GRAFT_REPO_ROOT:-.}
mkdir -p gpurun_out
timeout 1200 python -m pytest tests/test_gpu_ingest.py tests/test_gpu_cli.py tests/test_gpu_bed_device.py tests/test_gpu_baseline_configs.py -x -q 2>&1 | tail -4
NM_INGEST_DENSE=1 timeout 1200 python -m pytest tests/test_gpu_ingest.py -x -q 2>&1 | tail -2
for mode in list dense list dense; do
if [ $mode = dense ]; then export NM_INGEST_DENSE=1; else unset NM_INGEST_DENSE; fi
timeout 600 python bench.py --workload e2e --steps 1 --warmup 0 --cpu-bins 0 > gpurun_out/e2e_d.json 2>/dev/null
python -c "
import json; d=json.loads(open('gpurun_out/e2e_d.json').read().strip().splitlines()[-1]); t=d['timings_rank0']; print('$mode', round(d['value'],4), d['per_rank'][0]['motif_rows'], d['per_rank'][0]['planted_recovered'], t['rows_kept'], t['rows_confident'], {k: round(v,4) for k,v in t.items() if k in ('upload_filter_s','gpu_busy_s')})"
done
python tools/ingest_probe.py 2>&1 | grep ingest | tail -3
