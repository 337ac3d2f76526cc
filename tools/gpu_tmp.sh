#!/bin/bash
cd ${GRAFT_REPO_ROOT:-.}
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_fuzz.py tests/test_gpu_lanes.py -x -q 2>&1 | tail -3
for mode in desc mask desc mask; do
if [ $mode = mask ]; then export NM_NO_DESC=1; else unset NM_NO_DESC; fi
timeout 600 python bench.py --cpu-bins 0 --extras none --hbm-round-steps 0 --steps 20 --warmup 5 > gpurun_out/b_$mode.json 2>/dev/null
python -c "
import json; d=json.loads(open('gpurun_out/b_$mode.json').read().strip().splitlines()[-1]); print('$mode', d['ms_per_step'], d['roofline']['kernel_ms'], d['roofline'].get('kernel'))"
done
