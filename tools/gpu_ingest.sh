#!/bin/bash
mkdir -p gpurun_out/ingest
timeout 900 python -m pytest tests/test_gpu_ingest.py tests/test_gpu_cli.py tests/test_gpu_synth.py tests/test_gpu_baseline_configs.py -x -q -m gpu 2>&1 | tail -4
for i in 1 2; do python bench.py --workload e2e 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read().strip().split('\n')[-1]); t=d['timings_rank0']; print('e2e wall %.3f upload_filter %.3f search %.3f plan %.3f bg %.3f native %.3f co %.3f' % (d['value'], t['upload_filter_s'], t['search_s'], t['plan_s'], t['background_s'], t['native_search_s'], t['coroutines_s']))"; done
export TMPDIR=/tmp
rm -rf /tmp/pe; timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/pe -- python3 bench.py --workload e2e > gpurun_out/ingest/e2e_trace.log 2>&1
f=$(find /tmp/pe -name "*kernel_stats.csv" | head -1); head -12 $f | cut -c1-150
cp $f gpurun_out/ingest/e2e_kernel_stats.csv
