#!/bin/bash
# round 5: inflate loop (decode + copy in one turn), page-table entries dropped while copying, lazy count tables: device parser / CLI tests, e2e, cli1g with timing legs
cd ${GRAFT_REPO_ROOT:-.}
mkdir -p gpurun_out/r5y
timeout 1500 python -m pytest tests/test_gpu_bed_device.py tests/test_gpu_cli.py tests/test_gpu_baseline_configs.py tests/test_gpu_windows.py -x -q -m gpu > gpurun_out/r5y/tests.log 2>&1; echo "tests rc=$?"; tail -3 gpurun_out/r5y/tests.log
for rep in 1 2; do
  NM_SEARCH_TIMING=1 timeout 900 python bench.py --workload e2e --steps 3 --warmup 1 > gpurun_out/r5y/e2e_${rep}.log 2>&1
  tail -1 gpurun_out/r5y/e2e_${rep}.log | python3 -c "
import sys, json
d = json.loads(sys.stdin.readline()); t = d.get('timings_rank0', d.get('e2e', {}).get('timings_rank0', {}))
print({k: round(t.get(k, 0), 4) for k in ('upload_filter_s', 'window_pipeline_s', 'plan_s', 'background_s', 'native_search_s', 'postprocess_s', 'gpu_busy_s')}, 'ms/step', round(d.get('ms_per_step'), 2), 'iters', t.get('search_iterations'))
"
done
NM_BENCH_CLI1G_LEGS="timing:NM_BED_TIMING=1;NM_INGEST_TIMING=1;NM_SEARCH_TIMING=1,slab35:NM_BED_INFLATE_SLAB=3758096384" timeout 2400 python bench.py --extras cli1g --cpu-bins 0 --steps 3 --warmup 1 > gpurun_out/r5y/cli1g.log 2>&1
echo "cli1g rc=$?"; tail -1 gpurun_out/r5y/cli1g.log | python3 -c "
import sys, json
d = json.loads(sys.stdin.readline()); c = d.get('cli1g', d.get('extra_errors'))
if isinstance(c, dict) and 'legs' in c:
    for leg, v in c['legs'].items():
        print(leg, json.dumps({k: (round(x, 3) if isinstance(x, float) else x) for k, x in v.get('phases', v).items()})[:1200], 'wall', v.get('wall_s'), 'the wall is', v.get('the_wall_is'))
        for ln in v.get('parser_slab_log', [])[:40]:
            if 'slab' not in ln or 'slab 0:' in ln or 'slab 5:' in ln or 'slab 23' in ln: print('   ', ln[:400])
    print('parity', c.get('parity', {}).get('byte_equal_to_the_oracle_pipeline'), 'both equal', c.get('both_runs_byte_equal'), 'write_s', c.get('write_s'))
else:
    print(json.dumps(c)[:3000])
"
