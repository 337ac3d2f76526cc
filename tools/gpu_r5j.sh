#!/bin/bash
# round 5: after the wave-parallel factoring in spec_children_kernel and the mapping-based chunk copies: the affected suites, the
# end-to-end run, cli1g with inflate slabs of 1.5 GiB (default) and 3 GiB
cd ${GRAFT_REPO_ROOT:-.}
mkdir -p gpurun_out/r5j
timeout 1500 python -m pytest tests/test_gpu_search_speculation.py tests/test_gpu_bed_device.py tests/test_gpu_cli.py tests/test_gpu_synth.py tests/test_gpu_baseline_configs.py -x -q -m gpu > gpurun_out/r5j/tests.log 2>&1
echo "tests rc=$?" >> gpurun_out/r5j/tests.log
tail -4 gpurun_out/r5j/tests.log
for rep in 1 2 3; do
NM_SEARCH_TIMING=1 timeout 600 python bench.py --workload e2e --steps 1 --warmup 0 --cpu-bins 0 > gpurun_out/r5j/e2e_$rep.log 2>&1
grep "nm_search" gpurun_out/r5j/e2e_$rep.log | tail -1
tail -1 gpurun_out/r5j/e2e_$rep.log | python3 -c "import sys, json; d = json.loads(sys.stdin.readline()); t = d['timings_rank0']; print('wall', round(d['value'], 4), {k: round(t.get(k), 4) for k in ('gpu_busy_s', 'search_s', 'native_search_s', 'postprocess_s', 'upload_filter_s', 'background_s')})"
done
NM_BENCH_CLI1G_SLABS=3221225472 timeout 2400 python bench.py --steps 3 --warmup 1 --extras cli1g --cpu-bins 0 > gpurun_out/r5j/cli1g.log 2>&1
echo "cli1g rc=$?"; tail -1 gpurun_out/r5j/cli1g.log | python3 -c "
import sys, json
d = json.loads(sys.stdin.readline()); c = d.get('cli1g', d.get('extra_errors'))
if isinstance(c, dict) and 'legs' in c:
    for leg, v in c['legs'].items():
        print(leg, json.dumps({k: (round(x, 3) if isinstance(x, float) else x) for k, x in v.get('phases', v).items()}), 'wall', v.get('wall_s'))
        for ln in v.get('parser_slab_log', [])[:6]: print('   ', ln)
    print('parity', c.get('parity', {}).get('byte_equal_to_the_oracle_pipeline'), 'both equal', c.get('both_runs_byte_equal'), 'write_s', c.get('write_s'))
else:
    print(json.dumps(c)[:3000])
"
