#!/bin/bash
# round 5, last call: the whole GPU suite, smoke, the default bench line, the kept cli1g run (timing leg)
cd ${GRAFT_REPO_ROOT:-.}
mkdir -p gpurun_out/r5af
timeout 3000 python -m pytest tests -x -q -m gpu > gpurun_out/r5af/tests.log 2>&1
echo "tests rc=$?" >> gpurun_out/r5af/tests.log
tail -4 gpurun_out/r5af/tests.log
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" > gpurun_out/r5af/smoke.log 2>&1; echo "smoke rc=$?"; tail -1 gpurun_out/r5af/smoke.log
timeout 900 python bench.py > gpurun_out/r5af/bench.log 2>&1; echo "bench rc=$?"
tail -1 gpurun_out/r5af/bench.log | python3 -c "
import sys, json
d = json.loads(sys.stdin.readline())
print('value', d['value'], 'ms', d['ms_per_step'], 'frac', d['roofline']['frac'], 'stale', d['roofline'].get('traffic_stale'), 'cpu', d.get('cpu_baseline', {}).get('value'), 'parity', d.get('parity'))
print('e2e', {k: d.get('e2e', {}).get(k) for k in ('wall_s', 'gpu_busy_over_wall', 'rounds', 'search_iterations', 'speculation_hits', 'speculation_misses')})
"
NM_BENCH_CLI1G_LEGS="timing:NM_BED_TIMING=1;NM_INGEST_TIMING=1;NM_SEARCH_TIMING=1;NM_POST_TIMING=1;NM_PLAN_TIMING=1" timeout 2400 python bench.py --extras cli1g --cpu-bins 0 --steps 3 --warmup 1 > gpurun_out/r5af/cli1g.log 2>&1
echo "cli1g rc=$?"; tail -1 gpurun_out/r5af/cli1g.log | python3 -c "
import sys, json
d = json.loads(sys.stdin.readline()); c = d.get('cli1g', d.get('extra_errors'))
if isinstance(c, dict) and 'legs' in c:
    for leg, v in c['legs'].items():
        print(leg, json.dumps({k: (round(x, 3) if isinstance(x, float) else x) for k, x in v.get('phases', v).items()})[:1500], 'wall', v.get('wall_s'), 'the wall is', v.get('the_wall_is'))
        for ln in v.get('parser_slab_log', [])[:60]:
            if 'slab' not in ln or 'slab 0:' in ln or 'slab 5:' in ln or 'slab 23' in ln: print('   ', ln[:400])
    print('parity', c.get('parity', {}).get('byte_equal_to_the_oracle_pipeline'), 'both equal', c.get('both_runs_byte_equal'), 'write_s', c.get('write_s'))
else:
    print(json.dumps(c)[:3000])
"
