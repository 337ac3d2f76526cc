#!/bin/bash
# round 5, final: the whole GPU suite, smoke, the default bench line, the kept cli1g run
cd ${GRAFT_REPO_ROOT:-.}
mkdir -p gpurun_out/r5o
timeout 3000 python -m pytest tests -x -q -m gpu > gpurun_out/r5o/tests.log 2>&1
echo "tests rc=$?" >> gpurun_out/r5o/tests.log
tail -5 gpurun_out/r5o/tests.log
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" > gpurun_out/r5o/smoke.log 2>&1; echo "smoke rc=$?"; tail -1 gpurun_out/r5o/smoke.log
timeout 900 python bench.py > gpurun_out/r5o/bench.log 2>&1; echo "bench rc=$?"
tail -1 gpurun_out/r5o/bench.log | python3 -c "
import sys, json
d = json.loads(sys.stdin.readline())
print('value', d['value'], 'ms', d['ms_per_step'], 'frac', d['roofline']['frac'], 'stale', d['roofline'].get('traffic_stale'), 'cpu', d.get('cpu_baseline', {}).get('value'), 'parity', d.get('parity'))
print('e2e', {k: d.get('e2e', {}).get(k) for k in ('wall_s', 'gpu_busy_over_wall', 'rounds', 'search_iterations', 'speculation_hits', 'speculation_misses')})
print('hbm round', {k: d.get('roofline_hbm_bound_round', {}).get(k) for k in ('kernel_ms', 'frac', 'counter_frac', 'counter_entry_stale')})
"
timeout 2400 python bench.py --extras cli1g --cpu-bins 0 --steps 3 --warmup 1 > gpurun_out/r5o/cli1g.log 2>&1
echo "cli1g rc=$?"; tail -1 gpurun_out/r5o/cli1g.log | python3 -c "
import sys, json
d = json.loads(sys.stdin.readline()); c = d.get('cli1g', d.get('extra_errors'))
if isinstance(c, dict) and 'legs' in c:
    for leg, v in c['legs'].items():
        print(leg, json.dumps({k: (round(x, 3) if isinstance(x, float) else x) for k, x in v.get('phases', v).items()}), 'wall', v.get('wall_s'), 'the wall is', v.get('the_wall_is'))
        for ln in v.get('parser_slab_log', [])[:5]: print('   ', ln)
    print('parity', c.get('parity', {}).get('byte_equal_to_the_oracle_pipeline'), 'both equal', c.get('both_runs_byte_equal'), 'write_s', c.get('write_s'))
else:
    print(json.dumps(c)[:3000])
"
