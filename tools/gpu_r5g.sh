#!/bin/bash
# round 5: bench --extras cli1g at 1 Gbp with the pipelined device inflate + parallel block walk; NM_PLAN_TIMING of the end-to-end run
cd ${GRAFT_REPO_ROOT:-.}
mkdir -p gpurun_out/r5g
NM_PLAN_TIMING=1 NM_SEARCH_TIMING=1 timeout 600 python bench.py --workload e2e --steps 1 --warmup 0 --cpu-bins 0 > gpurun_out/r5g/e2e.log 2>&1
grep "nm_search\|nm_plan\|\[plan\]" gpurun_out/r5g/e2e.log | tail -5
timeout 2400 python bench.py --steps 3 --warmup 1 --extras cli1g --cpu-bins 0 > gpurun_out/r5g/cli1g.log 2>&1
echo "cli1g rc=$?"; tail -1 gpurun_out/r5g/cli1g.log | python3 -c "
import sys, json
d = json.loads(sys.stdin.readline()); c = d.get('cli1g', d.get('extra_errors'))
if isinstance(c, dict) and 'legs' in c:
    for leg, v in c['legs'].items():
        print(leg, json.dumps({k: (round(x, 3) if isinstance(x, float) else x) for k, x in v.get('phases', v).items()}), 'wall', v.get('wall_s'))
        for ln in v.get('parser_slab_log', [])[:40]: print('   ', ln)
    print('parity', c.get('parity', {}).get('byte_equal_to_the_oracle_pipeline'), 'both equal', c.get('both_runs_byte_equal'), 'write_s', c.get('write_s'))
else:
    print(json.dumps(c)[:3000])
"
