#!/bin/bash
# round 5: the kept cli1g run after the per-task files went to the end of the run (cold + again + a timing leg)
cd ${GRAFT_REPO_ROOT:-.}
mkdir -p gpurun_out/r5ag
NM_BENCH_CLI1G_LEGS="timing:NM_BED_TIMING=1;NM_INGEST_TIMING=1;NM_SEARCH_TIMING=1;NM_POST_TIMING=1;NM_PLAN_TIMING=1" timeout 2400 python bench.py --extras cli1g --cpu-bins 0 --steps 3 --warmup 1 > gpurun_out/r5ag/cli1g.log 2>&1
echo "cli1g rc=$?"; tail -1 gpurun_out/r5ag/cli1g.log | python3 -c "
import sys, json
d = json.loads(sys.stdin.readline()); c = d.get('cli1g', d.get('extra_errors'))
if isinstance(c, dict) and 'legs' in c:
    for leg, v in c['legs'].items():
        print(leg, json.dumps({k: (round(x, 3) if isinstance(x, float) else x) for k, x in v.get('phases', v).items()})[:1500], 'wall', v.get('wall_s'), 'the wall is', v.get('the_wall_is'))
    print('parity', c.get('parity', {}).get('byte_equal_to_the_oracle_pipeline'), 'both equal', c.get('both_runs_byte_equal'), 'write_s', c.get('write_s'))
else:
    print(json.dumps(c)[:3000])
"
