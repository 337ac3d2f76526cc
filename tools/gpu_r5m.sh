#!/bin/bash
# round 5: the kept `bench.py --extras cli1g` run (1 Gbp from FASTA + .bed.gz + .tbi) -> profiles/r5/cli_1gbp.json
cd ${GRAFT_REPO_ROOT:-.}
mkdir -p gpurun_out/r5m
timeout 2400 python bench.py --extras cli1g,e2e --cpu-bins 0 > gpurun_out/r5m/cli1g.log 2>&1
echo "cli1g rc=$?"; tail -1 gpurun_out/r5m/cli1g.log | python3 -c "
import sys, json
d = json.loads(sys.stdin.readline()); c = d.get('cli1g', d.get('extra_errors'))
print('value', d['value'], 'frac', d['roofline']['frac'], 'stale', d['roofline'].get('traffic_stale'), 'e2e', {k: d.get('e2e', {}).get(k) for k in ('wall_s', 'gpu_busy_over_wall', 'rounds', 'search_iterations', 'speculation_hits', 'speculation_misses')})
if isinstance(c, dict) and 'legs' in c:
    for leg, v in c['legs'].items():
        print(leg, json.dumps({k: (round(x, 3) if isinstance(x, float) else x) for k, x in v.get('phases', v).items()}), 'wall', v.get('wall_s'), 'the wall is', v.get('the_wall_is'))
        for ln in v.get('parser_slab_log', [])[:6]: print('   ', ln)
    print('parity', c.get('parity', {}).get('byte_equal_to_the_oracle_pipeline'), 'both equal', c.get('both_runs_byte_equal'), 'write_s', c.get('write_s'))
else:
    print(json.dumps(c)[:3000])
"
