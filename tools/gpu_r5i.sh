#!/bin/bash
# round 5: cli1g with the compressed bytes read as file ranges (pread), inflate slabs of 1.5 GiB (default), 3 GiB and 0.75 GiB
cd ${GRAFT_REPO_ROOT:-.}
mkdir -p gpurun_out/r5i
NM_BENCH_CLI1G_SLABS=3221225472,805306368 timeout 2400 python bench.py --steps 3 --warmup 1 --extras cli1g --cpu-bins 0 > gpurun_out/r5i/cli1g.log 2>&1
echo "cli1g rc=$?"; tail -1 gpurun_out/r5i/cli1g.log | python3 -c "
import sys, json
d = json.loads(sys.stdin.readline()); c = d.get('cli1g', d.get('extra_errors'))
if isinstance(c, dict) and 'legs' in c:
    for leg, v in c['legs'].items():
        print(leg, json.dumps({k: (round(x, 3) if isinstance(x, float) else x) for k, x in v.get('phases', v).items()}), 'wall', v.get('wall_s'))
        for ln in v.get('parser_slab_log', [])[:8]: print('   ', ln)
    print('parity', c.get('parity', {}).get('byte_equal_to_the_oracle_pipeline'), 'both equal', c.get('both_runs_byte_equal'), 'write_s', c.get('write_s'))
else:
    print(json.dumps(c)[:3000])
"
