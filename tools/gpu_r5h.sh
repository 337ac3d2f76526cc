#!/bin/bash
# round 5: cli1g after the pinned inflate results + pread block walk; then every profile of the round (profiles/r5)
cd ${GRAFT_REPO_ROOT:-.}
mkdir -p gpurun_out/r5h
timeout 2400 python bench.py --steps 3 --warmup 1 --extras cli1g --cpu-bins 0 > gpurun_out/r5h/cli1g.log 2>&1
echo "cli1g rc=$?"; tail -1 gpurun_out/r5h/cli1g.log | python3 -c "
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
bash tools/gpu_profiles.sh r5 2>&1 | tail -40
