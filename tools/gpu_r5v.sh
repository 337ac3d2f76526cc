#!/bin/bash
# round 5: the inflate kernel with 25 KB of tables per workgroup (six per CU): device parser tests, then the CLI at 1 Gbp with slabs of 6 / 3 / 12 GiB
cd ${GRAFT_REPO_ROOT:-.}
mkdir -p gpurun_out/r5v
timeout 1200 python -m pytest tests/test_gpu_bed_device.py tests/test_gpu_cli.py -x -q -m gpu > gpurun_out/r5v/tests.log 2>&1; echo "tests rc=$?"; tail -3 gpurun_out/r5v/tests.log
NM_BENCH_CLI1G_LEGS="slab3g:NM_BED_INFLATE_SLAB=3221225472,slab12g:NM_BED_INFLATE_SLAB=12884901888" timeout 2400 python bench.py --extras cli1g --cpu-bins 0 --steps 3 --warmup 1 > gpurun_out/r5v/cli1g.log 2>&1
echo "cli1g rc=$?"; tail -1 gpurun_out/r5v/cli1g.log | python3 -c "
import sys, json
d = json.loads(sys.stdin.readline()); c = d.get('cli1g', d.get('extra_errors'))
if isinstance(c, dict) and 'legs' in c:
    for leg, v in c['legs'].items():
        print(leg, json.dumps({k: (round(x, 3) if isinstance(x, float) else x) for k, x in v.get('phases', v).items()}), 'wall', v.get('wall_s'), 'the wall is', v.get('the_wall_is'))
        for ln in v.get('parser_slab_log', [])[:4]: print('   ', ln)
    print('parity', c.get('parity', {}).get('byte_equal_to_the_oracle_pipeline'), 'both equal', c.get('both_runs_byte_equal'), 'write_s', c.get('write_s'))
else:
    print(json.dumps(c)[:3000])
"
