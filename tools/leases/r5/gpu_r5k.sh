#!/bin/bash
# round 5: does the host-to-device copy of the next slab run BESIDE the inflate kernel?  The CLI at 200 Mbp (10 slabs) under a few environments
cd ${GRAFT_REPO_ROOT:-.}
mkdir -p gpurun_out/r5k
NM_BENCH_CLI1G_LEGS="q8:GPU_MAX_HW_QUEUES=8;NM_BED_TIMING=1,q2:GPU_MAX_HW_QUEUES=2;NM_BED_TIMING=1,nosdma:HSA_ENABLE_SDMA=0;NM_BED_TIMING=1,blit:GPU_FORCE_BLIT_COPY_SIZE=65536;NM_BED_TIMING=1,slab3g:NM_BED_INFLATE_SLAB=3221225472;NM_BED_TIMING=1" timeout 1800 python bench.py --steps 3 --warmup 1 --extras cli1g --cli1g-bp 200000000 --cpu-bins 0 > gpurun_out/r5k/cli.log 2>&1
echo "rc=$?"; tail -1 gpurun_out/r5k/cli.log | python3 -c "
import sys, json
d = json.loads(sys.stdin.readline()); c = d.get('cli1g', d.get('extra_errors'))
if isinstance(c, dict) and 'legs' in c:
    for leg, v in c['legs'].items():
        p = v.get('phases', v)
        print(leg, 'wall', round(v.get('wall_s', 0), 3), {k: round(p.get(k), 3) for k in ('pileup_s', 'pileup_read_s', 'pileup_inflate_s', 'pileup_parse_s', 'pileup_index_and_block_walk_s') if isinstance(p.get(k), float)})
        for ln in v.get('parser_slab_log', [])[1:5]: print('   ', ln)
    print('parity', c.get('parity', {}).get('byte_equal_to_the_oracle_pipeline'), 'all equal', c.get('both_runs_byte_equal'))
else:
    print(json.dumps(c)[:3000])
"
