#!/bin/bash
# round 5: workgroups per CU of bed_inflate_kernel through unused dynamic LDS (NM_BED_INFLATE_LDS_PAD): 6 (shipped) / 5 / 4 per CU, cli1g legs
cd ${GRAFT_REPO_ROOT:-.}
mkdir -p gpurun_out/r5ad
NM_BENCH_CLI1G_LEGS="wg5:NM_BED_INFLATE_LDS_PAD=6144,wg4:NM_BED_INFLATE_LDS_PAD=14336,wg4_one_stream:NM_BED_INFLATE_LDS_PAD=14336;NM_BED_ONE_INFLATE_STREAM=1,wg5_slab35:NM_BED_INFLATE_LDS_PAD=6144;NM_BED_INFLATE_SLAB=3758096384" timeout 2400 python bench.py --extras cli1g --cpu-bins 0 --steps 3 --warmup 1 > gpurun_out/r5ad/cli1g.log 2>&1
echo "cli1g rc=$?"; tail -1 gpurun_out/r5ad/cli1g.log | python3 -c "
import sys, json
d = json.loads(sys.stdin.readline()); c = d.get('cli1g', d.get('extra_errors'))
if isinstance(c, dict) and 'legs' in c:
    for leg, v in c['legs'].items():
        p = v.get('phases', v)
        print(leg, {k: (round(p.get(k), 3) if isinstance(p.get(k), float) else p.get(k)) for k in ('pileup_s', 'pileup_inflate_s', 'pileup_parse_s', 'filters_s', 'search_s')}, 'wall', v.get('wall_s'))
    print('parity', c.get('parity', {}).get('byte_equal_to_the_oracle_pipeline'), 'both equal', c.get('both_runs_byte_equal'))
else:
    print(json.dumps(c)[:3000])
"
