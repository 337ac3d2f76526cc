#!/bin/bash
# round 5: the block cache of the command line (NANOMOTIF_BLOCK_CACHE_GB=0: off): its test, the CLI tests, cli1g with and without
cd ${GRAFT_REPO_ROOT:-.}
mkdir -p gpurun_out/r5ah
timeout 1500 python -m pytest tests/test_gpu_block_cache.py tests/test_gpu_cli.py tests/test_gpu_integration_stub.py -x -q -m gpu > gpurun_out/r5ah/tests.log 2>&1; echo "tests rc=$?"; tail -12 gpurun_out/r5ah/tests.log
NM_BENCH_CLI1G_LEGS="nocache:NANOMOTIF_BLOCK_CACHE_GB=0;NM_INGEST_TIMING=1,timing:NM_INGEST_TIMING=1,nocache2:NANOMOTIF_BLOCK_CACHE_GB=0" timeout 2400 python bench.py --extras cli1g --cpu-bins 0 --steps 3 --warmup 1 > gpurun_out/r5ah/cli1g.log 2>&1
echo "cli1g rc=$?"; tail -1 gpurun_out/r5ah/cli1g.log | python3 -c "
import sys, json
d = json.loads(sys.stdin.readline()); c = d.get('cli1g', d.get('extra_errors'))
if isinstance(c, dict) and 'legs' in c:
    for leg, v in c['legs'].items():
        p = v.get('phases', v)
        print(leg, {k: (round(p.get(k), 3) if isinstance(p.get(k), float) else p.get(k)) for k in ('engine_start_s', 'fasta_s', 'pileup_s', 'pileup_inflate_s', 'pileup_parse_s', 'filters_s', 'search_s')}, 'wall', v.get('wall_s'), 'err' if 'error' in v else '')
        if 'error' in v: print(v['error'][-1500:])
        for ln in v.get('parser_slab_log', []):
            if 'nm_ingest' in ln: print('   ', ln[:260])
    print('parity', c.get('parity', {}).get('byte_equal_to_the_oracle_pipeline'), 'both equal', c.get('both_runs_byte_equal'))
else:
    print(json.dumps(c)[:3000])
"
