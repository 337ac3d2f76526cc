#!/bin/bash
# round 5: bed_inflate_kernel with matches copied in bounded pieces: the parser suites, the parser probe (7.5 GB of text), the CLI at 200 Mbp
cd ${GRAFT_REPO_ROOT:-.}
mkdir -p gpurun_out/r5n
timeout 1500 python -m pytest tests/test_gpu_bed_device.py tests/test_gpu_cli.py -x -q -m gpu > gpurun_out/r5n/tests.log 2>&1
echo "tests rc=$?" >> gpurun_out/r5n/tests.log
tail -4 gpurun_out/r5n/tests.log
timeout 900 python3 tools/bed_probe.py 20000000 5 > gpurun_out/r5n/bed_probe.log 2>&1
grep -h "device parse" gpurun_out/r5n/bed_probe.log
NM_BENCH_CLI1G_LEGS="flatq:NM_BED_FLAT_PRIORITY=1;NM_BED_TIMING=1" timeout 1800 python bench.py --steps 3 --warmup 1 --extras cli1g --cli1g-bp 200000000 --cpu-bins 0 > gpurun_out/r5n/cli.log 2>&1
echo "rc=$?"; tail -1 gpurun_out/r5n/cli.log | python3 -c "
import sys, json
d = json.loads(sys.stdin.readline()); c = d.get('cli1g', d.get('extra_errors'))
if isinstance(c, dict) and 'legs' in c:
    for leg, v in c['legs'].items():
        p = v.get('phases', v)
        print(leg, 'wall', round(v.get('wall_s', 0), 3), {k: round(p.get(k), 3) for k in ('engine_start_s', 'fasta_s', 'pileup_s', 'pileup_read_s', 'pileup_inflate_s', 'pileup_parse_s', 'pileup_index_and_block_walk_s', 'pileup_plan_on_a_thread_s') if isinstance(p.get(k), float)})
        for ln in v.get('parser_slab_log', [])[1:4]: print('   ', ln)
    print('parity', c.get('parity', {}).get('byte_equal_to_the_oracle_pipeline'), 'all equal', c.get('both_runs_byte_equal'))
else:
    print(json.dumps(c)[:3000])
"
