#!/bin/bash
# round 5: streams with a priority (= hardware queue) of their own for the inflate kernels and the second flight of the search: the CLI at
# 200 Mbp and the end-to-end run, each against the flat-priority streams and against GPU_MAX_HW_QUEUES=8
cd ${GRAFT_REPO_ROOT:-.}
mkdir -p gpurun_out/r5l
NM_BENCH_CLI1G_LEGS="flat:NM_BED_FLAT_PRIORITY=1;NM_BED_TIMING=1,q8:GPU_MAX_HW_QUEUES=8;NM_BED_TIMING=1,slab3g:NM_BED_INFLATE_SLAB=3221225472;NM_BED_TIMING=1" timeout 1800 python bench.py --steps 3 --warmup 1 --extras cli1g --cli1g-bp 200000000 --cpu-bins 0 > gpurun_out/r5l/cli.log 2>&1
echo "rc=$?"; tail -1 gpurun_out/r5l/cli.log | python3 -c "
import sys, json
d = json.loads(sys.stdin.readline()); c = d.get('cli1g', d.get('extra_errors'))
if isinstance(c, dict) and 'legs' in c:
    for leg, v in c['legs'].items():
        p = v.get('phases', v)
        print(leg, 'wall', round(v.get('wall_s', 0), 3), {k: round(p.get(k), 3) for k in ('pileup_s', 'pileup_read_s', 'pileup_inflate_s', 'pileup_parse_s', 'pileup_index_and_block_walk_s') if isinstance(p.get(k), float)})
        for ln in v.get('parser_slab_log', [])[1:4]: print('   ', ln)
    print('parity', c.get('parity', {}).get('byte_equal_to_the_oracle_pipeline'), 'all equal', c.get('both_runs_byte_equal'))
else:
    print(json.dumps(c)[:3000])
"
run() {
  tag=$1; shift
  env "$@" NM_SEARCH_TIMING=1 timeout 600 python bench.py --workload e2e --steps 1 --warmup 0 --cpu-bins 0 > gpurun_out/r5l/e2e_$tag.log 2>&1
  echo "== $tag rc=$?"; grep "nm_search" gpurun_out/r5l/e2e_$tag.log | tail -1
  tail -1 gpurun_out/r5l/e2e_$tag.log | python3 -c "import sys, json; d = json.loads(sys.stdin.readline()); t = d['timings_rank0']; print('wall', round(d['value'], 4), {k: round(t.get(k), 4) for k in ('gpu_busy_s', 'search_s', 'native_search_s', 'postprocess_s', 'upload_filter_s', 'background_s')})"
}
for rep in 1 2; do
run prio A=1
run flat NM_FLIGHT_FLAT_PRIORITY=1
run q8 GPU_MAX_HW_QUEUES=8
run q8flat GPU_MAX_HW_QUEUES=8 NM_FLIGHT_FLAT_PRIORITY=1
done
