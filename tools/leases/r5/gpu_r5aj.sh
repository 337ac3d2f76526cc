#!/bin/bash
# round 5, end: the end-to-end run beyond BASELINE's size on one MI355X (2 and 4 Gbp, 2e9 / 4e9 raw rows), as profiles/r4/e2e_beyond_baseline.txt
cd ${GRAFT_REPO_ROOT:-.}
mkdir -p gpurun_out/r5aj
for G in 2 4; do
  timeout 1200 python bench.py --workload e2e --steps 2 --warmup 1 --total-bp ${G}000000000 --contigs ${G}0000 --bins $((G * 500)) > gpurun_out/r5aj/e2e_${G}g.log 2>&1
  echo "${G} Gbp rc=$?"
  tail -1 gpurun_out/r5aj/e2e_${G}g.log | python3 -c "
import sys, json
d = json.loads(sys.stdin.readline()); t = d.get('timings_rank0', d.get('e2e', {}).get('timings_rank0', {}))
print('wall %.3f s' % (d['ms_per_step'] / 1e3), {k: (round(t.get(k), 4) if isinstance(t.get(k), float) else t.get(k)) for k in ('rows_raw', 'rows_kept', 'upload_filter_s', 'plan_s', 'background_s', 'native_search_s', 'postprocess_s', 'gpu_busy_s', 'rounds', 'candidates', 'search_iterations')})
"
done
