#!/bin/bash
# round 4: two task groups taking turns on the device (native search): parity suite, then the 1 Gbp end-to-end A/B
cd ${GRAFT_REPO_ROOT:-.}
mkdir -p gpurun_out/r4f
timeout 2400 python -m pytest tests -x -q -m gpu > gpurun_out/r4f/tests.log 2>&1; echo "tests rc=$?"; grep -E "passed|failed" gpurun_out/r4f/tests.log | tail -2
for rep in 1 2; do
for mode in two one noissuer; do
  unset NM_SEARCH_ONE_GROUP NM_SEARCH_NO_ISSUER; if [ $mode = one ]; then export NM_SEARCH_ONE_GROUP=1; fi; if [ $mode = noissuer ]; then export NM_SEARCH_NO_ISSUER=1; fi
  NM_SEARCH_TIMING=1 timeout 900 python bench.py --workload e2e > gpurun_out/r4f/e2e_${mode}_$rep.json 2> gpurun_out/r4f/e2e_${mode}_$rep.log
  echo "== $mode rep $rep rc=$?"
  grep "nm_search\]" gpurun_out/r4f/e2e_${mode}_$rep.log | tail -1
  python3 - <<PY
import json
d = json.load(open('gpurun_out/r4f/e2e_${mode}_$rep.json'))
e = d.get('e2e', d)
t = e.get('timings_rank0', {})
print({k: round(v, 4) if isinstance(v, float) else v for k, v in e.items() if k in ('wall_s', 'search_s', 'upload_filter_s', 'gpu_busy_s', 'rounds', 'candidates', 'motif_rows', 'gpu_busy_over_wall')}, {k: round(v, 4) for k, v in t.items() if isinstance(v, float) and k.endswith('_s')})
PY
done
done
