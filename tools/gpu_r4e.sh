#!/bin/bash
# round 4: ingest_decide_kernel<true> with 4 / 2 / 1 pieces of 64 rows requested together (same-device A/B)
cd ${GRAFT_REPO_ROOT:-.}
for rep in 1 2; do
for v in "" u2 u1; do
  if [ -n "$v" ]; then export NM_LIB=$PWD/tools/_ab/libnmscan_$v.so; else unset NM_LIB; fi
  echo "== ${v:-u4}"
  bash tools/gpu_ingest_prof.sh r4_ingest_${v:-u4} 2>&1 | grep -E "^ingest" | tail -1
  python3 - <<PY
import csv
for r in csv.DictReader(open('gpurun_out/r4_ingest_${v:-u4}/kernel_stats.csv')):
    if 'decide' in r['Name'] or 'count_kernel' in r['Name']: print('   ', r['Name'].split('(')[1][-30:] if False else r['Name'][:60], '%.3f ms' % (float(r['AverageNs'])/1e6))
PY
done
done
