#!/bin/bash
# round 4: ingest_decide_kernel with / without the wave-uniform contig memo (same-device A/B at 1e9 rows)
cd ${GRAFT_REPO_ROOT:-.}
for rep in 1 2; do
for v in "" ${NM_AB_VARIANTS:-nomemo}; do
  if [ -n "$v" ]; then export NM_LIB=$PWD/tools/_ab/libnmscan_$v.so; else unset NM_LIB; fi
  echo "== ${v:-memo}"
  bash tools/gpu_ingest_prof.sh r4m_ingest_${v:-memo} 2>&1 | grep -E "^ingest" | tail -1
  python3 - <<PY
import csv
for r in csv.DictReader(open('gpurun_out/r4m_ingest_${v:-memo}/kernel_stats.csv')):
    if 'decide' in r['Name'] or 'count_kernel' in r['Name']: print('   ', r['Name'][:60], '%.3f ms' % (float(r['AverageNs'])/1e6))
PY
done
done
