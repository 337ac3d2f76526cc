#!/bin/bash
# A/B kernel variants in ONE gpurun session (same device): tools/ab_bench.sh "<lib1> <lib2> ..." [rounds]
libs=${1:-"libnmscan.so"}; rounds=${2:-2}
cd ${GRAFT_REPO_ROOT:-.}
for r in $(seq $rounds); do
 for lib in $libs; do
  for w in "--workload cfg5" "--workload greedy --per-group 2" "--workload greedy --per-group 4"; do
   NM_LIB=$PWD/nanomotif_amd/$lib python bench.py $w --steps 20 --warmup 3 --cpu-bins 0 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('$lib', '$w', 'kernel_ms %.4f'%d['roofline']['kernel_ms'], 'ms/step %.3f'%d['ms_per_step'], 'value %.3e'%d['value'])"
  done
 done
done
