#!/bin/bash
# A/B of kernel switches on one device: tools/gpu_ab.sh "<ENV=1 ...>;<...>" [rounds]   (';'-separated environment sets)
cd ${GRAFT_REPO_ROOT:-.}
mkdir -p gpurun_out/ab
IFS=';' read -ra SETS <<< "$1"
rounds=${2:-2}
out=gpurun_out/ab/${3:-ab}.log
: > $out
for r in $(seq $rounds); do
 for env in "${SETS[@]}"; do
  for w in "--workload greedy --per-group 2" "--workload greedy --per-group 4" "--workload cfg5"; do
   env $env python bench.py $w --steps 20 --warmup 3 --cpu-bins 0 --extras none --hbm-round-steps 0 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('$env | $w', 'kernel_ms %.4f'%d['roofline']['kernel_ms'], 'ms/step %.3f'%d['ms_per_step'], 'value %.3e'%d['value'], d['counts_checksum'])" >> $out
  done
 done
done
cat $out
