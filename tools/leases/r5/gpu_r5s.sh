#!/bin/bash
# round 5: the lock-step loop's time split into sending the batches and waiting for them
cd ${GRAFT_REPO_ROOT:-.}
mkdir -p gpurun_out/r5s
for rep in 1 2; do
  NM_SEARCH_TIMING=1 timeout 900 python bench.py --workload e2e --steps 3 --warmup 1 > gpurun_out/r5s/e2e_${rep}.log 2>&1
  grep "nm_search\]" gpurun_out/r5s/e2e_${rep}.log | tail -2
  NM_SEARCH_ONE_FLIGHT=1 NM_SEARCH_TIMING=1 timeout 900 python bench.py --workload e2e --steps 3 --warmup 1 > gpurun_out/r5s/e2e_one_${rep}.log 2>&1
  grep "nm_search\]" gpurun_out/r5s/e2e_one_${rep}.log | tail -2
done
