#!/bin/bash
# fixed costs of pinned allocations / streams / first uses on a fresh process (tools/alloc_costs_probe.hip, built here into tools/build/)
cd ${GRAFT_REPO_ROOT:-.}
mkdir -p gpurun_out/alloc_costs
for i in 1 2; do timeout 120 tools/build/alloc_costs_probe > gpurun_out/alloc_costs/run$i.txt 2>&1; echo "rc=$?"; done
cat gpurun_out/alloc_costs/run2.txt
