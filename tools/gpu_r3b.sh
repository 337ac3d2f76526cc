#!/bin/bash
cd ${GRAFT_REPO_ROOT:-.}
mkdir -p gpurun_out/r3b
timeout 1500 python -m pytest tests -q -m gpu -x -k "read_methylation or per_contig or failed_pileup or allocator or bench_json or many_mod or hit_positions" > gpurun_out/r3b/tests.log 2>&1
echo "tests rc=$?" >> gpurun_out/r3b/tests.log
tail -40 gpurun_out/r3b/tests.log
