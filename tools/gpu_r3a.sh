#!/bin/bash
# round 3, first GPU pass: whole GPU suite, smoke, default bench, and the N > 1 verification path with an RCCL world of one
cd ${GRAFT_REPO_ROOT:-.}
mkdir -p gpurun_out/r3a
timeout 3000 python -m pytest tests -q -m gpu > gpurun_out/r3a/tests.log 2>&1
echo "tests rc=$?" >> gpurun_out/r3a/tests.log
tail -15 gpurun_out/r3a/tests.log
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -2
timeout 900 python bench.py > gpurun_out/r3a/bench.json 2> gpurun_out/r3a/bench.log
echo "bench rc=$?"; tail -2 gpurun_out/r3a/bench.log
timeout 600 python bench.py --force-allreduce --extras two_lanes --cpu-bins 0 --hbm-round-steps 0 > gpurun_out/r3a/bench_fa.json 2> gpurun_out/r3a/bench_fa.log
echo "bench --force-allreduce rc=$?"; tail -2 gpurun_out/r3a/bench_fa.log
timeout 600 python bench.py --as-rank-of 4 --force-allreduce --extras two_lanes --cpu-bins 0 --hbm-round-steps 0 > gpurun_out/r3a/bench_as4.json 2> gpurun_out/r3a/bench_as4.log
echo "bench --as-rank-of 4 rc=$?"; tail -2 gpurun_out/r3a/bench_as4.log
