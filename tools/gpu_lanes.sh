#!/bin/bash
# Strict stream order against two scoring lanes (nm_set_score_lanes): full metagenome and the shard shapes of 2 / 4 / 8 ranks
# (whole bins per rank, as shard.assign_contigs deals them), with the all-reduce step of the C ABI (RCCL world of 1)
mkdir -p gpurun_out/lanes
B="python bench.py --extras two_lanes --cpu-bins 0 --hbm-round-steps 0 --steps 100 --warmup 5"
for rep in 1 2; do
  $B 2>/dev/null | python tools/bench_brief.py "1Gbp"
  $B --force-allreduce 2>/dev/null | python tools/bench_brief.py "1Gbp+ar"
  $B --total-bp 500000000 --contigs 5000 --bins 250 --candidates 5000 --force-allreduce 2>/dev/null | python tools/bench_brief.py "500Mbp+ar"
  $B --total-bp 250000000 --contigs 2500 --bins 125 --candidates 2500 --force-allreduce 2>/dev/null | python tools/bench_brief.py "250Mbp+ar"
  $B --total-bp 125000000 --contigs 1250 --bins 63 --candidates 1260 --force-allreduce 2>/dev/null | python tools/bench_brief.py "125Mbp+ar"
  $B --total-bp 125000000 --contigs 1250 --bins 63 --candidates 1260 2>/dev/null | python tools/bench_brief.py "125Mbp"
done 2>&1 | tee gpurun_out/lanes/ab5.txt
python bench.py > gpurun_out/lanes/bench_default.json 2> gpurun_out/lanes/bench_default.log; echo bench rc=$?
