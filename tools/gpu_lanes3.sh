#!/bin/bash
mkdir -p gpurun_out/lanes
B="python bench.py --extras none --cpu-bins 0 --hbm-round-steps 0 --steps 100 --warmup 5"
for rep in 1 2; do
  NM_BENCH_OWN_STREAM=1 $B 2>/dev/null | python tools/bench_brief.py "own stream 1Gbp"
  NM_BENCH_OWN_STREAM=1 $B --total-bp 125000000 --contigs 1250 --bins 63 --candidates 1260 --force-allreduce 2>/dev/null | python tools/bench_brief.py "own stream 125Mbp+ar"
  NM_BENCH_OWN_STREAM=1 $B --total-bp 125000000 --contigs 1250 --bins 63 --candidates 1260 2>/dev/null | python tools/bench_brief.py "own stream 125Mbp"
done 2>&1 | tee gpurun_out/lanes/ab4.txt
python tools/gap_probe.py 125000000 1250 63 1260 1 2>/dev/null | tee -a gpurun_out/lanes/ab4.txt
python tools/gap_probe.py 125000000 1250 63 1260 2 2>/dev/null | tee -a gpurun_out/lanes/ab4.txt
python tools/gap_probe.py 1000000000 10000 500 10000 2 2>/dev/null | tee -a gpurun_out/lanes/ab4.txt
