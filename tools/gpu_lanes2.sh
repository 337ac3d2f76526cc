#!/bin/bash
mkdir -p gpurun_out/lanes
B="python bench.py --extras none --cpu-bins 0 --hbm-round-steps 0 --steps 60 --warmup 5"
S="--total-bp 125000000 --contigs 1250 --bins 63 --candidates 1260"
for rep in 1 2; do
for q in 4 8 16; do
for lanes in 1 2; do
  GPU_MAX_HW_QUEUES=$q $B --lanes $lanes $S 2>/dev/null | python tools/bench_brief.py "125Mbp queues=$q lanes=$lanes"
  GPU_MAX_HW_QUEUES=$q $B --lanes $lanes 2>/dev/null | python tools/bench_brief.py "1Gbp queues=$q lanes=$lanes"
done
done
done 2>&1 | tee gpurun_out/lanes/ab2.txt
