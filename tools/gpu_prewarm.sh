#!/bin/bash
mkdir -p gpurun_out/lanes
B="python bench.py --cpu-bins 0 --hbm-round-steps 0 --extras two_lanes"
for rep in 1 2 3; do
  $B --prewarm 0 2>/dev/null | python tools/bench_brief.py "K=20 W=3 prewarm 0"
  $B --prewarm 200 2>/dev/null | python tools/bench_brief.py "K=20 W=3 prewarm 200"
  $B --prewarm 1000 2>/dev/null | python tools/bench_brief.py "K=20 W=3 prewarm 1000"
  $B --prewarm 0 --steps 200 --warmup 20 2>/dev/null | python tools/bench_brief.py "K=200 W=20 prewarm 0"
done 2>&1 | tee gpurun_out/lanes/prewarm.txt
