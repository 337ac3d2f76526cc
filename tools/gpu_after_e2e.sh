#!/bin/bash
# why is the cfg 5 kernel slower when the e2e extra ran before it in the same process?
mkdir -p gpurun_out/lanes
B="python bench.py --cpu-bins 0 --hbm-round-steps 0 --steps 40 --warmup 5"
for rep in 1 2; do
  $B --extras none 2>/dev/null | python tools/bench_brief.py "extras none"
  $B --extras e2e 2>/dev/null | python tools/bench_brief.py "extras e2e"
  NM_BENCH_EMPTY_CACHE=1 $B --extras e2e 2>/dev/null | python tools/bench_brief.py "extras e2e + empty_cache"
  $B --extras e2e --cooldown 3 2>/dev/null | python tools/bench_brief.py "extras e2e + cooldown 3 s"
done 2>&1 | tee gpurun_out/lanes/after_e2e.txt
