#!/bin/bash
# what rank 0 of an N-rank strong-scaling run costs per step, measured on one GPU with the shard it would hold and the WHOLE
# candidate table in every call (the library drops the candidates of bins it does not hold), all-reduce step of the C ABI included
B="python bench.py --extras two_lanes --cpu-bins 0 --hbm-round-steps 0 --steps 100 --warmup 5 --force-allreduce"
for rep in 1 2; do
for n in 2 4 8; do
  $B --as-rank-of $n 2>/dev/null | python tools/bench_brief.py "as rank 0 of $n"
done
done
